#!/usr/bin/env python3
"""Where the host thread of a StreamEstimator run spends its time: every _gme_native.Sequence call is timed.
usage: stream_gme_hosttime.py [chunk] [lanes] [min_chunk]"""
import collections
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "global-motion-estimation_amd"), REPO]
import _gme_native as native            # noqa: E402
import motion                           # noqa: E402
import sequence                         # noqa: E402
import synth                            # noqa: E402

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
min_chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n, H, W = 2049, 480, 720
frames = native.pinned_empty((n, H, W))
frames[...] = synth.sequence(1234, 0, 16, H, W)[[i % 16 for i in range(n)]]
tot, cnt, worst = collections.Counter(), collections.Counter(), collections.Counter()


def timed(cls, name):
    fn = getattr(cls, name)

    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t
            tot[name] += dt
            cnt[name] += 1
            worst[name] = max(worst[name], dt)
    setattr(cls, name, wrap)


for m in ("upload", "set_frames", "gme_begin_fit", "gme_fit", "compensate", "wait", "poll"):
    timed(native.Sequence, m)
solve = motion._solve_batch


def solve_t(x):
    t = time.perf_counter()
    try:
        return solve(x)
    finally:
        tot["solve"] += time.perf_counter() - t
        cnt["solve"] += 1


with sequence.StreamEstimator(H, W, 1, chunk, lanes, min_chunk=min_chunk) as est:
    est.run(frames, exact_psnr=False, solve=solve_t)
    tot.clear(); cnt.clear(); worst.clear()
    t0 = time.perf_counter()
    est.run(frames, exact_psnr=False, solve=solve_t)
    el = time.perf_counter() - t0
print("chunk %d lanes %d min %d: %.2f ms, %d chunks" % (chunk, lanes, min_chunk, 1e3 * el, len(est.schedule(n - 1))))
for k, v in tot.most_common():
    print("  %-14s %7.2f ms in %5d calls, worst %.3f ms" % (k, 1e3 * v, cnt[k], 1e3 * worst[k]))
print("  %-14s %7.2f ms" % ("other (python)", 1e3 * (el - sum(tot.values()))))
