#!/usr/bin/env python3
"""Per-chunk timeline of a StreamEstimator run (host clock): when each chunk's upload was queued and finished and when its
level-1 sums, level-2 sums and squared errors arrived.  usage: stream_gme_timeline.py [chunk] [lanes] [min_chunk]"""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "global-motion-estimation_amd"), REPO]
import _gme_native as native            # noqa: E402
import sequence                         # noqa: E402
import synth                            # noqa: E402

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
min_chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 512
n, H, W = 2049, 480, 720
frames = native.pinned_empty((n, H, W))
frames[...] = synth.sequence(1234, 0, 16, H, W)[[i % 16 for i in range(n)]]
log = []
t0 = [0.0]
for name in ("upload", "gme_begin_fit", "gme_fit", "compensate"):
    fn = getattr(native.Sequence, name)

    def wrap(self, *a, _fn=fn, _name=name, **k):
        log.append((1e3 * (time.perf_counter() - t0[0]), id(self) % 1000, _name, self.N - 1))
        return _fn(self, *a, **k)
    setattr(native.Sequence, name, wrap)
with sequence.StreamEstimator(H, W, 1, chunk, lanes, min_chunk=min_chunk) as est:
    est.run(frames, exact_psnr=False)
    log.clear()
    t0[0] = time.perf_counter()
    est.run(frames, exact_psnr=False)
    total = 1e3 * (time.perf_counter() - t0[0])
print("chunk %d lanes %d min %d: %.2f ms" % (chunk, lanes, min_chunk, total))
for t, lane, name, pairs in log:
    print("  %7.2f ms  lane %3d  %-14s %4d pairs" % (t, lane, name, pairs))
