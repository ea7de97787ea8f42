#!/usr/bin/env python3
"""sequence.estimate_stream (video in host memory, chunked upload overlapped with the estimate) against the copy alone
and against the resident path, 720x480: what bench.py prints as pcie_inclusive for gme720, at several chunk sizes."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "global-motion-estimation_amd"), REPO]
import numpy as np                      # noqa: E402
import _gme_native as native            # noqa: E402
import sequence                         # noqa: E402

n, H, W = 2049, 480, 720
ctx = native.default_context()
seq = native.Sequence(ctx, n, H, W)
seq.synth(1234, 0)
frames = native.pinned_empty((n, H, W))
for i in range(n):
    frames[i] = seq.read_frame(i)
seq.upload(0, frames)                   # first touch: the staging buffer is allocated here
t0 = time.perf_counter()
seq.upload(0, frames)
t_copy = time.perf_counter() - t0
seq.close()
print("copy alone: %.2f ms = %.1f GB/s -> %.0f pairs/s if nothing else" % (1e3 * t_copy, n * H * W / t_copy / 1e9, (n - 1) / t_copy))
for chunk, lanes, mn in ((128, 3, 64), (128, 4, 64), (192, 3, 64), (128, 3, 32), (256, 3, 64), (128, 3, 128), (512, 2, 64)):
    with sequence.StreamEstimator(H, W, 1, chunk, lanes, min_chunk=mn) as est:
        est.run(frames, exact_psnr=False)      # first touch
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            p, psnr = est.run(frames, exact_psnr=False)
            best = min(best, time.perf_counter() - t0)
    print("chunk %4d (min %3d) lanes %d: %.2f ms  %.0f pairs/s  %.1f GB/s  (%.0f%% of the copy ceiling)" % (
        chunk, mn, lanes, 1e3 * best, (n - 1) / best, n * H * W / best / 1e9, 100 * t_copy / best))
t0 = time.perf_counter()
sequence.estimate_stream(frames, 1, chunk_pairs=128, streams=2, exact_psnr=False)
print("one-shot estimate_stream (lanes set up and released inside): %.2f ms" % (1e3 * (time.perf_counter() - t0)))
