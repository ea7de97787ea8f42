#!/usr/bin/env python3
"""One warm StreamEstimator.run under rocprofv3 (--kernel-trace --memory-copy-trace): tools/r03_trace_stream.sh prints
how busy the host-to-device copies kept the link and what ran in the gaps."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "global-motion-estimation_amd"), REPO]
import _gme_native as native            # noqa: E402
import sequence                         # noqa: E402
import synth                            # noqa: E402

n, H, W = 2049, 480, 720
frames = native.pinned_empty((n, H, W))
frames[...] = synth.sequence(1234, 0, 16, H, W)[[i % 16 for i in range(n)]]
chunk, lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 2
with sequence.StreamEstimator(H, W, 1, chunk, lanes) as est:
    est.run(frames, exact_psnr=False)
    est.run(frames, exact_psnr=False)
