#!/usr/bin/env python3
"""Randomised soak of the exhaustive bs=16 kernels through the C ABI against the C oracle: random frame sizes, window
sizes (all five size classes), pair counts (not multiples of 8), contents (pan, noise, flat, coarse grey levels), both
norms, one-tile and persistent schedules, with and without the redo path.  With a third argument "walk" the three
walking searches (three-step, 2-D log, diamond: k_walk16) are soaked instead.
usage: python tools/soak.py [seconds] [seed] [walk]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "global-motion-estimation_amd"), REPO, os.path.join(REPO, "tests")]
import numpy as np                     # noqa: E402
import _gme_native as native           # noqa: E402
from helpers import c_oracle           # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
WALK = len(sys.argv) > 3 and sys.argv[3] == "walk"
ctx = native.default_context()
co = c_oracle()
t0, cases, pairs = time.time(), 0, 0
while time.time() - t0 < budget:
    H, W = int(rng.integers(16, 220)), int(rng.integers(16, 340))
    if WALK:
        H, W = max(H, 17), max(W, 17)          # the diamond search refuses frames without room for a block (bbme.py:503-505)
    n = int(rng.integers(2, 30))
    sw = int(rng.choice([0, 4, 8, 12, 16, 20, 24, 28, 32]))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        base = rng.integers(0, 256, (H + 80, W + 80), dtype=np.uint8)
        frames = np.stack([base[40 + 2 * (t % 7):40 + 2 * (t % 7) + H, 40 - 3 * (t % 5):40 - 3 * (t % 5) + W] for t in range(n)])
    elif kind == 1:
        frames = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    elif kind == 2:
        frames = np.full((n, H, W), int(rng.integers(0, 256)), np.uint8)
    elif kind == 3:
        base = (rng.integers(0, 3, (H + 40, W + 40)) * 100).astype(np.uint8)
        frames = np.stack([base[20 + (t % 3):20 + (t % 3) + H, 20 - (t % 4):20 - (t % 4) + W] for t in range(n)])
    else:                              # half noise, half pan
        base = rng.integers(0, 256, (H + 80, W + 80), dtype=np.uint8)
        frames = np.stack([base[40 + (t % 5):40 + (t % 5) + H, 40 - 2 * (t % 6):40 - 2 * (t % 6) + W] for t in range(n)]).copy()
        frames[:, :, :W // 2] = rng.integers(0, 256, (n, H, W // 2), dtype=np.uint8)
    frames = np.ascontiguousarray(frames)
    os.environ["GME_SEA_PERSIST"] = str(int(rng.choice([0, 1, 2, 2])))
    os.environ["GME_SEA_REDO"] = str(int(rng.choice([1, 1, 0])))
    if rng.integers(0, 4) == 0:
        os.environ["GME_SEA_REDO_FRAC"] = "0.05"
    else:
        os.environ.pop("GME_SEA_REDO_FRAC", None)
    seq = native.Sequence.from_frames(ctx, frames)
    fd = int(rng.integers(1, min(3, n - 1) + 1))
    for pn in (0, 1):
        proc = int(rng.integers(1, 4)) if WALK else 0
        if WALK and proc != 3 and sw == 0:
            sw = 4
        seq.bbme(fd, 16, sw, proc, pn)
        mv = seq.read_mv()
        info = ctx.last_bbme_info()
        for p in range(n - fd):
            want = co.bbme(frames[p], frames[p + fd], 16, sw, proc, pn)
            if not np.array_equal(mv[p], want):
                print("MISMATCH", dict(H=H, W=W, n=n, sw=sw, proc=proc, kind=kind, fd=fd, pn=pn, pair=p, env={k: v for k, v in os.environ.items() if k.startswith("GME_")}, info=info))
                sys.exit(1)
            pairs += 1
    seq.close()
    cases += 1
print("soak ok: %d cases, %d pair checks in %.0f s" % (cases, pairs, time.time() - t0))
