#!/usr/bin/env python3
"""CPU study (NumPy, no GPU): how much would a SECOND bound level prune in the exhaustive bs = 16 search?
For every block of a frame pair and every candidate of its window (bbme.py:146-177: 2 sw + 16 positions per axis,
out-of-frame candidates skipped) it computes the exact cost (SAD / SSD), the quadrant bound the kernels use
(four 8x8 cells: sum |dS| <= SAD, sum dS^2 / 64 <= SSD) and finer bounds over 8 cells (8x4) and 16 cells (4x4:
sum dS^2 / 16), then reports, against the BEST upper bound a kernel could have (the block's true minimum):
  * the share of candidates whose own bound does not exceed it (what a per-candidate test would have to evaluate),
  * the share of PATCHES (3 rows x 4 columns of candidates, the unit of phase D) with at least one such candidate
    (what the kernels evaluate today: every candidate of a surviving patch).
usage: python tools/bound_study.py [synthetic|race|pan240x2] [pairs]     (about 20 s per pair at 720x480)"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "global-motion-estimation_amd"), REPO]
import numpy as np                     # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
BS, SW = 16, 16
NC = 2 * SW + 16

if kind == "synthetic":
    import synth
    frames = synth.sequence(1234, 0, n_pairs + 1, 480, 720)
else:
    import bench
    frames, _, _ = bench.host_content(kind, n_pairs + 1, 480, 720)
frames = np.asarray(frames)


def cell_sums(img, ch, cw):
    """sums over ch x cw cells anchored at every pixel: out[y, x] = sum img[y:y+ch, x:x+cw] (int64)"""
    ii = np.zeros((img.shape[0] + 1, img.shape[1] + 1), np.int64)
    ii[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
    return ii[ch:, cw:] - ii[:-ch, cw:] - ii[ch:, :-cw] + ii[:-ch, :-cw]


tot = {}
t0 = time.time()
for p in range(n_pairs):
    prev, cur = frames[p].astype(np.int32), frames[p + 1].astype(np.int32)
    H, W = prev.shape
    nbr, nbc = H // BS, W // BS
    r0 = (np.arange(nbr) * BS)[:, None]
    c0 = (np.arange(nbc) * BS)[None, :]
    cost = {0: np.full((NC, NC, nbr, nbc), np.iinfo(np.int64).max), 1: np.full((NC, NC, nbr, nbc), np.iinfo(np.int64).max)}
    lb = {(pn, cells): np.zeros((NC, NC, nbr, nbc), np.int64) for pn in (0, 1) for cells in (4, 8, 16)}
    geo = {4: (8, 8), 8: (8, 4), 16: (4, 4)}
    sums_prev = {c: cell_sums(prev, *geo[c]) for c in geo}
    sums_cur = {c: cell_sums(cur, *geo[c]) for c in geo}
    valid = np.zeros((NC, NC, nbr, nbc), bool)
    for ci in range(NC):                      # column offset in the outer loop like bbme.py:146-149
        for ri in range(NC):
            dy, dx = ri - SW, ci - SW
            ok = (r0 + dy >= 0) & (r0 + dy <= H - BS) & (c0 + dx >= 0) & (c0 + dx <= W - BS)
            if not ok.any():
                continue
            valid[ci, ri] = ok
            # shifted view of cur aligned with prev's blocks, zero where the candidate leaves the frame
            ys, xs = max(0, -dy), max(0, -dx)
            ye, xe = min(H, H - dy), min(W, W - dx)
            d = np.zeros((H, W), np.int64)
            d[ys:ye, xs:xe] = prev[ys:ye, xs:xe] - cur[ys + dy:ye + dy, xs + dx:xe + dx]
            blk = lambda a: a[:nbr * BS, :nbc * BS].reshape(nbr, BS, nbc, BS).sum((1, 3))    # noqa: E731
            cost[0][ci, ri] = np.where(ok, blk(np.abs(d)), cost[0][ci, ri])
            cost[1][ci, ri] = np.where(ok, blk(d * d), cost[1][ci, ri])
            for cells, (ch, cw) in geo.items():
                sp, sc = sums_prev[cells], sums_cur[cells]
                l1 = np.zeros((nbr, nbc), np.int64)
                l2 = np.zeros((nbr, nbc), np.int64)
                for a in range(0, BS, ch):
                    for b in range(0, BS, cw):
                        yy, xx = r0 + a, c0 + b
                        ya, xa = np.clip(yy + dy, 0, sc.shape[0] - 1), np.clip(xx + dx, 0, sc.shape[1] - 1)
                        dd = sp[yy, xx] - sc[ya, xa]
                        l1 += np.abs(dd)
                        l2 += dd * dd
                lb[(0, cells)][ci, ri] = l1
                lb[(1, cells)][ci, ri] = l2 // (ch * cw)          # floor keeps it a lower bound of the integer SSD
    for pn in (0, 1):
        best = cost[pn].reshape(-1, nbr, nbc).min(0)
        n_valid = valid.sum()
        for cells in (4, 8, 16):
            passes = valid & (lb[(pn, cells)] <= best[None, None])
            # patches: 4 columns x 3 rows of candidates (phase D's unit)
            pp = passes.reshape(NC // 4, 4, NC // 3, 3, nbr, nbc).any((1, 3))
            pv = valid.reshape(NC // 4, 4, NC // 3, 3, nbr, nbc).any((1, 3))
            key = ("MAE" if pn == 0 else "MSE", cells)
            t = tot.setdefault(key, [0, 0, 0, 0, 0])
            t[0] += passes.sum(); t[1] += n_valid; t[2] += pp.sum(); t[3] += pv.sum()
            if cells == 16:            # candidates inside patches that survive the QUADRANT test: what a second level sees
                q = (valid & (lb[(pn, 4)] <= best[None, None])).reshape(NC // 4, 4, NC // 3, 3, nbr, nbc).any((1, 3))
                inside = valid.reshape(NC // 4, 4, NC // 3, 3, nbr, nbc) & q[:, None, :, None]
                t[4] += inside.sum()
    print("pair %d done, %.0f s" % (p, time.time() - t0), flush=True)

print("content %s, %d pair(s), 720x480 bs 16 sw 16, upper bound = each block's true minimum" % (kind, n_pairs))
for (norm, cells), t in sorted(tot.items()):
    line = "%s %2d cells: candidates with bound <= UB %6.3f %%   patches holding one %6.3f %%" % (
        norm, cells, 100.0 * t[0] / t[1], 100.0 * t[2] / t[3])
    if cells == 16:
        q = tot[(norm, 4)]
        line += "   | inside quadrant-surviving patches: %.1f %% pass the quadrant test themselves, %.1f %% the 16-cell test" % (
            100.0 * q[0] / t[4], 100.0 * t[0] / t[4])
    print(line)
