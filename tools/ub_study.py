#!/usr/bin/env python3
"""CPU study (NumPy, no GPU): which UPPER bound would phase C of the elimination kernels need on real content?
For every block of a frame pair it computes every candidate's exact cost and quadrant bound (as tools/bound_study.py)
and then the share of phase-D patches (3 rows x 4 columns of candidates) that survive under several ways of
choosing the candidates whose true cost becomes the upper bound:
  P2     min-bound candidate + zero vector                       (rounds 1-2)
  P3     + the winner of the block one tile (4 blocks) to the left      (round 3, approximated)
  PP     the whole patches around the P3 probes
  TOPk   the k patches with the smallest bounds, evaluated whole
  NB     P3 + the P3 winners of the other blocks of the 2x4 tile
  TRUE   the block's true minimum (what no kernel can know in advance)
usage: python tools/ub_study.py [synthetic|race|pan240x2] [pairs] [first pair]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "global-motion-estimation_amd"), REPO]
import numpy as np                     # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
BS, SW, R = 16, 16, 3
NC = 2 * SW + 16
BIG = 1 << 40

if kind == "synthetic":
    import synth
    frames = synth.sequence(1234, first, n_pairs + 1, 480, 720)
else:
    import bench
    frames, _, _ = bench.host_content(kind, first + n_pairs + 1, 480, 720)
    frames = frames[first:]
frames = np.asarray(frames)


def cell_sums(img, ch, cw):
    ii = np.zeros((img.shape[0] + 1, img.shape[1] + 1), np.int64)
    ii[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
    return ii[ch:, cw:] - ii[:-ch, cw:] - ii[ch:, :-cw] + ii[:-ch, :-cw]


def tables(prev, cur):
    H, W = prev.shape
    nbr, nbc = H // BS, W // BS
    r0 = (np.arange(nbr) * BS)[:, None]
    c0 = (np.arange(nbc) * BS)[None, :]
    cost = {pn: np.full((NC, NC, nbr, nbc), BIG, np.int64) for pn in (0, 1)}
    lb = {pn: np.full((NC, NC, nbr, nbc), BIG, np.int64) for pn in (0, 1)}
    sp, sc = cell_sums(prev, 8, 8), cell_sums(cur, 8, 8)
    for ci in range(NC):
        for ri in range(NC):
            dy, dx = ri - SW, ci - SW
            ok = (r0 + dy >= 0) & (r0 + dy <= H - BS) & (c0 + dx >= 0) & (c0 + dx <= W - BS)
            if not ok.any():
                continue
            ys, xs = max(0, -dy), max(0, -dx)
            ye, xe = min(H, H - dy), min(W, W - dx)
            d = np.zeros((H, W), np.int64)
            d[ys:ye, xs:xe] = prev[ys:ye, xs:xe] - cur[ys + dy:ye + dy, xs + dx:xe + dx]
            blk = lambda a: a[:nbr * BS, :nbc * BS].reshape(nbr, BS, nbc, BS).sum((1, 3))    # noqa: E731
            cost[0][ci, ri] = np.where(ok, blk(np.abs(d)), BIG)
            cost[1][ci, ri] = np.where(ok, blk(d * d), BIG)
            l1 = np.zeros((nbr, nbc), np.int64)
            l2 = np.zeros((nbr, nbc), np.int64)
            for a in (0, 8):
                for b in (0, 8):
                    yy, xx = r0 + a, c0 + b
                    ya, xa = np.clip(yy + dy, 0, sc.shape[0] - 1), np.clip(xx + dx, 0, sc.shape[1] - 1)
                    dd = sp[yy, xx] - sc[ya, xa]
                    l1 += np.abs(dd)
                    l2 += dd * dd
            lb[0][ci, ri] = np.where(ok, l1, BIG)
            lb[1][ci, ri] = np.where(ok, l2 // 64, BIG)
    return cost, lb


tot = {}
t0 = time.time()
for p in range(n_pairs):
    prev, cur = frames[p].astype(np.int32), frames[p + 1].astype(np.int32)
    cost_all, lb_all = tables(prev, cur)
    nbr, nbc = cost_all[0].shape[2:]
    scan = (np.arange(NC)[:, None] * NC + np.arange(NC)[None, :])          # ci * NC + ri
    for pn in (0, 1):
        cost, lb = cost_all[pn], lb_all[pn]
        valid = cost < BIG
        key = np.where(valid, cost * 8192 + scan[:, :, None, None], BIG * 8192)       # the reference's order: first strict minimum
        lbkey = np.where(valid, lb * 8192 + scan[:, :, None, None], BIG * 8192)
        true_key = key.reshape(-1, nbr, nbc).min(0)
        # patches: 4 columns x R rows
        pshape = (NC // 4, 4, NC // R, R, nbr, nbc)
        p_lb = np.where(valid, lb, BIG).reshape(pshape).min((1, 3))                   # smallest bound of each patch
        p_first = (np.arange(NC // 4)[:, None] * 4 * NC + np.arange(NC // R)[None, :] * R)[:, :, None, None]
        p_floor = np.where(p_lb < BIG, p_lb * 8192 + p_first, BIG * 8192)             # no candidate of the patch has a smaller key
        p_valid = p_lb < BIG
        p_key = key.reshape(pshape).min((1, 3))                                      # best real key inside each patch
        n_patches = p_valid.sum()

        def surviving(ub_key):
            return ((p_floor < ub_key[None, None]) & p_valid).sum()

        flat_key = key.reshape(NC * NC, nbr, nbc)
        # P2
        i_minlb = lbkey.reshape(NC * NC, nbr, nbc).argmin(0)
        k_minlb = np.take_along_axis(flat_key, i_minlb[None], 0)[0]
        k_zero = key[SW, SW]
        ub2 = np.minimum(k_minlb, k_zero)
        # P3: winner of the block 4 to the left (final result), valid here?
        win_idx = flat_key.argmin(0)
        left = np.roll(win_idx, 4, axis=1)
        left[:, :4] = SW * NC + SW
        k_left = np.take_along_axis(flat_key, left[None], 0)[0]
        ub3 = np.minimum(ub2, k_left)
        # PP: whole patches around the three probes
        def patch_of(idx):
            ci, ri = idx // NC, idx % NC
            return ci // 4, ri // R
        pk = p_key.reshape(-1, nbr, nbc)
        def patch_key(idx):
            a, b = patch_of(idx)
            return np.take_along_axis(pk, (a * (NC // R) + b)[None], 0)[0]
        ubpp = np.minimum(np.minimum(patch_key(i_minlb), patch_key(np.full_like(i_minlb, SW * NC + SW))), patch_key(left))
        ubpp = np.minimum(ubpp, ub3)
        # TOPk
        order = np.argsort(p_floor.reshape(-1, nbr, nbc), axis=0, kind="stable")
        sorted_keys = np.take_along_axis(pk, order, 0)
        ubtop = {k: np.minimum(np.minimum.accumulate(sorted_keys, 0)[k - 1], ub3) for k in (1, 2, 4, 8, 16)}
        # NB: the P3 winners of the other blocks of the tile (2 x 4)
        idx3 = np.where(k_left <= np.minimum(k_minlb, k_zero), left, np.where(k_minlb <= k_zero, i_minlb, SW * NC + SW))
        ubnb = ub3.copy()
        for tr in range(0, nbr, 2):
            for tc in range(0, nbc, 4):
                cand = idx3[tr:tr + 2, tc:tc + 4].reshape(-1)
                for c in cand:
                    ubnb[tr:tr + 2, tc:tc + 4] = np.minimum(ubnb[tr:tr + 2, tc:tc + 4], flat_key[c, tr:tr + 2, tc:tc + 4])
        ubnbpp = np.minimum(ubnb, ubpp)
        res = {"P2": ub2, "P3": ub3, "PP": ubpp, "NB": ubnb, "NB+PP": ubnbpp, "TRUE": true_key}
        res.update({"TOP%d" % k: v for k, v in ubtop.items()})
        for name, ub in res.items():
            t = tot.setdefault((pn, name), [0, 0, 0])
            t[0] += surviving(ub); t[1] += n_patches; t[2] += (ub == true_key).sum()

        # BAND16: each of the 16 row bands (3 candidate rows: the patches one quad of phase B owns) offers its smallest-bound
        # patch; those that survive the first upper bound are scored first (no selection cost in the kernel).
        pf3 = p_floor.reshape(NC // 4, NC // R, nbr, nbc); pk3 = p_key.reshape(NC // 4, NC // R, nbr, nbc)
        sel = pf3.argmin(0)                                                       # per band: column group of its best patch
        bf = np.take_along_axis(pf3, sel[None], 0)[0]; bk = np.take_along_axis(pk3, sel[None], 0)[0]
        use = bf < ub3[None]
        ub_band = np.minimum(ub3, np.where(use, bk, BIG * 8192).min(0))
        is_sel = np.zeros(pf3.shape, bool); np.put_along_axis(is_sel, sel[None], use[None], 0)
        n_band = use.sum() + ((pf3 < ub_band[None, None]) & ~is_sel & p_valid.reshape(pf3.shape)).sum()
        t = tot.setdefault((pn, "BAND16"), [0, 0, 0]); t[0] += n_band; t[1] += n_patches; t[2] += (ub_band == true_key).sum()
        t = tot.setdefault((pn, "BAND16own"), [0, 0, 0]); t[0] += use.sum(); t[1] += n_patches

        # FRACf / FRACfa: no search for the threshold -- patches whose bound lies in the lowest fraction f of [smallest bound, UB]
        # are taken in the kernel's rank order (column group k, then lane = band * 4 + q), at most 16; "a": one count first,
        # and the fraction is halved once (twice) when more than 32 (64) patches lie below it.
        lbp = np.where(p_valid.reshape(pf3.shape), pf3 >> 13, BIG)                  # [column group 0..11][band][blocks]
        lo_b = lbp.reshape(-1, nbr, nbc).min(0); hi_b = (ub3 >> 13)
        # kernel order: k = cg % 3 outer, lane = band * 4 + cg // 3 inner
        cg = np.arange(NC // 4)[:, None]; band = np.arange(NC // R)[None, :]
        order_key = ((cg % R) * 64 + band * 4 + cg // R)[:, :, None, None] + np.zeros(pf3.shape, np.int64)
        surv3 = (pf3 < ub3[None, None]) & p_valid.reshape(pf3.shape)
        crowded = surv3.reshape(-1, nbr, nbc).sum(0) > 16
        for f, adapt in ((8, 0), (4, 0), (2, 0), (4, 1), (2, 1), (2, 2)):
            T = lo_b + (hi_b - lo_b) // f
            below = surv3 & (lbp <= T[None, None])
            if adapt:
                for _ in range(adapt):
                    cnt = below.reshape(-1, nbr, nbc).sum(0)
                    T = np.where(cnt > 32, lo_b + (T - lo_b) // 2, T)
                    below = surv3 & (lbp <= T[None, None])
            rk = np.where(below, order_key, 1 << 30).reshape(-1, nbr, nbc)
            kth = np.sort(rk, axis=0)[15]                                          # 16th smallest order key among those below
            take = below & (order_key <= kth[None, None]) & crowded[None, None]
            ubf = np.minimum(ub3, np.where(take, pk3, BIG * 8192).reshape(-1, nbr, nbc).min(0))
            n_f = take.sum() + ((pf3 < ubf[None, None]) & ~take & p_valid.reshape(pf3.shape)).sum()
            name = "FRAC1/%d%s" % (f, "a" * adapt)
            t = tot.setdefault((pn, name), [0, 0, 0]); t[0] += n_f; t[1] += n_patches; t[2] += (ubf == true_key).sum()
            t = tot.setdefault((pn, name + "own"), [0, 0, 0]); t[0] += take.sum(); t[1] += n_patches

        # two-round list (round 4 design): per block, patches whose bound is below a threshold found by ITER bisection steps on
        # [min bound, UB] (so that at most K pass) go first; the rest waits and is re-filtered with the tightened UB.
        pf = p_floor.reshape(-1, nbr, nbc); pv = p_valid.reshape(-1, nbr, nbc)
        for K in (4, 8, 12, 16, 24):
            for ITER in (3, 5):
                chunks_now = chunks_new = 0
                for tr in range(0, nbr, 2):
                    for tc in range(0, nbc, 4):
                        n_now = n1 = n2 = 0
                        for br in range(tr, min(tr + 2, nbr)):
                            for bc in range(tc, min(tc + 4, nbc)):
                                f = pf[:, br, bc][pv[:, br, bc]]; kk = pk[:, br, bc][pv[:, br, bc]]
                                ub = ub3[br, bc]
                                surv = f < ub
                                n_now += surv.sum()
                                if surv.sum() <= K:
                                    first_round = surv
                                else:
                                    lb_s = f[surv] >> 13
                                    lo, hi = int(lb_s.min()), int(ub >> 13) + 1
                                    for _ in range(ITER):
                                        mid = (lo + hi) // 2
                                        if (lb_s <= mid).sum() > K: hi = mid
                                        else: lo = mid
                                    first_round = surv & ((f >> 13) <= lo)
                                n1 += first_round.sum()
                                ub_new = min(ub, kk[first_round].min()) if first_round.any() else ub
                                n2 += (surv & ~first_round & (f < ub_new)).sum()
                        chunks_now += -(-n_now // 16)
                        chunks_new += -(-n1 // 16) + -(-n2 // 16)
                t = tot.setdefault((pn, "2R K%d I%d" % (K, ITER)), [0, 0])
                t[0] += chunks_now; t[1] += chunks_new
        tot.setdefault((pn, "blocks"), [0])[0] += nbr * nbc
    print("pair %d done, %.0f s" % (first + p, time.time() - t0), flush=True)

print("content %s, %d pair(s) from %d, 720x480 (or the content's size) bs 16 sw 16" % (kind, n_pairs, first))
for pn in (0, 1):
    nb = tot[(pn, "blocks")][0]
    for name in ("P2", "P3", "PP", "NB", "NB+PP", "TOP1", "TOP2", "TOP4", "TOP8", "TOP16", "BAND16", "BAND16own", "FRAC1/8", "FRAC1/8own", "FRAC1/4", "FRAC1/4own", "FRAC1/2", "FRAC1/2own", "FRAC1/4a", "FRAC1/4aown", "FRAC1/2a", "FRAC1/2aown", "FRAC1/2aa", "FRAC1/2aaown", "TRUE"):
        t = tot[(pn, name)]
        print("%s %-12s surviving patches %6.2f %%   blocks whose UB is the true minimum %5.1f %%" % (
            "MAE" if pn == 0 else "MSE", name, 100.0 * t[0] / t[1], 100.0 * t[2] / nb))
    for K in (4, 8, 12, 16, 24):
        for ITER in (3, 5):
            t = tot[(pn, "2R K%d I%d" % (K, ITER))]
            print("%s two rounds, quota %2d, %d bisection steps: wave-chunks of 16 patches per pair %7.1f -> %7.1f (%.3f)" % (
                "MAE" if pn == 0 else "MSE", K, ITER, t[0] / n_pairs, t[1] / n_pairs, t[1] / t[0]))
