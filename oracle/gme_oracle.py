"""CPU oracle (NumPy) for the per-frame-pair BBME + affine GME hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module; the product path (``global-motion-estimation_amd/``) never
does and fails loudly when the HIP library is missing.

Every function restates one function of the reference and cites it
(paths relative to ``/root/reference/global_motion_estimation/``).  The loop
structure -- one float32 block-distance evaluation per candidate, Python loops
over blocks and candidates -- follows the reference on purpose: timed on the
host it stands for "the reference NumPy CPU path" (BASELINE.md §3).

Pinning: this oracle is checked bit-for-bit against golden vectors generated
by importing the real reference in the build container
(``oracle/refimport/make_golden.py`` -> ``tests/golden/*.npz``).  One piece is
PARITY UNPINNED: ``pyr_down`` restates ``cv2.pyrDown`` (opencv-python
4.5.5.62, utils.py:48) from OpenCV's documented algorithm because OpenCV is
not available to check against; pyramid levels 0 and 1 depend on it, level 2
(full resolution) does not.
"""
import cmath
import itertools

import numpy as np

BBME_BLOCK_SIZE = 16          # motion.py:9
OUTLIER_FRACTION = 0.3        # motion.py:10
_INF = np.float32(np.inf)


# --------------------------------------------------------------------------
# block distance -- bbme.py:41-94
# --------------------------------------------------------------------------
def block_cost(block_a, block_b, pnorm):
    """float32 sum of |a-b| (pnorm 0) or (a-b)^2 (pnorm 1); bbme.py:41-64,79,94."""
    assert block_a.shape == block_b.shape
    if pnorm not in (0, 1):
        raise IndexError("list index out of range")      # bbme.py:60
    d = np.array(block_a, dtype=np.float32) - np.array(block_b, dtype=np.float32)
    return np.sum(np.abs(d)) if pnorm == 0 else np.sum(d * d)


def _block_origins(height, width, bs):
    # bbme.py:133-136 (same ranges in all four searches)
    return itertools.product(range(0, height - (bs - 1), bs),
                             range(0, width - (bs - 1), bs))


def _inside(top, left, bs, height, width):
    # bbme.py:157-162: the candidate must lie entirely inside the frame
    return top >= 0 and left >= 0 and top + bs - 1 <= height - 1 and left + bs - 1 <= width - 1


# --------------------------------------------------------------------------
# exhaustive search -- bbme.py:105-179
# --------------------------------------------------------------------------
def search_exhaustive(previous, current, mf, height, width, pnorm, bs, sw, block_rows=None):
    """`block_rows=(lo, hi)` restricts the block loop to those block rows (bench.py's
    bounded CPU sample); everything else is the full-frame search."""
    span = range(-sw, sw + bs)            # asymmetric window, bbme.py:146-149
    for r0, c0 in _block_origins(height, width, bs):
        if block_rows is not None and not block_rows[0] <= r0 // bs < block_rows[1]:
            continue
        anchor = previous[r0:r0 + bs, c0:c0 + bs]
        best = _INF
        best_col = best_row = 0
        for wc in span:                   # column offset is the OUTER loop
            for wr in span:
                top, left = r0 + wr, c0 + wc
                if not _inside(top, left, bs, height, width):
                    continue
                cost = block_cost(current[top:top + bs, left:left + bs], anchor, pnorm)
                if cost < best:           # strict: first minimum wins (bbme.py:171)
                    best, best_col, best_row = cost, wc, wr
        mf[r0 // bs, c0 // bs, 0] = best_col
        mf[r0 // bs, c0 // bs, 1] = best_row
    return mf


# --------------------------------------------------------------------------
# three-step search -- bbme.py:182-341
# --------------------------------------------------------------------------
def search_threestep(previous, current, mf, height, width, pnorm, bs, sw):
    steps = (int((2 * sw + bs) / 3), int((2 * sw + bs) / 5), int((2 * sw + bs) / 10))
    # bbme.py:215 -- initialised once, before the block loop, and carried over
    d_row = d_col = t_row = t_col = 0
    for r0, c0 in _block_origins(height, width, bs):
        anchor = previous[r0:r0 + bs, c0:c0 + bs]

        def scan(org_r, org_c, s, keep_r, keep_c):
            best = _INF
            for wc in (-s, 0, s):
                for wr in (-s, 0, s):
                    top, left = org_r + wr, org_c + wc
                    if not _inside(top, left, bs, height, width):
                        continue
                    cost = block_cost(current[top:top + bs, left:left + bs], anchor, pnorm)
                    if cost < best:
                        best, keep_r, keep_c = cost, wr, wc
            return keep_r, keep_c

        d_row, d_col = scan(r0, c0, steps[0], d_row, d_col)          # bbme.py:229-257
        r2, c2 = r0 + d_row, c0 + d_col                                # bbme.py:260-261
        t_row, t_col = scan(r2, c2, steps[1], t_row, t_col)          # bbme.py:267-294
        d_row += t_row
        d_col += t_col
        # bbme.py:300-301: origin 3 adds the already accumulated displacement again
        r3, c3 = r2 + d_row, c2 + d_col
        t_row, t_col = scan(r3, c3, steps[2], t_row, t_col)          # stale t_* if none valid
        d_row += t_row
        d_col += t_col
        mf[r0 // bs, c0 // bs, 0] = d_col
        mf[r0 // bs, c0 // bs, 1] = d_row
    return mf


# --------------------------------------------------------------------------
# 2-D logarithmic search -- bbme.py:344-433
# --------------------------------------------------------------------------
def search_twodlog(previous, current, mf, height, width, pnorm, bs, sw):
    for r0, c0 in _block_origins(height, width, bs):
        anchor = previous[r0:r0 + bs, c0:c0 + bs]
        best_r = best_c = 0               # bbme.py:371 (absolute coordinates later)
        pos_r, pos_c = r0, c0
        step = sw
        while step > 1:
            if step > 2:                  # cross, centre first (bbme.py:387-393)
                cand = [(pos_r, pos_c), (pos_r + step, pos_c), (pos_r - step, pos_c),
                        (pos_r, pos_c + step), (pos_r, pos_c - step)]
            else:                         # 3x3 ring with stride 2, centre is 5th
                cand = list(itertools.product((pos_r - 2, pos_r, pos_r + 2),
                                              (pos_c - 2, pos_c, pos_c + 2)))
            best = _INF
            for top, left in cand:
                if not _inside(top, left, bs, height, width):
                    continue
                cost = block_cost(current[top:top + bs, left:left + bs], anchor, pnorm)
                if cost < best:
                    best, best_r, best_c = cost, top, left
            # bbme.py:423 -- ``a and b or c`` precedence
            if (best_r == pos_r and best_c == pos_c) or step == 2:
                step //= 2
            pos_r, pos_c = best_r, best_c
        mf[r0 // bs, c0 // bs, 1] = best_r - r0
        mf[r0 // bs, c0 // bs, 0] = best_c - c0
    return mf


# --------------------------------------------------------------------------
# diamond search -- bbme.py:436-534
# --------------------------------------------------------------------------
LDSP = ((0, 0), (2, 0), (1, 1), (0, 2), (-1, 1), (-2, 0), (-1, -1), (0, -2), (1, -1))
SDSP = ((0, 0), (1, 0), (0, 1), (-1, 0), (0, -1))


def search_diamond(previous, current, mf, height, width, pnorm, bs, sw=-1):
    max_r, max_c = height - bs - 1, width - bs - 1       # off-by-one clamp, bbme.py:503-504
    for r0, c0 in _block_origins(height, width, bs):
        anchor = previous[r0:r0 + bs, c0:c0 + bs]
        centre = (r0, c0)
        while True:                       # large pattern until the centre wins
            best, best_pos = _INF, centre
            for o_r, o_c in LDSP:
                rr = min(max(centre[0] + o_r, 0), max_r)
                cc = min(max(centre[1] + o_c, 0), max_c)
                cost = block_cost(anchor, current[rr:rr + bs, cc:cc + bs], pnorm)
                if cost < best:
                    best, best_pos = cost, (rr, cc)
            done = best_pos == centre
            centre = best_pos
            if done:
                break
        best = _INF
        for o_a, o_b in SDSP:             # small pattern, offsets applied swapped (bbme.py:518-521)
            rr = min(max(centre[0] + o_b, 0), max_r)
            cc = min(max(centre[1] + o_a, 0), max_c)
            cost = block_cost(anchor, current[rr:rr + bs, cc:cc + bs], pnorm)
            if cost < best:
                best, best_pos = cost, (rr, cc)
        mf[r0 // bs, c0 // bs, 1] = best_pos[0] - r0
        mf[r0 // bs, c0 // bs, 0] = best_pos[1] - c0
    return mf


SEARCHES = (search_exhaustive, search_threestep, search_twodlog, search_diamond)   # bbme.py:609-614


def get_motion_field(previous, current, block_size=4, search_window=2,
                     searching_procedure=1, pnorm_distance=1):
    """bbme.py:12-38 (defaults included: three-step, MSE)."""
    height, width = previous.shape[0], previous.shape[1]
    mf = np.zeros((int(height / block_size), int(width / block_size), 2), dtype=np.int32)
    return SEARCHES[searching_procedure](previous, current, mf, height, width,
                                         pnorm_distance, block_size, search_window)


# --------------------------------------------------------------------------
# pyramids -- utils.py:34-51 around cv2.pyrDown (PARITY UNPINNED, see header)
# --------------------------------------------------------------------------
def _reflect101(p, n):
    if n == 1:
        return 0
    while p < 0 or p >= n:
        p = -p if p < 0 else 2 * (n - 1) - p
    return p


def pyr_down(src):
    src = np.asarray(src, dtype=np.uint8)
    h, w = src.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    taps = (1, 4, 6, 4, 1)
    s = src.astype(np.int64)
    cols = [[_reflect101(2 * x + d, w) for x in range(dw)] for d in range(-2, 3)]
    rows = [[_reflect101(2 * y + d, h) for y in range(dh)] for d in range(-2, 3)]
    hp = sum(taps[i] * s[:, cols[i]] for i in range(5))
    vp = sum(taps[i] * hp[rows[i], :] for i in range(5))
    return ((vp + 128) >> 8).astype(np.uint8)


def get_pyramids(image, levels=3):
    """Coarse level first (utils.py:45-51)."""
    pyr = [image]
    for _ in range(1, levels):
        pyr.insert(0, pyr_down(pyr[0]))
    return pyr


# --------------------------------------------------------------------------
# affine model, robust fit, compensation -- motion.py
# --------------------------------------------------------------------------
def affine_field(shape, params):
    """motion.py:91-105,139-157: int16 field of round-half-even(A @ p), raw block indices."""
    p = np.asarray(params).astype(np.float64)      # int32 @ float32 promotes to float64
    out = np.zeros((shape[0], shape[1], 2), dtype=np.int16)
    for i in range(shape[0]):
        for j in range(shape[1]):
            # np.matmul(A, p) goes through BLAS gemv, whose summation order is build
            # dependent; the order below is the one the reference shows under its
            # NumPy 1.x/OpenBLAS in the build container (20000/20000 trap cases,
            # tests/golden/g6_edges.npz): (p0 + p2*j) + p1*i.  It only matters when a
            # displacement sits within one ulp of k + 0.5.
            dx = (p[0] + p[2] * j) + p[1] * i
            dy = (p[3] + p[5] * j) + p[4] * i
            # Python round() on float64 = round-half-even; int16 store wraps
            out[i, j, 0] = np.int64(np.rint(dx)).astype(np.int16)
            out[i, j, 1] = np.int64(np.rint(dy)).astype(np.int16)
    return out


def first_parameters(dense_mf):
    """motion.py:176-188."""
    return np.array([np.mean(dense_mf[:, :, 0]), 0.0, 0.0,
                     np.mean(dense_mf[:, :, 1]), 0.0, 0.0], dtype=np.float32)


def project_parameters(params):
    """motion.py:191-207 -- in place."""
    params[0] = params[0] * 2
    params[3] = params[3] * 2
    return params


def outlier_mask(gt, model, fraction=OUTLIER_FRACTION):
    """motion.py:236-244 -> (diff int, threshold, mask bool)."""
    diff = np.abs(gt.astype(np.int32) - model.astype(np.int32)).sum(axis=2)
    ordered = np.sort(diff.flatten())
    k = int(fraction * len(ordered))
    thr = ordered[-k]                      # k == 0 -> ordered[0]
    return diff, int(thr), diff > thr


def normal_sums(gt, mask, level_shape):
    """motion.py:248-261,266-279: sequential float64 sums over inliers.

    Returns F (3x3, identical for both passes), Sx (3), Sy (3).
    """
    w = 1 / (level_shape[0] * level_shape[1])
    F = np.zeros((3, 3), dtype=np.float64)
    Sx = np.zeros(3, dtype=np.float64)
    Sy = np.zeros(3, dtype=np.float64)
    for i in range(gt.shape[0]):
        for j in range(gt.shape[1]):
            if mask[i, j]:
                continue
            v = (1.0, float(i * 4), float(j * 4))        # literal 4, motion.py:254-255
            for a in range(3):
                for b in range(3):
                    F[a, b] += (v[a] * v[b]) * w
                Sx[a] += (v[a] * float(gt[i, j, 0])) * w
                Sy[a] += (v[a] * float(gt[i, j, 1])) * w
    return F, Sx, Sy


def solve_parameters(F, Sx, Sy):
    """motion.py:262-264,280-286: inv(np.matrix(F)) @ S for both passes."""
    finv = np.array(np.linalg.inv(np.matrix(F)))
    ax = np.matmul(finv, Sx.reshape(3, 1)).reshape(3)
    finv = np.array(np.linalg.inv(np.matrix(F)))
    ay = np.matmul(finv, Sy.reshape(3, 1)).reshape(3)
    return np.concatenate([ax, ay])


def robust_fit(previous, current, old_params, block_size=None, fraction=None,
               procedure=3, search_window=2, stages=None):
    """motion.py:210-286.  ``procedure``/``search_window`` default to the
    reference's hard-coded diamond search; other values serve BASELINE config 4
    (SURVEY.md §0 D9)."""
    bs = BBME_BLOCK_SIZE if block_size is None else block_size
    fr = OUTLIER_FRACTION if fraction is None else fraction
    gt = get_motion_field(previous, current, block_size=bs, search_window=search_window,
                          searching_procedure=procedure)
    model = affine_field(gt.shape, old_params)
    diff, thr, mask = outlier_mask(gt, model, fr)
    F, Sx, Sy = normal_sums(gt, mask, previous.shape)
    params = solve_parameters(F, Sx, Sy)
    if stages is not None:
        stages.append(dict(gt=gt, model=model, diff=diff, thr=thr, mask=mask,
                           F=F, Sx=Sx, Sy=Sy, params_in=np.array(old_params, copy=True),
                           params=params))
    return params


def global_motion_estimation(previous, current, stages=None, block_size=None, fraction=None):
    """motion.py:109-136."""
    pp, cp = get_pyramids(previous), get_pyramids(current)
    dense = get_motion_field(pp[0], cp[0], block_size=2, searching_procedure=3)   # motion.py:27-29
    params = first_parameters(dense)
    if stages is not None:
        stages.append(dict(dense=dense, params0=params.copy(), pyr_prev=pp, pyr_cur=cp))
    for lvl in range(1, len(pp)):
        params = project_parameters(params)
        params = robust_fit(pp[lvl], cp[lvl], params, block_size=block_size,
                            fraction=fraction, stages=stages)
    return params


def compensate_frame(frame, mf):
    """motion.py:289-321 as a per-block gather (same result as the per-pixel loop)."""
    out = np.copy(frame)
    H, W = frame.shape
    bs = H // mf.shape[0]
    for i in range(mf.shape[0]):
        for j in range(mf.shape[1]):
            d0, d1 = int(mf[i, j, 0]), int(mf[i, j, 1])
            for a in range(i * bs, (i + 1) * bs):
                for b in range(j * bs, (j + 1) * bs):
                    sa, sb = a - d1, b - d0
                    if a < H and b < W and 0 <= sa < H and 0 <= sb < W:
                        out[a, b] = frame[sa, sb]
    return out


def motion_compensation(previous, current):
    """motion.py:324-341."""
    params = global_motion_estimation(previous, current)
    shape = (previous.shape[0] // BBME_BLOCK_SIZE, previous.shape[1] // BBME_BLOCK_SIZE)
    return compensate_frame(previous, affine_field(shape, params))


def psnr(original, noisy):
    """utils.py:100-116, real part (the reference returns a cmath complex)."""
    mse = np.mean((original.astype("int") - noisy.astype("int")) ** 2)
    if mse == 0:
        return -1
    return (20 * cmath.log10(255.0 / cmath.sqrt(mse))).real
