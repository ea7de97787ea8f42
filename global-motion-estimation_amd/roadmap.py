"""The authors' roadmap (``recap_future_updates.md:9-14``), built on the same device pipeline.

EXTENSIONS -- nothing here has a reference implementation to be bit-compared with (SURVEY.md §8(f)4:
"other motion models, auto-selection of bs / sw / outlier fraction ... change results").  The default
path (``motion.global_motion_estimation``, affine model, fixed constants) is untouched; these helpers
are opt-in and tested for self-consistency (``tests/test_gpu_round2.py``).

1. **Other motion models.**  The device fit stage (``k_fit_level``, motion.py:232-279) leaves the
   weighted normal-equation sums ``F = sum [1 x y]^T [1 x y]``, ``Sx = sum [1 x y] dx``,
   ``Sy = sum [1 x y] dy`` over the inlier blocks.  Those sums also determine the least-squares fit of
   every model that is linear in ``[1, x, y]``; only the small host solve differs:

   * ``affine``       6 parameters -- the reference (two 3x3 systems, motion.py:262-282)
   * ``translation``  2 parameters -- ``dx = a0, dy = b0``
   * ``similarity``   4 parameters -- zoom ``s`` + rotation ``r`` + shift.  In the reference's convention ``x`` is the
     ROW coordinate and ``dx`` the COLUMN displacement (motion.py:254-259, bbme.py:176-177), so the model reads
     ``dx = a0 - r x + s y``, ``dy = b0 + s x + r y``

   Every model is returned in the affine layout ``[a0, a1, a2, b0, b1, b2]`` so the model field, the
   outlier mask of the next level and the compensation run unchanged -- including the reference's own
   coordinate mismatch: the fit sees ``x = 4 i, y = 4 j`` (motion.py:254-255) while the model field is
   evaluated at the raw block indices (motion.py:139-157), so the linear terms of ANY model act with a
   quarter of their fitted strength downstream, exactly as in the reference's affine path.
2. **Parameter heuristics** (``suggest_parameters``): block size from the frame height (the authors'
   slide settings, docs/presentation/main.tex:382-558, follow ``H / 20`` in 4 of 5 cases), search window
   from the dense coarse field, outlier fraction from the spread of the block vectors.
3. **One CLI** for all scripts: ``gme_cli.py``.
"""
import numpy as np

import motion

MODELS = ("affine", "translation", "similarity")


def solve_model(sums, model="affine"):
    """float64[P, 15] normal-equation sums (F | Sx | Sy) -> float64[P, 6] parameters in affine layout."""
    sums = np.asarray(sums, dtype=np.float64).reshape(-1, 15)
    if model == "affine":
        return motion._solve_batch(sums)
    F = sums[:, :9].reshape(-1, 3, 3)
    sx, sy = sums[:, 9:12], sums[:, 12:15]
    out = np.zeros((len(sums), 6))
    if model == "translation":
        n = F[:, 0, 0]
        if np.any(n == 0):
            raise np.linalg.LinAlgError("Singular matrix")
        out[:, 0] = sx[:, 0] / n
        out[:, 3] = sy[:, 0] / n
        return out
    if model == "similarity":
        n, mx, my = F[:, 0, 0], F[:, 0, 1], F[:, 0, 2]
        q = F[:, 1, 1] + F[:, 2, 2]
        # unknowns (a0, b0, s, r); rows [1, 0, y, -x] for dx and [0, 1, x, y] for dy
        N = np.zeros((len(sums), 4, 4))
        N[:, 0, 0] = n; N[:, 0, 2] = my; N[:, 0, 3] = -mx
        N[:, 1, 1] = n; N[:, 1, 2] = mx; N[:, 1, 3] = my
        N[:, 2, 0] = my; N[:, 2, 1] = mx; N[:, 2, 2] = q
        N[:, 3, 0] = -mx; N[:, 3, 1] = my; N[:, 3, 3] = q
        rhs = np.stack([sx[:, 0], sy[:, 0], sx[:, 2] + sy[:, 1], sy[:, 2] - sx[:, 1]], axis=1)
        th = np.linalg.solve(N, rhs[:, :, None])[:, :, 0]            # LinAlgError on a singular system
        a0, b0, zoom, rot = th[:, 0], th[:, 1], th[:, 2], th[:, 3]
        out[:, 0], out[:, 1], out[:, 2] = a0, -rot, zoom
        out[:, 3], out[:, 4], out[:, 5] = b0, zoom, rot
        return out
    raise ValueError("unknown motion model %r (choose from %r)" % (model, MODELS))


def estimate_sequence(seq, frame_distance=1, model="affine", procedure=3, search_window=2):
    """motion.estimate_sequence with a selectable motion model -> float64[P, 6] (affine layout)."""
    if getattr(seq, "_split", False):
        raise RuntimeError("estimate_sequence needs blocking calls: the sequence is in split-phase mode")
    frac = float(motion.MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE)
    _, sums = seq.gme_begin_fit(frame_distance, int(motion.BBME_BLOCK_SIZE), frac, procedure, search_window)
    params = solve_model(sums, model)
    params[:, 0] = params[:, 0] * 2
    params[:, 3] = params[:, 3] * 2
    return solve_model(seq.gme_fit(2, params, frac), model)


def global_motion_estimation(previous, current, model="affine"):
    """motion.global_motion_estimation (motion.py:109-136) with a selectable model."""
    return estimate_sequence(motion._pair_sequence(previous, current), 1, model)[0]      # the cached two-frame sequence of motion.py


def suggest_parameters(previous, current):
    """Heuristics for the constants the authors tuned by hand per video (recap_future_updates.md:4-8):

    * ``block_size``     frame height / 20, to a multiple of 4 in [8, 32]
    * ``search_window``  covers the 95th percentile of the dense coarse field (pyramid level 0,
                         motion.py:13-30) scaled to full resolution, multiple of 4 in [4, 32]
    * ``outlier_fraction``  share of level-2 block vectors further than 2 px (L1) from the translational
                         fit of the rest, plus a margin, in [0.1, 0.5] (motion.py:10 uses a fixed 0.3)
    """
    import bbme
    import utils
    H, W = previous.shape
    bs = int(min(32, max(8, 4 * round(H / 20.0 / 4.0))))
    pyr_p, pyr_c = utils.get_pyramids(previous), utils.get_pyramids(current)
    dense = motion.dense_motion_estimation(pyr_p[0], pyr_c[0]).reshape(-1, 2)
    reach = 4.0 * np.percentile(np.abs(dense).max(axis=1), 95) if len(dense) else 0.0
    sw = int(min(32, max(4, 4 * int(np.ceil((reach + 2.0) / 4.0)))))
    field = bbme.get_motion_field(previous, current, block_size=bs, searching_procedure=3).reshape(-1, 2)
    frac = 0.3
    if len(field):
        med = np.median(field, axis=0)
        far = np.abs(field - med).sum(axis=1) > 2
        frac = float(min(0.5, max(0.1, far.mean() + 0.05)))
    return {"block_size": bs, "search_window": sw, "outlier_fraction": round(frac, 3)}
