"""Global motion estimation on MI355X -- drop-in for the reference's ``motion`` module.

Same call surface as ``global_motion_estimation/motion.py``; pyramids, block matching,
model fields, outlier masks, normal-equation sums, compensation run on the GPU
(``csrc/*.hip`` through ``_gme_native``).  The only arithmetic left on the host is what
the reference itself hands to NumPy/LAPACK per level: two 3x3 ``np.linalg.inv`` + matmul
(motion.py:262-264,280-282) -- kept here so the parameters round exactly like the
NumPy installed next to this module.

``BBME_BLOCK_SIZE`` and ``MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE`` are read at call
time, so the authors' habit of patching them (12/24/32 for their figures) keeps working.
"""
import numpy as np

import _gme_native as _native
from bbme import get_motion_field

BBME_BLOCK_SIZE = 16                               # motion.py:9
MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE = .3      # motion.py:10


def dense_motion_estimation(previous, current):
    """motion.py:13-30: diamond search on 2x2 blocks (MSE by default)."""
    return get_motion_field(previous, current, block_size=2, searching_procedure=3)


def _solve(sums):
    """motion.py:262-264,280-286 for one pair: [a0,a1,a2,b0,b1,b2] from F | Sx | Sy."""
    F = sums[:9].reshape(3, 3)
    finv = np.array(np.linalg.inv(F))              # LinAlgError on a singular system, as upstream
    ax = np.matmul(finv, sums[9:12].reshape(3, 1)).reshape((3,))
    finv = np.array(np.linalg.inv(F))
    ay = np.matmul(finv, sums[12:15].reshape(3, 1)).reshape((3,))
    return np.concatenate([ax, ay])


def _solve_batch(sums):
    """`_solve` for float64[P, 15]: stacked ``inv``/``matmul`` run the same LAPACK/BLAS routine
    per matrix, so the rows equal the per-pair calls bit for bit (tests/test_host.py checks).
    (0.7-0.9 ms per 2048 pairs on one host core, nearly all of it ``inv``'s per-matrix LAPACK calls;
    cutting the batch over host threads gained nothing -- several streams per GPU hide it instead.)"""
    sums = np.asarray(sums, dtype=np.float64)
    if len(sums) == 0:
        return np.zeros((0, 6))
    finv = np.linalg.inv(sums[:, :9].reshape(-1, 3, 3))
    ax = np.matmul(finv, sums[:, 9:12, None])[:, :, 0]
    ay = np.matmul(finv, sums[:, 12:15, None])[:, :, 0]
    return np.concatenate([ax, ay], axis=1)


_pair_cache = __import__("threading").local()


def _pair_sequence(previous, current):
    """A two-frame device sequence holding (previous, current) for the single-pair functions.  It is kept per
    thread and per frame shape: allocating the planes, pyramids and stage buffers anew for every call cost more
    than the kernels of a small pair (the reference's per-pair functions are called in loops, results.py:41-59)."""
    previous, current = _native.as_frame(previous, "previous"), _native.as_frame(current, "current")
    if previous.shape != current.shape:
        raise AssertionError("previous and current differ in shape (bbme.py:59)")
    ctx = _native.default_context()
    ent = getattr(_pair_cache, "entry", None)
    if ent is None or ent[0] != previous.shape or ent[1].handle is None or ent[1].ctx is not ctx:
        if ent is not None:
            ent[1].close()
        ent = (previous.shape, _native.Sequence(ctx, 2, previous.shape[0], previous.shape[1]))
        _pair_cache.entry = ent
    seq = ent[1]
    seq.upload(0, np.stack([previous, current]))
    return seq


def _fit_pair(previous, current, old_parameters, fraction):
    seq = _pair_sequence(previous, current)
    seq.bbme(1, int(BBME_BLOCK_SIZE), 2, 3, 1)                         # motion.py:224-229 (diamond, MSE)
    p = np.asarray(old_parameters).astype(np.float64).reshape(1, 6)
    return _solve(seq.gme_fit(-1, p, fraction)[0])


def best_affine_parameters(previous, current):
    """motion.py:33-88: the unmasked least-squares fit (no live caller upstream)."""
    return _fit_pair(previous, current, np.zeros(6), -1.0)


def best_affine_parameters_robust(previous, current, old_parameters):
    """motion.py:210-286: BBME field, model field from `old_parameters`, top ~30 % of L1
    deviations masked, weighted normal equations over the inliers, 3x3 solve."""
    return _fit_pair(previous, current, old_parameters, float(MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE))


def affine_model(x, y, parameters):
    """motion.py:91-105 (host helper; the device evaluates the same expression per block)."""
    A = np.asarray([[1, x, y, 0, 0, 0], [0, 0, 0, 1, x, y]], dtype=np.int32)
    return np.matmul(A, np.transpose(parameters))


def get_motion_field_affine(shape, parameters):
    """motion.py:139-157: int16[shape[0], shape[1], 2] of round-half-even displacements."""
    return _native.default_context().affine_field(parameters, int(shape[0]), int(shape[1]))


def compute_first_parameters(dense_motion_field):
    """motion.py:176-188."""
    a0 = np.mean(dense_motion_field[:, :, 0])
    b0 = np.mean(dense_motion_field[:, :, 1])
    return np.array([a0, 0.0, 0.0, b0, 0.0, 0.0], dtype=np.float32)


def first_parameter_estimation(previous, current):
    """motion.py:160-173."""
    return compute_first_parameters(dense_motion_estimation(previous, current))


def parameter_projection(parameters):
    """motion.py:191-207 -- in place."""
    parameters[0] = parameters[0] * 2
    parameters[3] = parameters[3] * 2
    return parameters


def estimate_sequence(seq, frame_distance=1, procedure=3, search_window=2):
    """motion.global_motion_estimation for every pair of a device-resident sequence.

    Returns float64[P, 6].  Three device phases (begin, fit level 1, fit level 2) with the
    per-pair 3x3 solves on the host in between, exactly the order of motion.py:123-136.
    """
    if getattr(seq, "_split", False):
        raise RuntimeError("estimate_sequence needs blocking calls: the sequence is in split-phase mode "
                           "(set_split_phase(False) first, or drive it with wait() like sequence._interleaved)")
    frac = float(MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE)
    # first parameters -> projection -> level-1 fit stay on the device (gme_seq_gme_begin_fit); the host solves level 1,
    # projects (in float64, the solution's dtype) and asks for level 2
    _, sums = seq.gme_begin_fit(frame_distance, int(BBME_BLOCK_SIZE), frac, procedure, search_window)
    params = _solve_batch(sums)
    params[:, 0] = params[:, 0] * 2                # parameter_projection
    params[:, 3] = params[:, 3] * 2
    return _solve_batch(seq.gme_fit(2, params, frac))


def global_motion_estimation(previous, current):
    """motion.py:109-136 -> float64[6] = [a0, a1, a2, b0, b1, b2] at full resolution."""
    return estimate_sequence(_pair_sequence(previous, current), 1)[0]


def compensate_frame(frame, motion_field):
    """motion.py:289-321: block-wise gather ``out[a, b] = frame[a - d[1], b - d[0]]`` where the
    source lies inside the frame, else the pixel is kept."""
    return _native.default_context().compensate(frame, motion_field)


def motion_compensation(previous, current):
    """motion.py:324-341."""
    parameters = global_motion_estimation(previous, current)
    shape = (previous.shape[0] // BBME_BLOCK_SIZE, previous.shape[1] // BBME_BLOCK_SIZE)
    motion_field = get_motion_field_affine(shape, parameters)
    return compensate_frame(previous, motion_field)
