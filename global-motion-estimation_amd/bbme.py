"""Block-based motion estimation on MI355X -- drop-in for the reference's ``bbme`` module.

Same names, defaults, dtypes and shapes as ``global_motion_estimation/bbme.py``
(reference lines cited per function), but the search runs in hand-written HIP kernels
(``csrc/bbme_fast.hip``, ``csrc/bbme_kernels.hip``) reached through ctypes
(``_gme_native``).  There is no CPU fallback: without ``libgme_hip.so`` and a gfx950
device these functions raise.

Motion-field layout (bbme.py:176-177): ``mf[i, j, 0]`` is the column (x) and
``mf[i, j, 1]`` the row (y) displacement of block ``(i, j)``, position in ``current``
minus position in ``previous``.
"""
import numpy as np

import _gme_native as _native

EXHAUSTIVE, THREESTEP, TWODLOG, DIAMOND = 0, 1, 2, 3
MAE, MSE = 0, 1


def _table_index(index, size, what):
    # the reference indexes Python lists (bbme.py:27,60): negative indices wrap, others raise
    index = int(index)
    if not -size <= index < size:
        raise IndexError("list index out of range (%s %d)" % (what, index))
    return index % size


def get_motion_field(previous, current, block_size=4, search_window=2,
                     searching_procedure=1, pnorm_distance=1) -> np.ndarray:
    """bbme.py:12-38.  Defaults are the reference's: three-step search, MSE.

    Returns int32[H // bs, W // bs, 2].
    """
    procedure = _table_index(searching_procedure, 4, "searching_procedure")
    pnorm = _table_index(pnorm_distance, 2, "pnorm_distance")
    return _native.default_context().bbme(previous, current, int(block_size), int(search_window),
                                          procedure, pnorm)


# -- block distances (bbme.py:41-94).  Host-side helpers kept for API completeness: the
# -- device searches never call them (they compute the same integers in registers).
def mae(diff_block):
    """bbme.py:67-79."""
    return np.sum(np.abs(diff_block))


def mse(diff_block):
    """bbme.py:82-94."""
    return np.sum(diff_block * diff_block)


pnorm_distances = [mae, mse]


def compute_dfd(block_1, block_2, pnorm_index=0):
    """bbme.py:41-64: float32 sum of |a-b| (index 0) or (a-b)^2 (index 1)."""
    assert block_1.shape == block_2.shape
    pnorm = pnorm_distances[pnorm_index]
    return pnorm(np.array(block_1, dtype=np.float32) - np.array(block_2, dtype=np.float32))


def _search(procedure):
    def run(previous, current, mf, height, width, pnorm_distance=0, block_size=4, search_window=2):
        pnorm = _table_index(pnorm_distance, 2, "pnorm_distance")
        field = _native.default_context().bbme(np.asarray(previous)[:height, :width],
                                               np.asarray(current)[:height, :width],
                                               int(block_size), int(search_window), procedure, pnorm)
        mf[:field.shape[0], :field.shape[1], :] = field      # filled in place and returned
        return mf
    return run


exhaustive_search = _search(EXHAUSTIVE)      # bbme.py:105-179
exhaustive_search.__doc__ = "bbme.py:105-179: window offsets range(-sw, sw + bs) on both axes, column outer."
threestep_search = _search(THREESTEP)        # bbme.py:182-341
threestep_search.__doc__ = "bbme.py:182-341."
twodlog_search = _search(TWODLOG)            # bbme.py:344-433
twodlog_search.__doc__ = "bbme.py:344-433."
diamond_search = _search(DIAMOND)            # bbme.py:436-534
diamond_search.__doc__ = "bbme.py:436-534 (search_window is ignored there too)."

searching_procedures = [exhaustive_search, threestep_search, twodlog_search, diamond_search]
