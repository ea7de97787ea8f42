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


def rescale_motion_field(motion_field, scale=2):
    """bbme.py:537-546: nearest-neighbour upsampling by `scale`, stored as int32 (float input is
    truncated toward zero on the store), then multiplied by 2 -- always 2, whatever `scale` is."""
    mf = np.zeros((motion_field.shape[0] * scale, motion_field.shape[1] * scale, 2), dtype=np.int32)
    mf[...] = np.repeat(np.repeat(np.asarray(motion_field)[:, :, :2], scale, axis=0), scale, axis=1)
    return mf * 2


def hierarchical_wrapper(previous, current, block_size=10, search_window=4, searching_procedure=3):
    """bbme.py:549-605: coarse-to-fine BBME over the 3-level pyramid.  The coarsest level uses
    `searching_procedure`, the finer ones always diamond; each level's field is averaged with the
    rescaled field of the level above, so the result is float64."""
    from utils import get_pyramids
    previous_pyr = get_pyramids(previous, levels=3)
    current_pyr = get_pyramids(current, levels=3)
    motion_field = get_motion_field(previous_pyr[0], current_pyr[0], block_size=block_size,
                                    searching_procedure=searching_procedure, search_window=search_window)
    for level in range(1, len(previous_pyr)):
        motion_field = rescale_motion_field(motion_field, scale=2)
        new_mf = get_motion_field(previous_pyr[level], current_pyr[level], block_size=block_size,
                                  searching_procedure=3, search_window=search_window)
        if motion_field.shape != new_mf.shape:          # bbme.py:596-602: pad ONE axis by one line
            if motion_field.shape[0] != new_mf.shape[0]:
                motion_field = np.vstack([motion_field, np.zeros((1, motion_field.shape[1], 2), dtype=np.int32)])
            else:
                motion_field = np.hstack([motion_field, np.zeros((motion_field.shape[0], 1, 2), dtype=np.int32)])
        motion_field = (motion_field + new_mf) / 2
    return motion_field


def main(args):
    """bbme.py:617-649: field + hierarchical field between frames fi-3 and fi, drawn and saved.
    As upstream, ``-pn`` is parsed but not forwarded (the norm stays MSE)."""
    import os
    from utils import draw_motion_field, get_video_frames, write_image
    frames = get_video_frames(args.path)
    previous, current = frames[args.fi - 3], frames[args.fi]
    motion_field = get_motion_field(previous, current, block_size=args.block_size,
                                    searching_procedure=args.searching_procedure, search_window=args.search_window)
    motion_field_hierarchical = hierarchical_wrapper(previous, current, block_size=args.block_size,
                                                     search_window=args.search_window,
                                                     searching_procedure=args.searching_procedure)
    out_dir = os.path.join("resources", "images")
    os.makedirs(out_dir, exist_ok=True)
    write_image(os.path.join(out_dir, f"{args.searching_procedure}-res.png"), draw_motion_field(current, motion_field))
    write_image(os.path.join(out_dir, f"{args.searching_procedure}h-res.png"),
                draw_motion_field(previous, motion_field_hierarchical))
    return motion_field, motion_field_hierarchical


if __name__ == "__main__":
    import argparse
    parser = argparse.ArgumentParser(description="Computes motion field between two frames using block matching algorithms")
    parser.add_argument("-p", "--video-path", dest="path", type=str, required=True, help="path of the video to analyze")
    parser.add_argument("-fi", "--frame-index", dest="fi", type=int, required=True,
                        help="index of the current frame to analyze in the video")
    parser.add_argument("-pn", "--p-norm", dest="pnorm", type=int, default=0, help="0: 1-norm (mae), 1: 2-norm (mse)")
    parser.add_argument("-bs", "--block-size", dest="block_size", type=int, default=12, help="size of the block")
    parser.add_argument("-sw", "--search-window", dest="search_window", type=int, default=8,
                        help="size of the search window")
    parser.add_argument("-sp", "--searching-procedure", dest="searching_procedure", type=int, default=1,
                        help="0: Exhaustive search, 1: Three Step search, 2: 2D Log search, 3: Diamond search")
    main(parser.parse_args())
