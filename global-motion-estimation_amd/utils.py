"""Image utilities -- drop-in for the on-path part of the reference's ``utils`` module.

``get_pyramids`` (utils.py:34-51) and ``PSNR`` (utils.py:100-116) run on the GPU.  The
video/drawing helpers need OpenCV, which is imported lazily so that this module (and
``results.py``-style drivers) import fine where ``cv2`` is absent.
"""
import time
from cmath import log10, sqrt

import numpy as np

import _gme_native as _native


def get_pyramids(original_image, levels=3):
    """utils.py:34-51: ``[pyrDown^(levels-1)(img), ..., pyrDown(img), img]``, coarse first.

    ``cv2.pyrDown`` is restated in ``csrc/gme_kernels.hip`` (5x5 [1 4 6 4 1]^2,
    BORDER_REFLECT_101, ``(s + 128) >> 8``); parity with OpenCV itself is unpinned
    (no OpenCV available offline).
    """
    ctx = _native.default_context()
    pyramid = [original_image]
    curr = original_image
    for _ in range(1, levels):
        curr = ctx.pyrdown(curr)
        pyramid.insert(0, curr)
    return pyramid


def PSNR(original, noisy):
    """utils.py:100-116: returns -1 for identical images, else a *complex* number (the
    reference uses cmath); take ``.real``."""
    sse = _native.default_context().sse(original, noisy)
    mse = sse / (original.shape[0] * original.shape[1])        # == np.mean of the squared ints
    if mse == 0:
        return -1
    return 20 * log10(255.0 / sqrt(mse))


def timer(func):
    """utils.py:79-97."""
    def wrapper(*args, **kwargs):
        start = int(time.time())
        ret = func(*args, **kwargs)
        end = int(time.time())
        print(f"Execution of '{func.__name__}' in {end-start}s")
        return ret
    return wrapper


def _cv2():
    try:
        import cv2
    except ImportError as e:       # pragma: no cover - depends on the host
        raise ImportError("this helper needs OpenCV (cv2), which is not installed here; "
                          "feed uint8 arrays to bbme/motion directly instead") from e
    return cv2


def get_video_frames(path):
    """utils.py:9-31 (needs cv2): list of grayscale uint8 frames."""
    cv2 = _cv2()
    cap = cv2.VideoCapture(path)
    frames = []
    while cap.grab():
        ok, frame = cap.retrieve()
        if not ok:
            break
        if frame.ndim == 3 and frame.shape[2] == 3:
            frame = cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY)
        frames.append(frame)
    return frames


def draw_motion_field(frame, motion_field):
    """utils.py:54-76 (needs cv2): needle diagram."""
    cv2 = _cv2()
    height, width = frame.shape
    canvas = cv2.cvtColor(frame, cv2.COLOR_GRAY2RGB)
    mf_height, mf_width, _ = motion_field.shape
    bs = height // mf_height
    for y in range(mf_height):
        for x in range(mf_width):
            cx, cy = x * bs + bs // 2, y * bs + bs // 2
            mv_x, mv_y = motion_field[y][x]
            cv2.arrowedLine(canvas, (cx, cy), (int(cx + mv_x), int(cy + mv_y)), (0, 0, 255), 1,
                            line_type=cv2.LINE_AA)
    return canvas
