"""Stand-in for the one OpenCV call on the hot path: ``cv2.pyrDown``.

TEST INFRASTRUCTURE, used only by ``oracle/refimport/make_golden.py`` in the
build container so that the upstream Python modules can be imported there
(``utils.py:3`` and ``bbme.py:6`` do ``import cv2``; opencv-python is not
installed and cannot be fetched). It is never imported by the product path.

``pyrDown`` restates OpenCV's documented uint8 algorithm (opencv-python
4.5.5.62, pinned in the reference's requirements.txt:3; sole call site
utils.py:48): separable 5-tap [1 4 6 4 1] kernel, integer accumulation,
``(sum + 128) >> 8`` rounding, BORDER_REFLECT_101, destination size
``((w + 1) // 2, (h + 1) // 2)``.  PARITY UNPINNED: no OpenCV binary and no
reference fixture is available to check this function against.
"""
import numpy as np

# names the reference touches only in drawing / decoding helpers (off path)
LINE_AA = 16
COLOR_BGR2GRAY = 6
COLOR_GRAY2RGB = 8


def _reflect101(p, n):
    if n == 1:
        return np.zeros_like(p)
    p = np.array(p, dtype=np.int64)
    while True:
        neg = p < 0
        big = p >= n
        if not (neg.any() or big.any()):
            return p
        p = np.where(neg, -p, p)
        p = np.where(p >= n, 2 * (n - 1) - p, p)


def pyrDown(src):
    src = np.asarray(src)
    if src.ndim != 2 or src.dtype != np.uint8:
        raise TypeError("stub pyrDown handles 2-D uint8 only")
    h, w = src.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    k = (1, 4, 6, 4, 1)
    s = src.astype(np.int64)
    xs = 2 * np.arange(dw)
    hp = np.zeros((h, dw), dtype=np.int64)
    for d in range(-2, 3):
        hp += k[d + 2] * s[:, _reflect101(xs + d, w)]
    ys = 2 * np.arange(dh)
    vp = np.zeros((dh, dw), dtype=np.int64)
    for d in range(-2, 3):
        vp += k[d + 2] * hp[_reflect101(ys + d, h), :]
    return ((vp + 128) >> 8).astype(np.uint8)
