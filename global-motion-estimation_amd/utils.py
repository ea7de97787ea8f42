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


def _cv2(required=True):
    try:
        import cv2
        return cv2
    except ImportError as e:
        if required:
            raise ImportError("this helper needs OpenCV (cv2), which is not installed here") from e
        return None


def get_video_frames(path):
    """utils.py:9-31: list of grayscale uint8 frames.

    With OpenCV installed this is the reference's ``VideoCapture`` loop.  Without it the path may
    be a directory of images (sorted by the number in the file name), a ``.npy`` / ``.npz``
    stack ``uint8[N, H, W]``, or a raw-luma ``.y4m`` file -- decoded here with PIL/NumPy.
    """
    import os
    if os.path.isdir(path):
        return _frames_from_dir(path)
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return [np.ascontiguousarray(f, dtype=np.uint8) for f in np.load(path)]
    if ext == ".npz":
        z = np.load(path)
        return [np.ascontiguousarray(f, dtype=np.uint8) for f in z[z.files[0]]]
    if ext == ".y4m":
        return _frames_from_y4m(path)
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


def _frames_from_dir(path):
    import os
    import re
    from PIL import Image
    names = [n for n in os.listdir(path) if n.lower().endswith((".png", ".jpg", ".jpeg", ".bmp", ".pgm"))]
    names.sort(key=lambda n: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", n)])
    return [np.array(Image.open(os.path.join(path, n)).convert("L"), dtype=np.uint8) for n in names]


def _frames_from_y4m(path):
    with open(path, "rb") as f:
        header = f.readline().decode("ascii", "replace").split()
        if not header or header[0] != "YUV4MPEG2":
            raise ValueError("%s is not a YUV4MPEG2 file" % path)
        w = int(next(t[1:] for t in header if t.startswith("W")))
        h = int(next(t[1:] for t in header if t.startswith("H")))
        cs = next((t[1:] for t in header if t.startswith("C")), "420")
        chroma = {"420": w * h // 2, "422": w * h, "444": 2 * w * h, "mono": 0}
        extra = next(v for k, v in chroma.items() if cs.startswith(k))
        frames = []
        while f.readline().startswith(b"FRAME"):
            luma = np.frombuffer(f.read(w * h), dtype=np.uint8)
            if luma.size < w * h:
                break
            frames.append(luma.reshape(h, w).copy())
            f.seek(extra, 1)
    return frames


def write_image(path, image):
    """cv2.imwrite where OpenCV exists (results.py:64-106), PIL otherwise."""
    cv2 = _cv2(required=False)
    if cv2 is not None:
        return cv2.imwrite(path, image)
    from PIL import Image
    arr = np.asarray(image)
    Image.fromarray(arr if arr.ndim == 2 else arr[:, :, ::-1]).save(path)       # BGR -> RGB like cv2 files
    return True


def draw_motion_field(frame, motion_field):
    """utils.py:54-76: needle diagram of a motion field (red arrows, BGR image).  Anti-aliased
    cv2.arrowedLine when OpenCV is present; otherwise plain PIL lines with a small head (same
    geometry, not pixel-identical)."""
    height, width = frame.shape
    mf_height, mf_width, _ = motion_field.shape
    bs = height // mf_height
    cv2 = _cv2(required=False)
    if cv2 is not None:
        canvas = cv2.cvtColor(frame, cv2.COLOR_GRAY2RGB)
        for y in range(mf_height):
            for x in range(mf_width):
                cx, cy = x * bs + bs // 2, y * bs + bs // 2
                mv_x, mv_y = motion_field[y][x]
                cv2.arrowedLine(canvas, (cx, cy), (int(cx + mv_x), int(cy + mv_y)), (0, 0, 255), 1,
                                line_type=cv2.LINE_AA)
        return canvas
    from PIL import Image, ImageDraw
    img = Image.fromarray(np.stack([frame] * 3, axis=-1))
    pen = ImageDraw.Draw(img)
    for y in range(mf_height):
        for x in range(mf_width):
            cx, cy = x * bs + bs // 2, y * bs + bs // 2
            mv_x, mv_y = motion_field[y][x]
            ex, ey = int(cx + mv_x), int(cy + mv_y)
            pen.line([(cx, cy), (ex, ey)], fill=(255, 0, 0), width=1)
            if ex != cx or ey != cy:                      # arrow head: 10 % of the length, like cv2's tipLength
                ang = np.arctan2(cy - ey, cx - ex)
                tip = 0.1 * np.hypot(ex - cx, ey - cy)
                for da in (np.pi / 4, -np.pi / 4):
                    pen.line([(ex, ey), (int(round(ex + tip * np.cos(ang + da))), int(round(ey + tip * np.sin(ang + da))))],
                             fill=(255, 0, 0), width=1)
    return np.asarray(img)[:, :, ::-1].copy()             # RGB -> BGR, the layout cv2 code expects


def some_data(psnr_path: str) -> None:
    """utils.py:138-164: average / variance / extremes of a ``psnr_records.json``.  Values are
    the strings ``str(complex)`` that results.py writes (``"(22.7+0j)"``); as upstream, a value
    that occurs twice contributes to the variance once (its squared deviation is stored at the
    index of its first occurrence)."""
    import json
    with open(psnr_path, "r") as f:
        psnrs = json.load(f)
    values = np.zeros(shape=[len(psnrs), 1])
    for count, key in enumerate(psnrs):
        text = psnrs[key]
        values[count] = text[1:text.index("+")]
    avg = values.sum() / len(psnrs)
    diff = np.zeros(shape=[len(psnrs), 1])
    as_list = values.tolist()
    for value in values:
        diff[as_list.index(value)] = (value - avg) ** 2
    var = diff.sum() / len(psnrs)
    print("Average: {:.3f}".format(avg))
    print("Variance: {:.3f}".format(var))
    print("Standard deviation: {:.3f}".format(var ** (1 / 2)))
    print("Highest: {:.3f}".format(values.max()))
    print("Lowest: {:.3f}".format(values.min()))
