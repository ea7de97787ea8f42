"""Deterministic synthetic luma sequences, host/NumPy form.

The same generator exists on the device (``csrc/synth.hip``,
``gme_seq_synth``); this module is the host definition used to build inputs
for tests, for the CPU baseline leg of ``bench.py`` and to check the device
generator.  No NumPy RNG is involved: every pixel is a pure function of
``(seed, t, y, x)`` through splitmix64, so NumPy, C and HIP agree bit for bit.

Scene: a 2048x4096 toroidal textured canvas seen by a camera that pans
``(-5, +3)`` px per frame (background content moves ``(+5, -3)`` in the
``(mf[...,0], mf[...,1])`` convention of bbme.py:176-177), a textured
foreground rectangle moving ``(-7, +4)`` per frame (the outlier population
for motion.py:236-244), and +-2 per-pixel noise.
"""
import numpy as np

CANVAS_H = 2048
CANVAS_W = 4096
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    """One splitmix64 output step on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = np.asarray(x, dtype=np.uint64) + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def hash64(seed, k):
    """h(seed, k) = splitmix64(seed * GOLDEN + k)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * _GOLDEN
        return splitmix64(base + np.asarray(k, dtype=np.uint64))


_canvas_cache = {}


def canvas(seed):
    """T = (box5x5(noise) + cell32) // 2 as uint8[2048, 4096]."""
    if seed in _canvas_cache:
        return _canvas_cache[seed]
    idx = np.arange(CANVAS_H * CANVAS_W, dtype=np.uint64)
    n = (hash64(seed, idx) & np.uint64(0xFF)).astype(np.int32)
    n = n.reshape(CANVAS_H, CANVAS_W)
    box = np.zeros_like(n)
    for dy in range(-2, 3):
        rolled = np.roll(n, -dy, axis=0)
        for dx in range(-2, 3):
            box += np.roll(rolled, -dx, axis=1)
    box //= 25
    cy = np.arange(CANVAS_H, dtype=np.uint64) // np.uint64(32)
    cx = np.arange(CANVAS_W, dtype=np.uint64) // np.uint64(32)
    cell_idx = cy[:, None] * np.uint64(128) + cx[None, :]
    cell = (hash64(seed + 1, cell_idx) & np.uint64(0xFF)).astype(np.int32)
    t = ((box + cell) // 2).astype(np.uint8)
    if len(_canvas_cache) > 2:
        _canvas_cache.clear()
    _canvas_cache[seed] = t
    return t


def frame(seed, t, height, width):
    """Frame ``t`` of the sequence ``seed`` as uint8[height, width]."""
    T = canvas(seed)
    ys = (np.arange(height, dtype=np.int64) + 3 * t) % CANVAS_H
    xs = (np.arange(width, dtype=np.int64) - 5 * t) % CANVAS_W
    img = T[ys[:, None], xs[None, :]].astype(np.int32)

    # foreground rectangle, toroidal in the frame so long sequences keep it
    rh, rw = height // 4, width // 6
    if rh > 0 and rw > 0:
        yy = np.arange(rh, dtype=np.uint64)
        xx = np.arange(rw, dtype=np.uint64)
        tex = hash64(seed + 2, yy[:, None] * np.uint64(rw) + xx[None, :])
        tex = (tex & np.uint64(0xFF)).astype(np.int32)
        rows = (height // 3 + 4 * t + np.arange(rh)) % height
        cols = (width // 3 - 7 * t + np.arange(rw)) % width
        img[rows[:, None], cols[None, :]] = tex

    pix = np.arange(height * width, dtype=np.uint64).reshape(height, width)
    noise = (hash64(seed + 3 + t, pix) % np.uint64(5)).astype(np.int32) - 2
    return np.clip(img + noise, 0, 255).astype(np.uint8)


def sequence(seed, t0, count, height, width):
    """uint8[count, height, width] holding frames t0 .. t0+count-1."""
    out = np.empty((count, height, width), dtype=np.uint8)
    for i in range(count):
        out[i] = frame(seed, t0 + i, height, width)
    return out
