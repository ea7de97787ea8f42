"""Exhaustive MSE at bs 16 on the matrix cores (csrc/bbme_mfma.hip, k_exh_mfma16<NT>): every candidate's cross term from
v_mfma_i32_16x16x64_i8, the reference's first-minimum rule on top (bbme.py:105-179, pnorm 1).  Bit-exact against the C
oracle on shapes the tile grid does not divide, at every frame edge, on pan / noise / flat (all ties) / mixed content, for
the four search windows the kernel takes (sw 8 .. 32), at the benched sizes, and against the vector-unit kernels.
Needs an MI355X."""
import os

import numpy as np
import pytest

from helpers import c_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    import _gme_native
    ctx = _gme_native.default_context()
    assert "gfx950" in ctx.info()["name"]
    return _gme_native


class _env:
    def __init__(self, **kv):
        self.kv = kv
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update({k: str(v) for k, v in self.kv.items()})
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _content(rng, kind, n, H, W):
    if kind == "pan":
        base = rng.integers(0, 256, (H + 80, W + 80), dtype=np.uint8)
        fr = np.stack([base[40 + 3 * (t % 7):40 + 3 * (t % 7) + H, 40 - 5 * (t % 5):40 - 5 * (t % 5) + W] for t in range(n)])
    elif kind == "noise":
        fr = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    elif kind == "flat":
        fr = np.full((n, H, W), int(rng.integers(0, 256)), np.uint8)
    elif kind == "extremes":                 # 0 / 255 only: the largest products and sums the int8 form meets
        fr = (rng.integers(0, 2, (n, H, W), dtype=np.uint8) * 255).astype(np.uint8)
    else:
        base = rng.integers(0, 256, (H + 80, W + 80), dtype=np.uint8)
        fr = np.stack([base[40 + (t % 5):40 + (t % 5) + H, 40 - 2 * (t % 6):40 - 2 * (t % 6) + W] for t in range(n)]).copy()
        fr[:, :, :W // 2] = rng.integers(0, 256, (n, H, W // 2), dtype=np.uint8)
    return np.ascontiguousarray(fr)


@pytest.mark.parametrize("sw", [8, 16, 24, 32])
def test_mfma_random_shapes_vs_oracle(native, sw):
    co = c_oracle()
    rng = np.random.default_rng(900 + sw)
    ctx = native.default_context()
    shapes = [(16, 16), (17, 33), (31, 95), (48, 64), (64, 96), (100, 130), (97, 143), (80, 176)]
    checked = 0
    with _env(GME_EXH_MFMA=1):
        for H, W in shapes:
            for kind in ("pan", "noise", "flat", "mixed", "extremes"):
                n = 3
                frames = _content(rng, kind, n, H, W)
                seq = native.Sequence.from_frames(ctx, frames)
                try:
                    fd = int(rng.integers(1, 3))
                    seq.bbme(fd, 16, sw, 0, 1)
                    mv = seq.read_mv()
                    plan = ctx.last_bbme_info()["plan"]
                    assert plan.startswith("k_exh_mfma16<%d>" % ((2 * sw + 16) // 16)), plan
                    for p in range(n - fd):
                        want = co.bbme(frames[p], frames[p + fd], 16, sw, 0, 1)
                        assert np.array_equal(mv[p], want), (H, W, kind, fd, sw, p, np.argwhere(mv[p] != want)[:4])
                        checked += 1
                finally:
                    seq.close()
    assert checked >= 40


@pytest.mark.parametrize("tile", ["1x1", "1x2", "1x3", "1x4"])
def test_mfma_tile_shapes(native, tile):
    """every tile shape the launcher may pick (and the ragged last tiles of each) gives the oracle's field"""
    co = c_oracle()
    rng = np.random.default_rng(77)
    ctx = native.default_context()
    frames = _content(rng, "mixed", 3, 112, 208)              # 7 x 13 blocks: no tile shape divides it
    want = [co.bbme(frames[p], frames[p + 1], 16, 16, 0, 1) for p in range(2)]
    with _env(GME_EXH_MFMA=1, GME_MFMA_TILE=tile):
        seq = native.Sequence.from_frames(ctx, frames)
        try:
            seq.bbme(1, 16, 16, 0, 1)
            mv = seq.read_mv()
            assert ("%s blocks per workgroup" % tile) in ctx.last_bbme_info()["plan"], ctx.last_bbme_info()
            for p in range(2):
                assert np.array_equal(mv[p], want[p]), (tile, p)
        finally:
            seq.close()


def test_mfma_bench_sizes_and_vector_kernels_agree(native):
    """720x480 sw 16 (configs[1]'s geometry under MSE) and 1920x1080 sw 32 (configs[3]): the synthetic sequence and uniform
    noise; sampled pairs against the C oracle, EVERY pair against the vector-unit kernels (elimination + redo)."""
    co = c_oracle()
    ctx = native.default_context()
    for (H, W, sw, n, chk) in ((480, 720, 16, 17, 6), (1080, 1920, 32, 5, 2)):
        for kind in ("synthetic", "noise"):
            if kind == "synthetic":                        # bench.py's sequence, generated on the device (bit-identical to synth.py)
                seq = native.Sequence(ctx, n, H, W)
                seq.synth(1234)
                frames = np.stack([seq.read_frame(t) for t in range(n)])
            else:
                frames = np.random.default_rng(5).integers(0, 256, (n, H, W), dtype=np.uint8)
                seq = native.Sequence.from_frames(ctx, np.ascontiguousarray(frames))
            try:
                with _env(GME_EXH_MFMA=0):
                    seq.bbme(1, 16, sw, 0, 1)
                    vec = seq.read_mv().copy()
                    assert ctx.last_bbme_info()["plan"].startswith("k_exh_sea16"), ctx.last_bbme_info()
                with _env(GME_EXH_MFMA=1):
                    seq.invalidate_pyramids()
                    seq.bbme(1, 16, sw, 0, 1)
                    mv = seq.read_mv()
                    assert ctx.last_bbme_info()["plan"].startswith("k_exh_mfma16"), ctx.last_bbme_info()
                assert np.array_equal(mv, vec), (H, W, kind, np.argwhere(mv != vec)[:4])
                for p in list(range(chk - 1)) + [n - 2]:
                    assert np.array_equal(mv[p], co.bbme(frames[p], frames[p + 1], 16, sw, 0, 1)), (H, W, kind, p)
            finally:
                seq.close()


def test_mfma_full_batch_equals_vector_unit(native):
    """bench.py's own batch (2048 resident pairs of 720x480, the synthetic sequence and the reference's real pan240 frames
    upscaled x2): the matrix-core kernel and the vector unit's elimination kernel -- two independent implementations, the
    second pinned on the goldens and checked pair by pair in test_gpu_round2.py -- must return the same 2048 fields, and the
    first and last pair must be the C oracle's."""
    import bench
    co = c_oracle()
    ctx = native.default_context()
    for kind in ("synthetic", "pan240x2"):
        if kind == "synthetic":
            seq = native.Sequence(ctx, 2049, 480, 720)
            seq.synth(1234)
            ends = [(seq.read_frame(p), seq.read_frame(p + 1)) for p in (0, 2047)]
        else:
            frames, H, W = bench.host_content("pan240x2", 2049, 480, 720)
            seq = native.Sequence.from_frames(ctx, frames)
            ends = [(frames[p], frames[p + 1]) for p in (0, 2047)]
        try:
            seq.bbme(1, 16, 16, 0, 1)
            assert ctx.last_bbme_info()["plan"].startswith("k_exh_mfma16<3>"), ctx.last_bbme_info()
            mv = seq.read_mv().copy()
            with _env(GME_EXH_MFMA=0):
                seq.invalidate_pyramids()
                seq.bbme(1, 16, 16, 0, 1)
                assert ctx.last_bbme_info()["plan"].startswith("k_exh_sea16p_mse"), ctx.last_bbme_info()
                vec = seq.read_mv()
            assert mv.shape[0] == 2048 and np.array_equal(mv, vec), (kind, np.argwhere(mv != vec)[:4])
            for p, (a, b) in zip((0, 2047), ends):
                assert np.array_equal(mv[p], co.bbme(a, b, 16, 16, 0, 1)), (kind, p)
        finally:
            seq.close()
