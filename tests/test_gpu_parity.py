"""Parity of the HIP path (through the C ABI / the Python drop-in surface) against the
golden vectors of the real reference and against the oracles.  Needs an MI355X.

Bars: motion vectors, model fields, masks, thresholds, compensated frames, squared
errors bit-exact; normal-equation sums bit-exact float64; parameters rtol 1e-10
(the 3x3 inverse is LAPACK-build dependent); PSNR |d| < 1e-9 dB
(north_star allows 0.01 dB).
"""
import re

import numpy as np
import pytest

from helpers import c_oracle, np_oracle, oracle_gme, oracle_results_flow, sha

pytestmark = pytest.mark.gpu

G1_KEY = re.compile(r"mf_(\w+?_\d+x\d+)_bs(\d+)_sw(\d+)_sp(\d)_pn(\d)$")


@pytest.fixture(scope="module")
def mods():
    import _gme_native
    import bbme
    import motion
    import utils
    ctx = _gme_native.default_context()
    assert "gfx950" in ctx.info()["name"]
    return _gme_native, bbme, motion, utils


def test_small_fields_all_searches(golden, mods):
    _, bbme, _, _ = mods
    g = golden("g1_small")
    n = 0
    for k in g.files:
        m = G1_KEY.match(k)
        if not m:
            continue
        name, bs, sw, sp, pn = m.group(1), *map(int, m.groups()[1:])
        got = bbme.get_motion_field(g["in_%s_prev" % name], g["in_%s_cur" % name], block_size=bs,
                                    search_window=sw, searching_procedure=sp, pnorm_distance=pn)
        assert got.dtype == np.int32 and got.shape == g[k].shape
        assert np.array_equal(got, g[k]), k
        n += 1
    assert n == 320


def test_synth720_all_searches(golden, mods):
    """BASELINE config 2 (720x480, bs=16, sw=16) plus the three fast searches, both norms."""
    _, bbme, _, _ = mods
    import synth
    g = golden("g2_synth720")
    p, c = synth.frame(1234, 0, 480, 720), synth.frame(1234, 1, 480, 720)
    for sp in range(4):
        for pn in range(2):
            got = bbme.get_motion_field(p, c, block_size=16, search_window=16, searching_procedure=sp,
                                        pnorm_distance=pn)
            assert np.array_equal(got, g["mf_sp%d_pn%d" % (sp, pn)]), (sp, pn)


def test_doc_frames_all_searches(golden, mods):
    """BASELINE config 1 stand-in: the reference's own doc frames (real content)."""
    _, bbme, _, _ = mods
    g = golden("g3_docframes")
    for tag in ("race", "pan240"):
        p, c = g["in_%s_prev" % tag], g["in_%s_cur" % tag]
        for sp in range(4):
            for pn in range(2):
                got = bbme.get_motion_field(p, c, block_size=16, search_window=16, searching_procedure=sp,
                                            pnorm_distance=pn)
                assert np.array_equal(got, g["%s_mf_sp%d_pn%d" % (tag, sp, pn)]), (tag, sp, pn)


def test_search_function_signatures(golden, mods):
    """bbme.exhaustive_search & co. fill `mf` in place and return it (bbme.py:105-114)."""
    _, bbme, _, _ = mods
    g = golden("g1_small")
    p, c = g["in_shifted_64x96_prev"], g["in_shifted_64x96_cur"]
    for sp, fn in enumerate(bbme.searching_procedures):
        mf = np.zeros((16, 24, 2), np.int32)
        out = fn(p, c, mf, 64, 96, 0, 4, 2)
        assert out is mf
        assert np.array_equal(mf, g["mf_shifted_64x96_bs4_sw2_sp%d_pn0" % sp])
    assert np.array_equal(bbme.get_motion_field(p, c, 4, 2, -1, -2), g["mf_shifted_64x96_bs4_sw2_sp3_pn0"])
    with pytest.raises(IndexError):
        bbme.get_motion_field(p, c, searching_procedure=4)
    with pytest.raises(IndexError):
        bbme.get_motion_field(p, c, pnorm_distance=2)
    with pytest.raises(TypeError):
        bbme.get_motion_field(p.astype(np.float32), c.astype(np.float32))


def test_corner_cases(golden, mods):
    native, bbme, _, _ = mods
    g, g1 = golden("g6_edges"), golden("g1_small")
    p, c = g1["in_shifted_64x96_prev"], g1["in_shifted_64x96_cur"]
    for sw in (1, 2, 3):
        assert np.array_equal(bbme.get_motion_field(p, c, 8, sw, 2, 0), g["tdl_sw%d" % sw])
    p, c = g1["in_random_50x70_prev"], g1["in_random_50x70_cur"]
    assert np.array_equal(bbme.get_motion_field(p, c, 4, 20, 1, 1), g["tss_wild"])
    assert np.array_equal(bbme.get_motion_field(p, c), g["tss_default"])
    # MSE with bs = 32 leaves float32's exact-integer range: the kernels then sum in NumPy's
    # float32 pairwise order, like the reference (goldens from the real reference)
    for pre, sw in (("bs32", 4), ("bs32hc", 6)):
        for sp in (0, 3):
            got = bbme.get_motion_field(g[pre + "_prev"], g[pre + "_cur"], 32, sw, sp, 1)
            assert np.array_equal(got, g["%s_sp%d" % (pre, sp)]), (pre, sp)
    co = c_oracle()
    rng = np.random.default_rng(5)
    hp, hc = (rng.integers(0, 2, (72, 100)) * 255).astype(np.uint8), (rng.integers(0, 2, (72, 100)) * 255).astype(np.uint8)
    for bs, sw in ((24, 5), (17, 3), (20, 8)):
        for sp in range(4):
            assert np.array_equal(bbme.get_motion_field(hp, hc, bs, sw, sp, 1), co.bbme(hp, hc, bs, sw, sp, 1)), (bs, sp)
    for k in range(3):          # near-ties decided by NumPy's float32 summation order
        cur = g["f32tie_cur_%d" % k]
        assert np.array_equal(bbme.get_motion_field(np.zeros_like(cur), cur, 32, 1, 0, 1), g["f32tie_mf_%d" % k]), k
    # ... while MAE at bs = 32 is exact
    for sp in range(4):
        assert np.array_equal(bbme.get_motion_field(g["bs32_prev"], g["bs32_cur"], 32, 4, sp, 0),
                              co.bbme(g["bs32_prev"], g["bs32_cur"], 32, 4, sp, 0))
    # diamond cannot place a block when H == bs (reference: AssertionError on an empty slice)
    with pytest.raises(AssertionError):
        bbme.get_motion_field(p[:16], c[:16], 16, 2, 3, 0)
    # frames smaller than one block give an empty field
    assert bbme.get_motion_field(p[:8, :8], c[:8, :8], 16, 2, 0, 0).shape == (0, 0, 2)


@pytest.mark.parametrize("shape,bs,sw", [((37, 53), 8, 7), ((70, 101), 16, 16), ((48, 64), 16, 8),
                                         ((33, 47), 4, 5), ((96, 130), 16, 32), ((40, 56), 2, 3),
                                         ((64, 80), 16, 4), ((90, 70), 12, 6)])
def test_random_frames_vs_c_oracle(mods, shape, bs, sw):
    """Ragged sizes, tie-heavy content, every search x norm against the C oracle."""
    _, bbme, _, _ = mods
    co = c_oracle()
    rng = np.random.default_rng(hash((shape, bs, sw)) & 0xFFFF)
    base = rng.integers(0, 256, (shape[0] + 8, shape[1] + 8), dtype=np.uint8)
    contents = {
        "noise": (base[4:-4, 4:-4], rng.integers(0, 256, shape, dtype=np.uint8)),
        "shift": (base[4:-4, 4:-4], base[2:-6, 7:shape[1] + 7]),
        "ties": ((base[4:-4, 4:-4] // 128 * 128).astype(np.uint8), (base[3:-5, 5:-3] // 128 * 128).astype(np.uint8)),
        "flat": (np.full(shape, 9, np.uint8), np.full(shape, 9, np.uint8)),
    }
    for name, (p, c) in contents.items():
        p, c = np.ascontiguousarray(p), np.ascontiguousarray(c)
        for sp in range(4):
            for pn in range(2):
                got = bbme.get_motion_field(p, c, bs, sw, sp, pn)
                assert np.array_equal(got, co.bbme(p, c, bs, sw, sp, pn)), (name, sp, pn)


def test_gme_stages_vs_golden(golden, mods):
    """motion.global_motion_estimation stage by stage on five real/synthetic pairs."""
    native, _, motion, _ = mods
    from test_oracle import _gme_inputs
    g = golden("g4_gme")
    ctx = native.default_context()
    for tag in ("synth720", "race", "pan240", "dp", "small", "bs12"):
        bs = 12 if tag == "bs12" else 16
        pre = tag + "_"
        prev, cur = _gme_inputs(golden, tag)
        seq = native.Sequence.from_frames(ctx, [prev, cur])
        p0 = seq.gme_begin(1, bs)
        pyr = [seq.read_frame(i, lvl) for i in (0, 1) for lvl in (0, 1, 2)]
        assert [sha(x) for x in pyr] == [str(s) for s in g[pre + "pyr_sha"]], tag
        assert np.array_equal(seq.gme_read_stage(0, 0)["gt"], g[pre + "dense"]), tag
        assert p0.dtype == np.float32 and np.array_equal(p0[0], g[pre + "params0"]), tag
        params = p0[0]
        for lvl in (1, 2):
            params = motion.parameter_projection(params)
            pin = g[pre + "l%d_params_in" % lvl]
            np.testing.assert_allclose(params, pin, rtol=1e-10, atol=1e-12)
            sums = seq.gme_fit(lvl, np.asarray(pin, np.float64)[None], 0.3)[0]
            st = seq.gme_read_stage(lvl, 0)
            assert np.array_equal(st["gt"], g[pre + "l%d_gt" % lvl]), (tag, lvl)
            assert np.array_equal(st["model"], g[pre + "l%d_model" % lvl]), (tag, lvl)
            assert st["thr"] == int(g[pre + "l%d_thr" % lvl]), (tag, lvl)
            assert np.array_equal(st["mask"], g[pre + "l%d_mask" % lvl]), (tag, lvl)
            want = np.concatenate([g[pre + "l%d_F" % lvl].reshape(9), g[pre + "l%d_Sx" % lvl], g[pre + "l%d_Sy" % lvl]])
            assert sums.tobytes() == want.tobytes(), (tag, lvl)          # bit patterns
            params = motion._solve(sums)
        np.testing.assert_allclose(params, g[pre + "params"], rtol=1e-10, atol=1e-12)
        sse = seq.compensate(1, bs, params[None])
        comp = seq.read_compensated(0)
        assert sha(comp) == str(g[pre + "comp_sha"]), tag
        mse = int(sse[0]) / prev.size
        assert abs(20 * np.log10(255.0 / np.sqrt(mse)) - float(g[pre + "psnr"])) < 1e-9
        seq.close()


def test_public_gme_functions_vs_golden(golden, mods):
    """The drop-in call sequence of results.py:50-59,109 on config 3 (720x480 full GME)."""
    _, _, motion, utils = mods
    from test_oracle import _gme_inputs
    g = golden("g4_gme")
    for tag in ("synth720", "pan240"):
        prev, cur = _gme_inputs(golden, tag)
        params = motion.global_motion_estimation(prev, cur)
        assert params.dtype == np.float64 and params.shape == (6,)
        np.testing.assert_allclose(params, g[tag + "_params"], rtol=1e-10, atol=1e-12)
        field = motion.get_motion_field_affine(
            (int(prev.shape[0] / motion.BBME_BLOCK_SIZE), int(prev.shape[1] / motion.BBME_BLOCK_SIZE), 2),
            parameters=params)
        assert field.dtype == np.int16 and np.array_equal(field, g[tag + "_field"])
        comp = motion.compensate_frame(prev, field)
        assert sha(comp) == str(g[tag + "_comp_sha"])
        psnr = utils.PSNR(cur, comp)
        assert isinstance(psnr, complex) and abs(psnr.real - float(g[tag + "_psnr"])) < 1e-9
        assert np.array_equal(motion.motion_compensation(prev, cur), comp)
    assert utils.PSNR(prev, prev) == -1
    # pyramids: coarse level first, same bytes as the oracle's restatement of cv2.pyrDown
    pyr = utils.get_pyramids(prev)
    co = c_oracle()
    assert [x.shape for x in pyr] == [(60, 80), (120, 160), (240, 320)]
    assert np.array_equal(pyr[1], co.pyrdown(prev)) and np.array_equal(pyr[0], co.pyrdown(co.pyrdown(prev)))


def test_block_size_patch_and_first_estimation(golden, mods):
    """Authors patch motion.BBME_BLOCK_SIZE (presentation/main.tex:382); it is read at call time."""
    _, _, motion, _ = mods
    from test_oracle import _gme_inputs
    g = golden("g4_gme")
    prev, cur = _gme_inputs(golden, "bs12")
    old = motion.BBME_BLOCK_SIZE
    motion.BBME_BLOCK_SIZE = 12
    try:
        np.testing.assert_allclose(motion.global_motion_estimation(prev, cur), g["bs12_params"], rtol=1e-10, atol=1e-12)
        assert np.array_equal(motion.motion_compensation(prev, cur), g["bs12_comp"])
    finally:
        motion.BBME_BLOCK_SIZE = old
    prev, cur = _gme_inputs(golden, "small")
    pyr_p, pyr_c = __import__("utils").get_pyramids(prev), __import__("utils").get_pyramids(cur)
    p0 = motion.first_parameter_estimation(pyr_p[0], pyr_c[0])
    assert p0.dtype == np.float32 and np.array_equal(p0, g["small_params0"])
    assert np.array_equal(motion.dense_motion_estimation(pyr_p[0], pyr_c[0]), g["small_dense"])
    # robust fit on explicit level frames (motion.py:210-286)
    pin = g["small_l1_params_in"]
    out = motion.best_affine_parameters_robust(pyr_p[1], pyr_c[1], pin.copy())
    np.testing.assert_allclose(motion.parameter_projection(out.copy()), g["small_l2_params_in"], rtol=1e-10, atol=1e-12)


def test_affine_compensate_psnr_edges(golden, mods):
    _, _, motion, utils = mods
    g = golden("g6_edges")
    for k in range(24):
        want = g["aff_f_%d" % k]
        got = motion.get_motion_field_affine(want.shape, g["aff_p_%d" % k])
        assert got.dtype == np.int16 and np.array_equal(got, want), k
    assert np.array_equal(motion.get_motion_field_affine((4, 6, 2), g["aff_p_0"]), g["aff_f_tuple3"])
    for k in range(5):
        f = g["comp_in_%d" % k]
        for t in ("16", "32"):
            assert np.array_equal(motion.compensate_frame(f, g["comp_mf%s_%d" % (t, k)]), g["comp_out%s_%d" % (t, k)]), (k, t)
    f = g["comp_in_narrow"]
    for t in ("narrow", "wide"):
        assert np.array_equal(motion.compensate_frame(f, g["comp_mf_" + t]), g["comp_out_" + t]), t
    assert abs(utils.PSNR(g["psnr_a"], g["psnr_b"]).real - float(g["psnr_ab"])) < 1e-12
    for k in range(4):
        assert np.array_equal(motion.compute_first_parameters(g["fp_in_%d" % k]), g["fp_out_%d" % k])


def test_compensate_fast_and_generic_kernels(mods, monkeypatch):
    """k_compensate16 (16 pixels per thread; bs % 16 == 0 and W % 16 == 0) and the generic
    k_compensate against the C oracle's motion.compensate_frame (motion.py:289-321): random
    vectors up to +-60 px so that sources leave the frame on every side, rows/columns beyond the
    field, bs 16 and 32, then the batched form with its fused squared error."""
    native, _, motion, _ = mods
    co = c_oracle()
    rng = np.random.default_rng(77)
    cases = []
    for (H, W, bs) in ((96, 160, 16), (100, 176, 16), (480, 720, 16), (130, 256, 32), (70, 90, 16), (64, 64, 8)):
        f = rng.integers(0, 256, (H, W), dtype=np.uint8)
        mf = rng.integers(-60, 61, (H // bs, W // bs, 2)).astype(np.int16)
        mf[0, 0] = (0, 0)
        cases.append((f, mf))
        want = co.compensate(f, mf)
        assert np.array_equal(motion.compensate_frame(f, mf), want), (H, W, bs)
        fewer = mf[:-1, :-1] if mf.shape[0] > 2 and H % mf[:-1].shape[0] == 0 else mf
        assert np.array_equal(motion.compensate_frame(f, fewer), co.compensate(f, fewer))
    monkeypatch.setenv("GME_FORCE_GENERIC", "1")
    for f, mf in cases:
        assert np.array_equal(motion.compensate_frame(f, mf), co.compensate(f, mf))
    monkeypatch.delenv("GME_FORCE_GENERIC")
    # batched: model field from parameters + compensation + squared error, both kernels
    ctx = native.default_context()
    seq = native.Sequence(ctx, 4, 128, 192)
    seq.synth(9, 0)
    frames = [seq.read_frame(i) for i in range(4)]
    params = np.array([[5.2, 0.11, -0.07, -3.4, 0.05, 0.2], [70.0, 0, 0, -90.0, 0, 0], [-0.4, 1.9, 0.0, 0.3, 0.0, -2.2]])
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("GME_FORCE_GENERIC", env)
        sse = seq.compensate(1, 16, params)
        for p in range(3):
            field = motion.get_motion_field_affine((8, 12), params[p])
            want = co.compensate(frames[p], field)
            assert np.array_equal(seq.read_compensated(p), want), (env, p)
            assert int(sse[p]) == co.sse(frames[p + 1], want), (env, p)
    seq.close()


def test_degenerate_fits(golden, mods):
    """N < 4 blocks gives threshold index 0 and a singular system: LinAlgError as upstream."""
    _, _, motion, _ = mods
    g = golden("g6_edges")
    for tag in ("n6", "n3", "n4"):
        p, c = g["fit_%s_prev" % tag], g["fit_%s_cur" % tag]
        pin = np.array([1.0, 0, 0, -1.0, 0, 0], np.float32)
        if str(g["fit_%s_err" % tag]):
            with pytest.raises(np.linalg.LinAlgError):
                motion.best_affine_parameters_robust(p, c, pin)
        else:
            np.testing.assert_allclose(motion.best_affine_parameters_robust(p, c, pin), g["fit_%s_out" % tag],
                                       rtol=1e-9, atol=1e-11)
    # unmasked variant (motion.py:33-88) against the oracle's sums with an all-false mask
    o, co = np_oracle(), c_oracle()
    p, c = g["fit_n6_prev"], g["fit_n6_cur"]
    p2, c2 = np.tile(p, (3, 3)), np.tile(c, (3, 3))
    gt = co.bbme(p2, c2, 16, 2, 3, 1)
    F, Sx, Sy = o.normal_sums(gt, np.zeros(gt.shape[:2], bool), p2.shape)
    np.testing.assert_allclose(motion.best_affine_parameters(p2, c2), o.solve_parameters(F, Sx, Sy), rtol=1e-10, atol=1e-12)


def test_sequence_batch_matches_reference_flow(golden, mods):
    """results.py:41-112 over a 6-frame sequence at frame distance 1 and 2, batched on the device."""
    native, _, motion, _ = mods
    import sequence
    import synth
    g = golden("g7_sequence")
    frames = synth.sequence(2000, 0, 6, 128, 192)
    for fd in (1, 2):
        sh = sequence.ShardedSequence(128, 192, 6, fd)
        sh.load(frames)
        params = sh.estimate()
        psnr = sh.compensate(params)
        assert params.shape == (6 - fd, 6)
        for p in range(6 - fd):
            k = "fd%d_i%d_" % (fd, p + fd)
            np.testing.assert_allclose(params[p], g[k + "params"], rtol=1e-10, atol=1e-12)
            assert np.array_equal(sh.seq.read_compensated(p), g[k + "comp"]), k
            assert abs(psnr[p] - float(g[k + "psnr"])) < 1e-9
    whole = sequence.ShardedSequence(128, 192, 6, 1)
    whole.load(frames)
    want = whole.estimate()
    want_psnr = whole.compensate(want)
    # several streams per GPU (one host thread each) change nothing but the schedule
    multi = sequence.ShardedSequence(128, 192, 6, 1, streams=3)
    multi.load(frames)
    assert len(multi.lanes) == 3 and multi.seq is None
    mp, mpsnr = multi.estimate_and_compensate(exact_psnr=True)
    assert np.array_equal(mp, want) and np.array_equal(mpsnr, want_psnr)
    assert np.abs(multi.estimate_and_compensate()[1] - want_psnr).max() < 1e-12
    assert np.array_equal(multi.read_compensated(4), whole.seq.read_compensated(4))
    assert np.array_equal(multi.motion_fields(16, 8, 0, 0), whole.motion_fields(16, 8, 0, 0))
    multi.close()
    # two virtual ranks on one device reproduce the single-rank answer
    parts = []
    for r in range(2):
        sh = sequence.ShardedSequence(128, 192, 6, 1, rank=r, world=2)
        sh.load(frames)
        parts.append(sh.estimate())
    assert np.array_equal(np.concatenate(parts), want)


def test_device_synth_matches_host(mods):
    native, _, _, _ = mods
    import synth
    ctx = native.default_context()
    for (seed, t0, n, h, w) in ((1234, 0, 2, 480, 720), (2000, 37, 3, 270, 480), (5, 1000, 2, 33, 47)):
        seq = native.Sequence(ctx, n, h, w)
        seq.synth(seed, t0)
        for i in range(n):
            assert np.array_equal(seq.read_frame(i), synth.frame(seed, t0 + i, h, w)), (seed, t0 + i)
        seq.close()
    seq = native.Sequence(ctx, 2, 480, 720)
    seq.synth(1234, 0)
    assert sha(seq.read_frame(0)) == "9736c2ac7184b594c41cb75edac231feac5775691bca78fc0af2cb8674ae7308"
    assert sha(seq.read_frame(1)) == "9652a5b6f736753131bdcc1961978ceb4238d311bb56ddba6e66d15ddffa7217"


def test_batched_exhaustive_is_pairwise(mods):
    """Every pair of a resident sequence gets the field the single-pair call gives."""
    native, bbme, _, _ = mods
    ctx = native.default_context()
    seq = native.Sequence(ctx, 11, 96, 160)
    seq.synth(42, 5)
    frames = [seq.read_frame(i) for i in range(11)]
    for fd, pn in ((1, 0), (3, 0), (2, 1)):
        seq.bbme(fd, 16, 16, 0, pn)
        mv = seq.read_mv()
        assert mv.shape == (11 - fd, 6, 10, 2)
        co = c_oracle()
        for p in range(11 - fd):
            assert np.array_equal(mv[p], co.bbme(frames[p], frames[p + fd], 16, 16, 0, pn)), (fd, p)


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_sea_schedules_agree(mods, monkeypatch, mode):
    """k_exh_sea16[_mse] (one tile per workgroup, GME_SEA_PERSIST=0) and the persistent k_exh_sea16p[_mse]
    with the static (1) and the dynamic (2) tile schedule must all give the oracle's fields: ragged
    block rows, pair counts that are not a multiple of the 8 XCDs, every window size class, both norms."""
    native, bbme, _, _ = mods
    monkeypatch.setenv("GME_SEA_PERSIST", mode)
    monkeypatch.setenv("GME_EXH_MFMA", "0")          # MSE at sw <= 16 would otherwise take the matrix-core kernel
    ctx = native.default_context()
    co = c_oracle()
    for (n, h, w, sw, seed) in ((12, 96, 176, 16, 3), (4, 70, 330, 8, 4), (10, 50, 66, 4, 5), (3, 130, 150, 32, 6),
                                (5, 80, 112, 24, 7), (9, 48, 80, 0, 8)):
        seq = native.Sequence(ctx, n, h, w)
        seq.synth(seed, 0)
        frames = [seq.read_frame(i) for i in range(n)]
        for pn in (0, 1):
            seq.bbme(1, 16, sw, 0, pn)
            mv = seq.read_mv()
            for p in range(n - 1):
                assert np.array_equal(mv[p], co.bbme(frames[p], frames[p + 1], 16, sw, 0, pn)), (mode, n, h, w, sw, pn, p)
        seq.close()


@pytest.mark.parametrize("pnorm", [0, 1])
def test_persistent_equals_one_tile_kernel_1080p(mods, monkeypatch, pnorm):
    """BASELINE size (1920x1080, sw=32, R=5 windows): the persistent kernels (static and dynamic
    schedule, prefetched tiles, 4-lane phase E for MAE) return exactly the fields of the
    one-tile-per-workgroup kernels, which the golden/oracle tests pin at this size."""
    native, _, _, _ = mods
    ctx = native.default_context()
    seq = native.Sequence(ctx, 4, 1080, 1920)
    seq.synth(4321, 0)
    fields = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("GME_SEA_PERSIST", mode)
        seq.bbme(1, 16, 32, 0, pnorm)
        fields[mode] = seq.read_mv()
    seq.close()
    assert fields["0"].shape == (3, 67, 120, 2)
    assert np.array_equal(fields["1"], fields["0"]) and np.array_equal(fields["2"], fields["0"])
    # the synthetic camera moves the background by (5, -3) per frame: most blocks must say so
    mv = fields["0"].reshape(-1, 2)
    assert np.mean((mv[:, 0] == 5) & (mv[:, 1] == -3)) > 0.7


def test_full_size_1080p(golden, mods):
    """BASELINE configs 4/5 sizes: exhaustive MSE sw=32 and the GME stages at 1920x1080."""
    native, bbme, motion, _ = mods
    import synth
    g = golden("g5_1080p")
    p, c = synth.frame(4321, 0, 1080, 1920), synth.frame(4321, 1, 1080, 1920)
    assert np.array_equal(bbme.get_motion_field(p, c, 16, 32, 0, 1), g["exh_mse_sw32"])
    co = c_oracle()
    assert np.array_equal(bbme.get_motion_field(p, c, 16, 32, 0, 0), co.bbme(p, c, 16, 32, 0, 0))
    seq = native.Sequence.from_frames(native.default_context(), [p, c])
    seq.gme_begin(1, 16)
    assert np.array_equal(seq.gme_read_stage(0, 0)["gt"], g["gme_dense"])
    for lvl in (1, 2):
        sums = seq.gme_fit(lvl, np.asarray(g["gme_l%d_params_in" % lvl], np.float64)[None], 0.3)[0]
        st = seq.gme_read_stage(lvl, 0)
        assert np.array_equal(st["gt"], g["gme_l%d_gt" % lvl])
        assert np.array_equal(st["mask"], g["gme_l%d_mask" % lvl])
        want = np.concatenate([g["gme_l%d_F" % lvl].reshape(9), g["gme_l%d_Sx" % lvl], g["gme_l%d_Sy" % lvl]])
        assert sums.tobytes() == want.tobytes()
    np.testing.assert_allclose(motion.global_motion_estimation(p, c), g["gme_params"], rtol=1e-10, atol=1e-12)


def _oracle_gme(prev, cur, procedure, sw, bs=16, frac=0.3):
    return oracle_gme(prev, cur, procedure, sw, bs, frac)


@pytest.mark.parametrize("procedure,sw", [(0, 8), (0, 16), (1, 7), (2, 6)])
def test_gme_with_other_searches(mods, procedure, sw):
    """gme_seq_gme_begin's procedure/search_window arguments (exhaustive MSE = BASELINE config 4)."""
    native, _, motion, _ = mods
    import synth
    frames = synth.sequence(31, 2, 3, 144, 208)
    seq = native.Sequence.from_frames(native.default_context(), frames)
    got = motion.estimate_sequence(seq, 1, procedure, sw)
    for p in range(2):
        want, stages = _oracle_gme(frames[p], frames[p + 1], procedure, sw)
        for lvl in (1, 2):
            st = seq.gme_read_stage(lvl, p)
            assert np.array_equal(st["gt"], stages[lvl - 1]["gt"]), (p, lvl)
        assert np.array_equal(seq.gme_read_stage(2, p)["mask"], stages[1]["mask"])
        np.testing.assert_allclose(got[p], want, rtol=1e-9, atol=1e-11)
    seq.close()


def test_generic_kernels_still_match(mods, monkeypatch):
    """GME_FORCE_GENERIC routes bs=16 / bs=2 work through k_walk<G> and k_exh_generic (the kernels
    other block sizes use); they must agree with the specialised ones."""
    native, bbme, _, _ = mods
    import synth
    p, c = synth.frame(9, 0, 80, 112), synth.frame(9, 1, 80, 112)
    fast = {(sp, pn, bs): bbme.get_motion_field(p, c, bs, 8, sp, pn) for sp in range(4) for pn in range(2) for bs in (16, 2)}
    monkeypatch.setenv("GME_FORCE_GENERIC", "1")
    for (sp, pn, bs), want in fast.items():
        assert np.array_equal(bbme.get_motion_field(p, c, bs, 8, sp, pn), want), (sp, pn, bs)


def test_hierarchical_wrapper_and_results_flow(golden, mods, tmp_path, capsys):
    """bbme.hierarchical_wrapper, the results.py driver and its records."""
    import json
    _, bbme, _, utils = mods
    import results
    import synth
    from test_oracle import _gme_inputs
    g = golden("g8_next")
    for tag in ("small", "odd", "pan240"):
        if tag == "odd":
            p, c = synth.frame(79, 0, 150, 210), synth.frame(79, 1, 150, 210)
        else:
            p, c = _gme_inputs(golden, tag)
        bs, sw = (10, 4) if tag != "odd" else (6, 3)
        for sp in (3, 1, 0):
            err = str(g["hier_%s_sp%d_err" % (tag, sp)])
            if err:
                with pytest.raises(ValueError):
                    bbme.hierarchical_wrapper(p, c, block_size=bs, search_window=sw, searching_procedure=sp)
            else:
                got = bbme.hierarchical_wrapper(p, c, block_size=bs, search_window=sw, searching_procedure=sp)
                assert got.dtype == np.float64 and np.array_equal(got, g["hier_%s_sp%d" % (tag, sp)]), (tag, sp)
    # results.py flow: the JSON strings of psnr_records.json and the files it writes
    frames = list(synth.sequence(2000, 0, 6, 128, 192))
    want = json.loads(str(g["psnr_records_json"]))
    save = str(tmp_path) + "/"
    for sub in ("frames", "compensated", "curr_prev_diff", "model_motion_field", "curr_comp_diff"):
        (tmp_path / sub).mkdir()
    rec = results.process_frames(frames, 1, save)
    assert rec == {k: want[k] for k in "12345"}
    assert json.load(open(save + "psnr_records.json")) == rec
    from PIL import Image
    g7 = golden("g7_sequence")
    assert np.array_equal(np.array(Image.open(save + "compensated/-2.png")), g7["fd1_i3_comp"])   # idx-5 naming
    assert np.array_equal(np.array(Image.open(save + "frames/0.png")), frames[4])
    diff = np.abs(frames[2].astype(int) - g7["fd1_i2_comp"].astype(int)).astype(np.uint8)
    assert np.array_equal(np.array(Image.open(save + "curr_comp_diff/2.png")), diff)
    assert Image.open(save + "model_motion_field/5.png").size == (192, 128)
    assert results.process_frames(frames[:1], 1) == {}


def test_gme_block_size_24_follows_float32_order(mods):
    """The authors' figures use BBME_BLOCK_SIZE 24/32 (presentation/main.tex:426,558): MSE block
    distances then leave float32's exact range and the NumPy oracle (same float32 sums as the
    reference) is the yardstick."""
    _, _, motion, _ = mods
    import synth
    o = np_oracle()
    prev, cur = synth.frame(91, 0, 192, 256), synth.frame(91, 1, 192, 256)
    old = motion.BBME_BLOCK_SIZE
    motion.BBME_BLOCK_SIZE = 24
    try:
        got = motion.global_motion_estimation(prev, cur)
    finally:
        motion.BBME_BLOCK_SIZE = old
    want = o.global_motion_estimation(prev, cur, block_size=24)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12)


def test_successive_elimination_equals_brute_force(golden, mods, monkeypatch):
    """k_exh_sea16 prunes candidates by exact lower bounds; GME_EXH_BRUTE=1 selects the plain
    k_exh_qsad16.  Both must give the reference's field on real, synthetic and tie-heavy frames."""
    native, bbme, _, _ = mods
    import synth
    g3 = golden("g3_docframes")
    rng = np.random.default_rng(12)
    flat = np.full((80, 112), 128, np.uint8)
    quant = (rng.integers(0, 4, (96, 144)) * 64).astype(np.uint8)
    cases = [(g3["in_race_prev"], g3["in_race_cur"], 16), (g3["in_pan240_prev"], g3["in_pan240_cur"], 8),
             (synth.frame(5, 0, 270, 480), synth.frame(5, 3, 270, 480), 32), (flat, flat, 16),
             (quant, np.roll(quant, (3, -5), (0, 1)), 12), (synth.frame(6, 0, 50, 70), synth.frame(6, 1, 50, 70), 4)]
    sea = {pn: [bbme.get_motion_field(p, c, 16, sw, 0, pn) for p, c, sw in cases] for pn in (0, 1)}
    monkeypatch.setenv("GME_EXH_BRUTE", "1")           # k_exh_qsad16 / k_exh_dot16 instead
    co = c_oracle()
    for pn in (0, 1):
        for (p, c, sw), want in zip(cases, sea[pn]):
            assert np.array_equal(bbme.get_motion_field(p, c, 16, sw, 0, pn), want), (pn, sw)
        for (p, c, sw), want in list(zip(cases, sea[pn]))[2:]:
            assert np.array_equal(co.bbme(p, c, 16, sw, 0, pn), want), (pn, sw)


@pytest.mark.parametrize("pnorm", [0, 1])
def test_full_size_properties_1080p(mods, pnorm):
    """BASELINE-size checks that need no oracle run: on a 1920x1080 random texture a pure shift is
    recovered by every interior block, identical frames give the zero field, and the batched
    sequence path returns the same fields as the single-pair call."""
    native, bbme, _, _ = mods
    rng = np.random.default_rng(1080 + pnorm)
    canvas = rng.integers(0, 256, (1080 + 64, 1920 + 64), dtype=np.uint8)
    prev = np.ascontiguousarray(canvas[32:32 + 1080, 32:32 + 1920])
    dx, dy = 23, -17                                   # content moves +23 columns, -17 rows
    cur = np.ascontiguousarray(canvas[32 - dy:32 - dy + 1080, 32 - dx:32 - dx + 1920])
    mf = bbme.get_motion_field(prev, cur, 16, 32, 0, pnorm)
    assert mf.shape == (67, 120, 2)
    inner = mf[3:-3, 3:-3].reshape(-1, 2)
    assert (inner == (dx, dy)).all()
    assert (bbme.get_motion_field(prev, prev, 16, 32, 0, pnorm) == 0).all()
    for sp in (1, 2, 3):                               # walks start at the zero vector and stay there
        assert (bbme.get_motion_field(prev, prev, 16, 32, sp, pnorm)[1:-1, 1:-1] == 0).all(), sp
    seq = native.Sequence.from_frames(native.default_context(), [prev, cur, prev])
    seq.bbme(1, 16, 32, 0, pnorm)
    both = seq.read_mv()
    assert np.array_equal(both[0], mf)
    assert (both[1][3:-3, 3:-3].reshape(-1, 2) == (-dx, -dy)).all()
    seq.close()


def test_exhaustive_bs16_random_geometries(mods):
    """Randomised sweep of the two fast exhaustive kernels (successive elimination for MAE, dot4 for
    MSE) over frame sizes from one block up, windows 0..36 and tie-heavy content, vs the C oracle."""
    _, bbme, _, _ = mods
    co = c_oracle()
    rng = np.random.default_rng(2026)
    for trial in range(40):
        H, W = int(rng.integers(16, 150)), int(rng.integers(16, 260))
        sw = int(rng.choice([0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 5, 7]))
        kind = trial % 4
        if kind == 0:
            p, c = rng.integers(0, 256, (H, W), dtype=np.uint8), rng.integers(0, 256, (H, W), dtype=np.uint8)
        elif kind == 1:
            base = rng.integers(0, 256, (H + 40, W + 40), dtype=np.uint8)
            dy, dx = int(rng.integers(-9, 10)), int(rng.integers(-9, 10))
            p, c = base[20:20 + H, 20:20 + W], base[20 + dy:20 + dy + H, 20 + dx:20 + dx + W]
        elif kind == 2:
            p = (rng.integers(0, 3, (H, W)) * 100).astype(np.uint8)
            c = np.roll(p, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), (0, 1))
        else:
            p = np.full((H, W), int(rng.integers(0, 256)), np.uint8)
            c = p.copy()
        p, c = np.ascontiguousarray(p), np.ascontiguousarray(c)
        for pn in (0, 1):
            got = bbme.get_motion_field(p, c, 16, sw, 0, pn)
            assert np.array_equal(got, co.bbme(p, c, 16, sw, 0, pn)), (trial, H, W, sw, pn)


def test_walk_searches_random_geometries(mods):
    """Randomised sweep of the specialised walk kernels (bs = 16 with its LDS window cache, bs = 2
    dense) incl. long walks that leave and re-centre the cached window, vs the C oracle."""
    _, bbme, _, _ = mods
    co = c_oracle()
    rng = np.random.default_rng(77)
    for trial in range(30):
        H, W = int(rng.integers(17, 200)), int(rng.integers(17, 300))
        sw = int(rng.choice([2, 3, 7, 12, 16, 25, 40]))
        kind = trial % 3
        if kind == 0:       # smooth ramp + texture: walks run far (dozens of steps)
            yy, xx = np.mgrid[0:H + 120, 0:W + 120]
            base = ((yy * 3 + xx * 2) // 4 % 256).astype(np.uint8) ^ (rng.integers(0, 4, (H + 120, W + 120)).astype(np.uint8))
            dy, dx = int(rng.integers(-50, 51)), int(rng.integers(-50, 51))
            p, c = base[60:60 + H, 60:60 + W], base[60 + dy:60 + dy + H, 60 + dx:60 + dx + W]
        elif kind == 1:
            p, c = rng.integers(0, 256, (H, W), dtype=np.uint8), rng.integers(0, 256, (H, W), dtype=np.uint8)
        else:
            p = (rng.integers(0, 2, (H, W)) * 200).astype(np.uint8)
            c = np.roll(p, (int(rng.integers(-6, 7)), int(rng.integers(-6, 7))), (0, 1))
        p, c = np.ascontiguousarray(p), np.ascontiguousarray(c)
        for bs in (16, 2):
            if bs == 2 and trial % 5:
                continue
            for sp in (1, 2, 3):
                for pn in (0, 1):
                    got = bbme.get_motion_field(p, c, bs, sw, sp, pn)
                    assert np.array_equal(got, co.bbme(p, c, bs, sw, sp, pn)), (trial, H, W, bs, sw, sp, pn)
