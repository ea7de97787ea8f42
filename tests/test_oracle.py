"""Pin the oracles (oracle/gme_oracle.py, oracle/gme_oracle.c) to the golden vectors
that oracle/refimport/make_golden.py produced by importing the real reference.

CPU only.  The NumPy oracle has the reference's per-candidate cost, so it is run on
the small cases; the C oracle covers every golden, including 720x480 and 1080p.
"""
import re

import numpy as np
import pytest

from helpers import c_oracle, np_oracle, oracle_gme, oracle_results_flow, sha

G1_KEY = re.compile(r"mf_(\w+?_\d+x\d+)_bs(\d+)_sw(\d+)_sp(\d)_pn(\d)$")


def _g1_cases(golden):
    g = golden("g1_small")
    for k in g.files:
        m = G1_KEY.match(k)
        if m:
            name, bs, sw, sp, pn = m.group(1), *map(int, m.groups()[1:])
            yield k, g["in_%s_prev" % name], g["in_%s_cur" % name], bs, sw, sp, pn, g[k]


def test_c_oracle_small_fields(golden):
    co = c_oracle()
    n = 0
    for k, p, c, bs, sw, sp, pn, want in _g1_cases(golden):
        got = co.bbme(p, c, bs, sw, sp, pn)
        assert got.dtype == np.int32 and got.shape == want.shape
        assert np.array_equal(got, want), k
        n += 1
    assert n == 320


def test_numpy_oracle_small_fields(golden):
    o = np_oracle()
    n = 0
    for k, p, c, bs, sw, sp, pn, want in _g1_cases(golden):
        # exhaustive at (16,16)/(2,3) is the slow corner; keep a representative subset
        if sp == 0 and (bs, sw) in ((16, 16), (2, 3)) and "shifted" not in k:
            continue
        got = o.get_motion_field(p, c, block_size=bs, search_window=sw,
                                 searching_procedure=sp, pnorm_distance=pn)
        assert got.dtype == np.int32
        assert np.array_equal(got, want), k
        n += 1
    assert n >= 250


def test_c_oracle_synth720_all_searches(golden):
    import synth
    g = golden("g2_synth720")
    p, c = synth.frame(1234, 0, 480, 720), synth.frame(1234, 1, 480, 720)
    assert sha(p) == str(g["sha_prev"]) and sha(c) == str(g["sha_cur"])
    co = c_oracle()
    for sp in range(4):
        for pn in range(2):
            assert np.array_equal(co.bbme(p, c, 16, 16, sp, pn), g["mf_sp%d_pn%d" % (sp, pn)]), (sp, pn)
    # SURVEY §8(d) self-check: the background vector dominates the exhaustive MAE field
    mf = g["mf_sp0_pn0"].reshape(-1, 2)
    assert ((mf[:, 0] == 5) & (mf[:, 1] == -3)).sum() == 1205


def test_c_oracle_doc_frames(golden):
    g = golden("g3_docframes")
    co = c_oracle()
    for tag in ("race", "pan240"):
        p, c = g["in_%s_prev" % tag], g["in_%s_cur" % tag]
        for sp in range(4):
            for pn in range(2):
                assert np.array_equal(co.bbme(p, c, 16, 16, sp, pn),
                                      g["%s_mf_sp%d_pn%d" % (tag, sp, pn)]), (tag, sp, pn)


def test_numpy_oracle_doc_frames_fast_searches(golden):
    g = golden("g3_docframes")
    o = np_oracle()
    p, c = g["in_pan240_prev"], g["in_pan240_cur"]
    for sp in (1, 2, 3):
        got = o.get_motion_field(p, c, block_size=16, search_window=16, searching_procedure=sp,
                                 pnorm_distance=1)
        assert np.array_equal(got, g["pan240_mf_sp%d_pn1" % sp])


def _gme_inputs(golden, tag):
    import synth
    if tag == "synth720":
        return synth.frame(1234, 0, 480, 720), synth.frame(1234, 1, 480, 720)
    if tag == "small":
        return synth.frame(77, 3, 128, 192), synth.frame(77, 4, 128, 192)
    if tag == "bs12":
        return synth.frame(78, 0, 240, 320), synth.frame(78, 2, 240, 320)
    if tag == "dp":
        g = golden("g4_gme")
        return g["in_dp_prev"], g["in_dp_cur"]
    g = golden("g3_docframes")
    return g["in_%s_prev" % tag], g["in_%s_cur" % tag]


@pytest.mark.parametrize("tag", ["synth720", "race", "pan240", "dp", "small", "bs12"])
def test_c_oracle_gme_stages(golden, tag):
    """Every GME stage of the reference (motion.py:109-136) reproduced by the C oracle."""
    g = golden("g4_gme")
    co = c_oracle()
    bs = 12 if tag == "bs12" else 16
    prev, cur = _gme_inputs(golden, tag)
    pre = tag + "_"
    pp = [co.pyrdown(co.pyrdown(prev)), co.pyrdown(prev), prev]
    cp = [co.pyrdown(co.pyrdown(cur)), co.pyrdown(cur), cur]
    assert [sha(x) for x in pp + cp] == [str(s) for s in g[pre + "pyr_sha"]]
    dense = co.bbme(pp[0], cp[0], 2, 2, 3, 1)
    assert np.array_equal(dense, g[pre + "dense"])
    params = co.first_parameters(dense)
    assert params.dtype == np.float32 and np.array_equal(params, g[pre + "params0"])
    o = np_oracle()
    for lvl in (1, 2):
        params = o.project_parameters(params)
        pin = g[pre + "l%d_params_in" % lvl]
        # level 1 input is float32 arithmetic (exact); level 2 input went through the
        # LAPACK-build dependent 3x3 inverse: tolerance, then continue
        # from the golden input so later stages compare bit for bit
        if lvl == 1:
            assert params.dtype == pin.dtype == np.float32 and np.array_equal(params, pin)
        else:
            np.testing.assert_allclose(params, pin, rtol=1e-10, atol=1e-12)
        gt = co.bbme(pp[lvl], cp[lvl], bs, 2, 3, 1)
        assert np.array_equal(gt, g[pre + "l%d_gt" % lvl])
        st = co.fit_level(gt, pin, 0.3, pp[lvl].shape)
        assert np.array_equal(st["model"], g[pre + "l%d_model" % lvl])
        assert st["thr"] == int(g[pre + "l%d_thr" % lvl])
        assert np.array_equal(st["mask"], g[pre + "l%d_mask" % lvl])
        # normal-equation sums: bit patterns, not tolerances
        for k in ("F", "Sx", "Sy"):
            assert st[k].tobytes() == np.ascontiguousarray(g[pre + "l%d_%s" % (lvl, k)]).tobytes(), (lvl, k)
        params = o.solve_parameters(st["F"], st["Sx"], st["Sy"])
    np.testing.assert_allclose(params, g[pre + "params"], rtol=1e-10, atol=1e-12)
    shape = (prev.shape[0] // bs, prev.shape[1] // bs)
    field = co.affine_field(params, *shape)
    assert np.array_equal(field, g[pre + "field"])
    comp = co.compensate(prev, field.astype(np.int32))
    assert sha(comp) == str(g[pre + "comp_sha"])
    assert abs(o.psnr(cur, comp) - float(g[pre + "psnr"])) < 1e-12
    assert abs(o.psnr(cur, prev) - float(g[pre + "psnr_prev"])) < 1e-12


def test_numpy_oracle_gme_small(golden):
    """The NumPy oracle end to end (global_motion_estimation + compensation) on the small pair."""
    g = golden("g4_gme")
    o = np_oracle()
    prev, cur = _gme_inputs(golden, "small")
    stages = []
    params = o.global_motion_estimation(prev, cur, stages=stages)
    assert np.array_equal(stages[0]["dense"], g["small_dense"])
    assert np.array_equal(stages[0]["params0"], g["small_params0"])
    for lvl in (1, 2):
        st = stages[lvl]
        assert np.array_equal(st["gt"], g["small_l%d_gt" % lvl])
        assert np.array_equal(st["model"], g["small_l%d_model" % lvl])
        assert np.array_equal(st["mask"], g["small_l%d_mask" % lvl])
        assert st["thr"] == int(g["small_l%d_thr" % lvl])
        for k in ("F", "Sx", "Sy"):
            assert st[k].tobytes() == np.ascontiguousarray(g["small_l%d_%s" % (lvl, k)]).tobytes()
    np.testing.assert_allclose(params, g["small_params"], rtol=1e-10, atol=1e-12)
    comp = o.motion_compensation(prev, cur)
    assert np.array_equal(comp, g["small_comp"])


def test_rounding_margin_of_goldens(golden):
    """A 1e-13 LAPACK scatter may flip round() only if a model displacement sits within
    ~1e-12 of k+0.5; assert the goldens are far from that edge."""
    g = golden("g4_gme")
    for tag in ("synth720", "race", "pan240", "dp", "small", "bs12"):
        for key in ("l2_params_in", "params"):
            p = np.asarray(g["%s_%s" % (tag, key)], np.float64)
            h, w = g[tag + "_l2_gt"].shape[:2]
            i, j = np.mgrid[0:h, 0:w]
            for a in (0, 3):
                d = (p[a] + p[a + 1] * i) + p[a + 2] * j
                margin = np.abs(d - np.floor(d) - 0.5).min()
                assert margin > 1e-9, (tag, key, margin)


def test_oracle_affine_fields(golden):
    g = golden("g6_edges")
    co, o = c_oracle(), np_oracle()
    for k in range(24):
        p, want = g["aff_p_%d" % k], g["aff_f_%d" % k]
        assert want.dtype == np.int16
        assert np.array_equal(co.affine_field(p, *want.shape[:2]), want), k
        if want.shape[0] <= 30:
            assert np.array_equal(o.affine_field(want.shape, p), want), k
    assert np.array_equal(o.affine_field((4, 6, 2), g["aff_p_0"]), g["aff_f_tuple3"])


def test_oracle_compensate(golden):
    g = golden("g6_edges")
    co, o = c_oracle(), np_oracle()
    for k in range(5):
        f = g["comp_in_%d" % k]
        for t in ("16", "32"):
            mf, want = g["comp_mf%s_%d" % (t, k)], g["comp_out%s_%d" % (t, k)]
            assert np.array_equal(co.compensate(f, mf), want), (k, t)
            assert np.array_equal(o.compensate_frame(f, mf), want), (k, t)
    f = g["comp_in_narrow"]
    for t in ("narrow", "wide"):
        assert np.array_equal(co.compensate(f, g["comp_mf_" + t]), g["comp_out_" + t])
        assert np.array_equal(o.compensate_frame(f, g["comp_mf_" + t]), g["comp_out_" + t])


def test_oracle_first_parameters_projection_psnr(golden):
    g = golden("g6_edges")
    co, o = c_oracle(), np_oracle()
    for k in range(4):
        want = g["fp_out_%d" % k]
        assert np.array_equal(co.first_parameters(g["fp_in_%d" % k]), want)
        got = o.first_parameters(g["fp_in_%d" % k])
        assert got.dtype == np.float32 and np.array_equal(got, want)
    p32 = np.array([1.3, .1, .2, -2.7, .3, .4], np.float32)
    q = o.project_parameters(p32)
    assert q is p32 and q.dtype == np.float32 and np.array_equal(q, g["proj32"])
    assert abs(o.psnr(g["psnr_a"], g["psnr_b"]) - float(g["psnr_ab"])) < 1e-12
    assert o.psnr(g["psnr_a"], g["psnr_a"]) == -1 == float(g["psnr_aa"])
    n = g["psnr_a"].size
    mse = co.sse(g["psnr_a"], g["psnr_b"]) / n
    assert abs(20 * np.log10(255.0 / np.sqrt(mse)) - float(g["psnr_ab"])) < 1e-9


def test_oracle_degenerate_fits(golden):
    g = golden("g6_edges")
    o = np_oracle()
    for tag in ("n6", "n3", "n4"):
        p, c = g["fit_%s_prev" % tag], g["fit_%s_cur" % tag]
        pin = np.array([1.0, 0, 0, -1.0, 0, 0], np.float32)
        err = str(g["fit_%s_err" % tag])
        if err:
            assert err == "LinAlgError"
            with pytest.raises(np.linalg.LinAlgError):
                o.robust_fit(p, c, pin)
        else:
            np.testing.assert_allclose(o.robust_fit(p, c, pin), g["fit_%s_out" % tag], rtol=1e-9, atol=1e-11)


def test_oracle_search_corner_cases(golden):
    g = golden("g6_edges")
    g1 = golden("g1_small")
    co, o = c_oracle(), np_oracle()
    p, c = g1["in_shifted_64x96_prev"], g1["in_shifted_64x96_cur"]
    for sw in (1, 2, 3):
        assert np.array_equal(co.bbme(p, c, 8, sw, 2, 0), g["tdl_sw%d" % sw])
        assert np.array_equal(o.get_motion_field(p, c, 8, sw, 2, 0), g["tdl_sw%d" % sw])
    p, c = g1["in_random_50x70_prev"], g1["in_random_50x70_cur"]
    assert np.array_equal(co.bbme(p, c, 4, 20, 1, 1), g["tss_wild"])
    assert np.array_equal(o.get_motion_field(p, c, 4, 20, 1, 1), g["tss_wild"])
    assert np.array_equal(o.get_motion_field(p, c), g["tss_default"])
    assert np.array_equal(co.bbme(p, c, 4, 2, 1, 1), g["tss_default"])


def test_numpy_oracle_bs32_float32_costs(golden):
    """MSE at bs=32 leaves float32's exact-integer range; the NumPy oracle follows the
    reference there because it sums in float32 the same way (bbme.py:61-64,94)."""
    g = golden("g6_edges")
    o = np_oracle()
    for pre in ("bs32", "bs32hc"):
        sw = 4 if pre == "bs32" else 6
        for sp in (0, 3):
            got = o.get_motion_field(g[pre + "_prev"], g[pre + "_cur"], 32, sw, sp, 1)
            assert np.array_equal(got, g["%s_sp%d" % (pre, sp)]), (pre, sp)


def test_c_oracle_float32_order_costs(golden):
    """Outside float32's exact-integer range the C oracle emulates NumPy's pairwise float32 sum;
    it must agree with the real reference (goldens) and with NumPy itself on hard cases."""
    g = golden("g6_edges")
    co, o = c_oracle(), np_oracle()
    for pre, sw in (("bs32", 4), ("bs32hc", 6)):
        for sp in (0, 3):
            assert np.array_equal(co.bbme(g[pre + "_prev"], g[pre + "_cur"], 32, sw, sp, 1), g["%s_sp%d" % (pre, sp)]), (pre, sp)
    rng = np.random.default_rng(11)
    p = (rng.integers(0, 2, (60, 84)) * 255).astype(np.uint8)
    c = (rng.integers(0, 2, (60, 84)) * 255).astype(np.uint8)
    for bs, sw, sp in ((24, 3, 0), (17, 2, 0), (20, 4, 3), (28, 5, 1), (19, 6, 2)):
        want = o.get_motion_field(p, c, bs, sw, sp, 1)
        assert np.array_equal(co.bbme(p, c, bs, sw, sp, 1), want), (bs, sw, sp)
    # crafted near-ties where the summation order decides: the emulation follows the reference,
    # exact integer costs do not
    for k in range(3):
        cur, want = g["f32tie_cur_%d" % k], g["f32tie_mf_%d" % k]
        prev = np.zeros_like(cur)
        assert np.array_equal(co.bbme(prev, cur, 32, 1, 0, 1), want), k
        assert np.array_equal(o.get_motion_field(prev, cur, 32, 1, 0, 1), want), k
        assert not np.array_equal(co.bbme(prev, cur, 32, 1, 0, 1, allow_inexact=1), want), k


def test_numpy_oracle_sequence(golden):
    """results.py:41-112 flow on two pairs of the 6-frame sequence."""
    import synth
    g = golden("g7_sequence")
    o = np_oracle()
    frames = synth.sequence(2000, 0, 6, 128, 192)
    assert sha(frames) == str(g["frames_sha"])
    for fd, i in ((1, 1), (2, 5)):
        prev, cur = frames[i - fd], frames[i]
        params = o.global_motion_estimation(prev, cur)
        k = "fd%d_i%d_" % (fd, i)
        np.testing.assert_allclose(params, g[k + "params"], rtol=1e-10, atol=1e-12)
        field = o.affine_field((prev.shape[0] // 16, prev.shape[1] // 16), params)
        assert np.array_equal(field, g[k + "field"])
        comp = o.compensate_frame(prev, field)
        assert np.array_equal(comp, g[k + "comp"])
        assert abs(o.psnr(cur, comp) - float(g[k + "psnr"])) < 1e-12


@pytest.mark.slow
def test_c_oracle_1080p(golden):
    import synth
    g = golden("g5_1080p")
    co = c_oracle()
    p, c = synth.frame(4321, 0, 1080, 1920), synth.frame(4321, 1, 1080, 1920)
    assert sha(p) == str(g["sha_prev"]) and sha(c) == str(g["sha_cur"])
    assert np.array_equal(co.bbme(p, c, 16, 32, 0, 1), g["exh_mse_sw32"])
    pp = [co.pyrdown(co.pyrdown(p)), co.pyrdown(p), p]
    cp = [co.pyrdown(co.pyrdown(c)), co.pyrdown(c), c]
    assert np.array_equal(co.bbme(pp[0], cp[0], 2, 2, 3, 1), g["gme_dense"])
    for lvl in (1, 2):
        gt = co.bbme(pp[lvl], cp[lvl], 16, 2, 3, 1)
        assert np.array_equal(gt, g["gme_l%d_gt" % lvl])
        st = co.fit_level(gt, g["gme_l%d_params_in" % lvl], 0.3, pp[lvl].shape)
        assert np.array_equal(st["mask"], g["gme_l%d_mask" % lvl])
        for k in ("F", "Sx", "Sy"):
            assert st[k].tobytes() == np.ascontiguousarray(g["gme_l%d_%s" % (lvl, k)]).tobytes()


def test_c_oracle_pan240_sequence(golden):
    """The reference's own 51-frame real sequence (docs/assets/gifs/pan240) through its results.py flow
    (results.py:41-112) at the code default (bs 16, fd 1) and at the slides' setting (bs 12, fd 5,
    docs/presentation/main.tex:382): parameters, model field, compensated frame, PSNR per pair."""
    import json
    g = golden("g9_pan240seq")
    frames = g["frames"]
    assert frames.shape == (51, 240, 320) and frames.dtype == np.uint8
    for bs, fd in ((16, 1), (12, 5)):
        rec = json.loads(str(g["bs%d_fd%d_psnr_records_json" % (bs, fd)]))
        assert list(rec) == [str(i) for i in range(fd, 51)]
        for i in range(fd, 51):
            k = "bs%d_fd%d_i%d_" % (bs, fd, i)
            params, field, comp, psnr = oracle_results_flow(frames[i - fd], frames[i], bs)
            np.testing.assert_allclose(params, g[k + "params"], rtol=1e-10, atol=1e-12, err_msg=k)
            assert np.array_equal(field, g[k + "field"]), k
            assert sha(comp) == str(g[k + "comp_sha"]), k
            assert abs(psnr - complex(rec[str(i)]).real) < 1e-12, k
    co = c_oracle()
    for pn in (0, 1):       # BASELINE configs[0]: frame 10 vs 13, bs 16 sw 16 exhaustive
        assert np.array_equal(co.bbme(frames[10], frames[13], 16, 16, 0, pn), g["exh_10_13_pn%d" % pn]), pn


def test_rounding_margin_of_sequence_goldens(golden):
    """Same margin check as test_rounding_margin_of_goldens for the 96 real-content pairs."""
    g = golden("g9_pan240seq")
    worst = 1.0
    for bs, fd in ((16, 1), (12, 5)):
        h, w = 240 // bs, 320 // bs
        i, j = np.mgrid[0:h, 0:w]
        for idx in range(fd, 51):
            p = np.asarray(g["bs%d_fd%d_i%d_params" % (bs, fd, idx)], np.float64)
            for a in (0, 3):
                d = (p[a] + p[a + 2] * j) + p[a + 1] * i
                worst = min(worst, np.abs(d - np.floor(d) - 0.5).min())
    assert worst > 1e-9, worst


def test_c_oracle_gme1080exh(golden):
    """BASELINE configs[3] at full size: exhaustive MSE sw=32 fields at pyramid levels 1 and 2 of the
    1920x1080 pair, then the reference's own fit applied to them."""
    import synth
    g = golden("g10_extra")
    co = c_oracle()
    p, c = synth.frame(4321, 0, 1080, 1920), synth.frame(4321, 1, 1080, 1920)
    params, stages = oracle_gme(p, c, 0, 32)
    assert np.array_equal(stages[0]["dense"], g["gme1080exh_dense"])
    assert np.array_equal(stages[0]["gt"], g["gme1080exh_l1_exh_mse_sw32"])
    for lvl in (1, 2):
        pre = "gme1080exh_l%d_" % lvl
        st = stages[lvl - 1]
        assert np.array_equal(st["gt"], g[pre + "gt"]), lvl
        np.testing.assert_allclose(st["params_in"], g[pre + "params_in"], rtol=1e-10, atol=1e-12)
        st = co.fit_level(st["gt"], g[pre + "params_in"], 0.3, (1080 >> (2 - lvl), 1920 >> (2 - lvl)))
        assert np.array_equal(st["model"], g[pre + "model"]) and st["thr"] == int(g[pre + "thr"])
        assert np.array_equal(st["mask"], g[pre + "mask"])
        for k in ("F", "Sx", "Sy"):
            assert st[k].tobytes() == np.ascontiguousarray(g[pre + k]).tobytes(), (lvl, k)
    np.testing.assert_allclose(params, g["gme1080exh_params"], rtol=1e-10, atol=1e-12)
    field = co.affine_field(params, 67, 120)
    assert np.array_equal(field, g["gme1080exh_field"])
    comp = co.compensate(p, field.astype(np.int32))
    assert sha(comp) == str(g["gme1080exh_comp_sha"])
    assert abs(np_oracle().psnr(c, comp) - float(g["gme1080exh_psnr"])) < 1e-12


def test_oracle_unmasked_fit(golden):
    """motion.best_affine_parameters (motion.py:33-88): diamond MSE field, no mask, x = 4i, y = 4j."""
    import synth
    g = golden("g10_extra")
    g3 = golden("g3_docframes")
    co, o = c_oracle(), np_oracle()
    cases = {"small": (synth.frame(77, 3, 128, 192), synth.frame(77, 4, 128, 192)),
             "pan240": (g3["in_pan240_prev"], g3["in_pan240_cur"]),
             "synth720": (synth.frame(1234, 0, 480, 720), synth.frame(1234, 1, 480, 720))}
    for tag, (p, c) in cases.items():
        gt = co.bbme(p, c, 16, 2, 3, 1)
        F, Sx, Sy = o.normal_sums(gt, np.zeros(gt.shape[:2], bool), p.shape)
        np.testing.assert_allclose(o.solve_parameters(F, Sx, Sy), g["bap_" + tag], rtol=1e-10, atol=1e-12, err_msg=tag)


def test_c_and_numpy_oracles_agree_on_random_frames():
    """The two restatements against each other where no golden reaches: random small frames (sizes that are no
    multiples of the block size, pans, noise, flat areas with ties), every search, both norms, block sizes 2-16, windows
    0-6 -- the C port (the checker of every GPU test) and the NumPy port (the line-by-line restatement) must agree
    bit for bit; so must the rest of the GME chain on a random pair."""
    co, o = c_oracle(), np_oracle()
    rng = np.random.default_rng(20261004)
    n = 0
    for case in range(60):
        H, W = int(rng.integers(17, 56)), int(rng.integers(17, 72))
        kind = case % 4
        if kind == 0:
            base = rng.integers(0, 256, (H + 12, W + 12), dtype=np.uint8)
            dy, dx = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
            p, c = base[6:6 + H, 6:6 + W], base[6 + dy:6 + dy + H, 6 + dx:6 + dx + W]
        elif kind == 1:
            p, c = rng.integers(0, 256, (2, H, W), dtype=np.uint8)
        elif kind == 2:
            p = np.full((H, W), 90, np.uint8)
            c = p.copy()
            c[H // 3:H // 2, W // 4:W // 2] = rng.integers(0, 256, (H // 2 - H // 3, W // 2 - W // 4), dtype=np.uint8)
        else:
            p = (rng.integers(0, 3, (H, W)) * 100).astype(np.uint8)
            c = np.roll(p, (1, -2), axis=(0, 1))
        p, c = np.ascontiguousarray(p), np.ascontiguousarray(c)
        bs = int(rng.choice([2, 4, 8, 16]))
        sw = int(rng.integers(0, 7))
        for proc in range(4):
            if proc == 3 and (H < bs + 1 or W < bs + 1):
                continue
            for pn in (0, 1):
                got = co.bbme(p, c, bs, sw, proc, pn)
                want = o.get_motion_field(p, c, block_size=bs, search_window=sw, searching_procedure=proc, pnorm_distance=pn)
                assert np.array_equal(got, want), (case, H, W, bs, sw, proc, pn)
                n += 1
    assert n >= 400
    for case in range(4):
        H, W = int(rng.integers(70, 110)), int(rng.integers(70, 130))
        base = rng.integers(0, 256, (H + 8, W + 8), dtype=np.uint8)
        p, c = np.ascontiguousarray(base[4:4 + H, 4:4 + W]), np.ascontiguousarray(base[3:3 + H, 5:5 + W])
        assert np.array_equal(co.pyrdown(p), o.pyr_down(p))
        want = o.global_motion_estimation(p, c)
        from helpers import oracle_gme
        got, _ = oracle_gme(p, c)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
        field = co.affine_field(got, H // 16, W // 16)
        assert np.array_equal(field, o.affine_field((H // 16, W // 16), got))
        assert np.array_equal(co.compensate(p, field.astype(np.int32)), o.compensate_frame(p, field))
