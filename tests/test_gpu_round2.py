"""Parity of the kernel instances and paths bench.py actually times, at the sizes it times them
(BASELINE configs[1], [3], [4]); the reference's real 51-frame sequence; the CLI mains; thread
safety and the chunked launches.  Needs an MI355X.  Same bars as test_gpu_parity.py.
"""
import argparse
import ctypes
import json
import os
import threading

import numpy as np
import pytest

from helpers import c_oracle, np_oracle, oracle_gme, oracle_results_flow, sha

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    import _gme_native
    ctx = _gme_native.default_context()
    assert "gfx950" in ctx.info()["name"]
    return _gme_native


def _contents(kind, n, H, W, rng):
    """uint8[n, H, W] frame stacks that stress the elimination bound differently."""
    if kind == "noise":            # nothing correlates: (almost) every candidate patch survives the bound
        return rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    if kind == "flat":             # every cost ties at 0: the first candidate in scan order must win
        return np.full((n, H, W), 93, np.uint8)
    if kind == "mixed":            # left half uncorrelated noise, right half a clean pan: hostile and friendly tiles side by side
        import synth
        f = synth.sequence(1234, 0, n, H, W).copy()
        f[:, :, :W // 2] = rng.integers(0, 256, (n, H, W // 2), dtype=np.uint8)
        return f
    if kind == "steps":            # few grey levels: many exact ties between real candidates
        base = (rng.integers(0, 3, (H + 64, W + 64)) * 90).astype(np.uint8)
        return np.stack([base[16 + 2 * (t % 5):16 + 2 * (t % 5) + H, 24 - 3 * (t % 4):24 - 3 * (t % 4) + W] for t in range(n)])
    raise KeyError(kind)


@pytest.mark.parametrize("pnorm", [0, 1])
def test_benched_instance_720_every_pair(native, pnorm):
    """BASELINE configs[1] as bench.py launches it: a batch of 720x480 pairs (67: not a multiple of the
    8 XCDs) goes to the persistent k_exh_sea16p<3,.> (2x4 tiles, ragged last tile column, dynamic
    per-XCD tile counters) -- asserted through the launch plan -- and EVERY pair equals the C oracle."""
    ctx = native.default_context()
    co = c_oracle()
    seq = native.Sequence(ctx, 68, 480, 720)
    seq.synth(1234, 0)
    if pnorm == 1:          # the default MSE path at sw 16 is the matrix-core kernel (tests/test_gpu_mfma.py); this test holds the
        os.environ["GME_EXH_MFMA"] = "0"         # vector-unit elimination instance to the same bar
    try:
        seq.bbme(1, 16, 16, 0, pnorm)
    finally:
        os.environ.pop("GME_EXH_MFMA", None)
    info = ctx.last_bbme_info()
    assert info["plan"].startswith("k_exh_sea16p%s<3," % ("_mse" if pnorm else "")), info
    assert "persistent-dynamic" in info["plan"] and "tiles 2x4" in info["plan"], info
    assert info["patches"] == 67 * 1350 * 64 * 3 and 0 < info["surviving"] < info["patches"] // 4, info
    mv = seq.read_mv()
    frames = [seq.read_frame(i) for i in range(68)]
    for p in range(67):
        assert np.array_equal(mv[p], co.bbme(frames[p], frames[p + 1], 16, 16, 0, pnorm)), p
    seq.close()


@pytest.mark.parametrize("kind", ["noise", "flat", "steps", "mixed"])
def test_benched_instance_720_hostile_content(native, kind, monkeypatch):
    """The same kernel instance on content where the bound prunes little (noise: those tiles go to the
    brute-force redo kernel), where every cost ties (flat: the key rule prunes everything behind the
    first candidate), with many exact ties (steps) and with hostile and friendly tiles side by side
    (mixed), both norms, every pair against the C oracle -- with and without the redo path."""
    ctx = native.default_context()
    co = c_oracle()
    monkeypatch.setenv("GME_EXH_MFMA", "0")          # the elimination + redo path under both norms (MSE's default at sw 16 is k_exh_mfma16)
    rng = np.random.default_rng(7 + len(kind))
    frames = _contents(kind, 22, 480, 720, rng)
    seq = native.Sequence.from_frames(ctx, frames)
    want = {pn: [co.bbme(frames[p], frames[p + 1], 16, 16, 0, pn) for p in range(21)] for pn in (0, 1)}
    for redo in ("1", "0"):
        monkeypatch.setenv("GME_SEA_REDO", redo)
        for pnorm in (0, 1):
            seq.bbme(1, 16, 16, 0, pnorm)
            info = ctx.last_bbme_info()
            assert info["plan"].startswith("k_exh_sea16p"), info
            mv = seq.read_mv()
            for p in range(21):
                assert np.array_equal(mv[p], want[pnorm][p]), (kind, redo, pnorm, p)
            if redo == "1" and kind == "noise":
                assert info["redo_tiles"] > 0.8 * 21 * 180, info
            if redo == "1" and kind == "mixed":
                assert 0.2 * 21 * 180 < info["redo_tiles"] < 0.8 * 21 * 180, info
            if redo == "0":
                assert info["redo_tiles"] == 0, info
            if kind == "flat":
                assert info["surviving"] < 0.02 * info["patches"] and info["redo_tiles"] == 0, info
    seq.close()


def test_redo_path_small_batches_and_window_sizes(native, monkeypatch):
    """The redo hand-off from the one-tile kernels (small batches) and from every window size class, forced
    for every tile (GME_SEA_REDO_FRAC=0: any listed patch makes a tile hostile) on ragged geometries."""
    ctx = native.default_context()
    co = c_oracle()
    monkeypatch.setenv("GME_SEA_REDO_FRAC", "0")
    monkeypatch.setenv("GME_EXH_MFMA", "0")
    for persist in ("0", "2"):
        monkeypatch.setenv("GME_SEA_PERSIST", persist)
        for (n, h, w, sw, seed) in ((5, 96, 176, 16, 3), (3, 70, 330, 8, 4), (10, 50, 66, 4, 5), (3, 130, 150, 32, 6), (4, 80, 112, 24, 7)):
            seq = native.Sequence(ctx, n, h, w)
            seq.synth(seed, 0)
            frames = [seq.read_frame(i) for i in range(n)]
            for pn in (0, 1):
                seq.bbme(1, 16, sw, 0, pn)
                info = ctx.last_bbme_info()
                assert info["redo_tiles"] > 0, (persist, info)
                mv = seq.read_mv()
                for p in range(n - 1):
                    assert np.array_equal(mv[p], co.bbme(frames[p], frames[p + 1], 16, sw, 0, pn)), (persist, n, h, w, sw, pn, p)
            seq.close()


def test_pyramids_odd_and_tiny_shapes(native):
    """k_pyrdown (two output rows per thread, seven row loads), k_pyrdown_edge16 (16-byte row chunks) and the per-pixel
    border kernel on shapes with odd heights / widths, widths that are no multiple of 4, fewer than 16 columns and
    fewer than 4 rows: every level equals the C oracle's restatement of cv2.pyrDown (utils.py:34-51)."""
    import utils
    co = c_oracle()
    rng = np.random.default_rng(5)
    for H, W in [(17, 17), (33, 47), (5, 9), (16, 20), (101, 203), (4, 16), (3, 40), (2, 2), (1, 7), (64, 66), (27, 128), (480, 722), (75, 1000)]:
        f = rng.integers(0, 256, (H, W), dtype=np.uint8)
        pyr = utils.get_pyramids(f)
        l1 = co.pyrdown(f)
        assert np.array_equal(pyr[2], f) and np.array_equal(pyr[1], l1), (H, W)
        assert np.array_equal(pyr[0], co.pyrdown(l1)), (H, W)


@pytest.mark.parametrize("proc", [1, 2, 3])
def test_walk16_random_shapes(native, proc):
    """k_walk16 (8 blocks per wave, prefetched through buffer resources) on shapes its grid does not divide evenly:
    frames barely larger than a block, last block rows / columns that start on max + 1, windows that touch all four
    frame edges, fields of 1 .. 33 blocks per workgroup; pan, noise, flat and half-noise content, both norms,
    frame distances 1-3 -- every pair against the C oracle, bit-exact (tools/soak.py ... walk runs the same for minutes)."""
    co = c_oracle()
    rng = np.random.default_rng(100 + proc)
    ctx = native.default_context()
    shapes = [(17, 17), (20, 305), (32, 32), (33, 47), (48, 64), (64, 96), (100, 130), (144, 176), (219, 339), (16 * 9, 16 * 11 + 5)]
    checked = 0
    for H, W in shapes:
        for kind in ("pan", "noise", "flat", "mixed"):
            n = int(rng.integers(3, 9))
            if kind == "pan":
                base = rng.integers(0, 256, (H + 80, W + 80), dtype=np.uint8)
                frames = np.stack([base[40 + 2 * (t % 7):40 + 2 * (t % 7) + H, 40 - 3 * (t % 5):40 - 3 * (t % 5) + W] for t in range(n)])
            elif kind == "noise":
                frames = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
            elif kind == "flat":
                frames = np.full((n, H, W), 77, np.uint8)
            else:
                base = rng.integers(0, 256, (H + 80, W + 80), dtype=np.uint8)
                frames = np.stack([base[40 + (t % 5):40 + (t % 5) + H, 40 - 2 * (t % 6):40 - 2 * (t % 6) + W] for t in range(n)]).copy()
                frames[:, :, :W // 2] = rng.integers(0, 256, (n, H, W // 2), dtype=np.uint8)
            frames = np.ascontiguousarray(frames)
            seq = native.Sequence.from_frames(ctx, frames)
            try:
                fd = int(rng.integers(1, min(3, n - 1) + 1))
                sw = int(rng.choice([4, 8, 16, 32]))
                for pn in (0, 1):
                    seq.bbme(fd, 16, sw, proc, pn)
                    mv = seq.read_mv()
                    assert "k_walk16" in ctx.last_bbme_info()["plan"]
                    for p in range(n - fd):
                        want = co.bbme(frames[p], frames[p + fd], 16, sw, proc, pn)
                        assert np.array_equal(mv[p], want), (H, W, kind, fd, sw, proc, pn, p)
                        checked += 1
            finally:
                seq.close()
    assert checked >= 200


@pytest.mark.parametrize("pnorm", [0, 1])
def test_benched_instance_1080p_every_pair(native, pnorm):
    """BASELINE configs[3] BBME as benched: 1920x1080, sw = 32, 9 pairs -> persistent k_exh_sea16p<5,.>
    with 2x6 tiles; every pair against the C oracle (not against another HIP kernel)."""
    ctx = native.default_context()
    co = c_oracle()
    seq = native.Sequence(ctx, 10, 1080, 1920)
    seq.synth(4321, 0)
    seq.bbme(1, 16, 32, 0, pnorm)
    info = ctx.last_bbme_info()
    assert info["plan"].startswith("k_exh_sea16p%s<5," % ("_mse" if pnorm else "")) and "persistent" in info["plan"], info
    mv = seq.read_mv()
    frames = [seq.read_frame(i) for i in range(10)]
    for p in range(9):
        assert np.array_equal(mv[p], co.bbme(frames[p], frames[p + 1], 16, 32, 0, pnorm)), p
    seq.close()


def test_gme1080exh_stages_vs_reference(golden, native):
    """BASELINE configs[3] at full size against the REFERENCE: exhaustive-MSE (sw 32) fields at pyramid
    levels 1 and 2, the reference's own fit applied to them (make_golden.py g10)."""
    import motion
    import synth
    g = golden("g10_extra")
    p, c = synth.frame(4321, 0, 1080, 1920), synth.frame(4321, 1, 1080, 1920)
    seq = native.Sequence.from_frames(native.default_context(), [p, c])
    p0 = seq.gme_begin(1, 16, 0, 32)
    assert np.array_equal(seq.gme_read_stage(0, 0)["gt"], g["gme1080exh_dense"])
    assert np.array_equal(p0[0], g["gme1080exh_params0"])
    for lvl in (1, 2):
        pre = "gme1080exh_l%d_" % lvl
        sums = seq.gme_fit(lvl, np.asarray(g[pre + "params_in"], np.float64)[None], 0.3)[0]
        st = seq.gme_read_stage(lvl, 0)
        assert np.array_equal(st["gt"], g[pre + "gt"]), lvl
        assert np.array_equal(st["model"], g[pre + "model"]) and st["thr"] == int(g[pre + "thr"]), lvl
        assert np.array_equal(st["mask"], g[pre + "mask"]), lvl
        want = np.concatenate([g[pre + "F"].reshape(9), g[pre + "Sx"], g[pre + "Sy"]])
        assert sums.tobytes() == want.tobytes(), lvl
    assert np.array_equal(seq.gme_read_stage(1, 0)["gt"], g["gme1080exh_l1_exh_mse_sw32"])
    params = motion.estimate_sequence(seq, 1, 0, 32)
    np.testing.assert_allclose(params[0], g["gme1080exh_params"], rtol=1e-10, atol=1e-12)
    sse = seq.compensate(1, 16, params)
    assert sha(seq.read_compensated(0)) == str(g["gme1080exh_comp_sha"])
    assert abs(20 * np.log10(255.0 / np.sqrt(int(sse[0]) / p.size)) - float(g["gme1080exh_psnr"])) < 1e-9
    seq.close()


def test_sharded_sequence_1080p_shard_vs_oracle(native):
    """BASELINE configs[4] shape: a 1080p diamond-GME shard with 3 streams per GPU and 2 virtual ranks;
    every pair's parameters, compensated frame and PSNR against the C-oracle chain."""
    import sequence
    n = 13
    got = []
    for r in range(2):
        sh = sequence.ShardedSequence(1080, 1920, n, 1, rank=r, world=2, streams=3)
        sh.synth(2000)
        params, psnr = sh.estimate_and_compensate()
        comps = [sh.read_compensated(k) for k in range(sh.n_pairs)]
        got.append((sh.pair_start, params, psnr, comps))
        sh.close()
    import synth
    frames = synth.sequence(2000, 0, n, 1080, 1920)
    for start, params, psnr, comps in got:
        for k in range(len(params)):
            p = start + k
            wp, _, wc, wpsnr = oracle_results_flow(frames[p], frames[p + 1])
            np.testing.assert_allclose(params[k], wp, rtol=1e-10, atol=1e-12, err_msg=str(p))
            assert np.array_equal(comps[k], wc), p
            assert abs(psnr[k] - wpsnr) < 1e-9, p
    assert sum(len(g[1]) for g in got) == n - 1


def test_interleaved_streams_equal_blocking_calls(native):
    """Split-phase calls (gme_seq_set_split_phase / gme_seq_wait): three pair ranges driven by ONE host thread give
    bit for bit the parameters, PSNR and compensated frames of the blocking single-stream path -- twice in a row on
    the same buffers, with ranges of unequal length -- and the C-oracle chain agrees on sampled pairs."""
    import sequence
    import synth
    n, H, W = 47, 240, 352
    one = sequence.ShardedSequence(H, W, n, 1, streams=1)
    one.synth(77)
    want_p, want_psnr = one.estimate_and_compensate()
    want_c = [one.read_compensated(k) for k in (0, 17, n - 2)]
    one.close()
    il = sequence.ShardedSequence(H, W, n, 1, streams=3, interleave=True)
    il.synth(77)
    for _ in range(2):
        got_p, got_psnr = il.estimate_and_compensate()
        assert np.array_equal(got_p, want_p) and np.array_equal(got_psnr, want_psnr)
        for k, c in zip((0, 17, n - 2), want_c):
            assert np.array_equal(il.read_compensated(k), c), k
    assert np.array_equal(il.estimate(), want_p)                        # estimate() alone takes the same route
    # ... and the blocking calls work again afterwards (compensate() evaluates the PSNR per pair with cmath, the batch call
    # vectorised: 2e-14 dB apart at most, sequence._psnr)
    np.testing.assert_allclose(il.compensate(want_p), want_psnr, rtol=0, atol=1e-12)
    lane = il.lanes[0]
    assert not lane.seq._split
    p = __import__("motion").estimate_sequence(lane.seq, 1)
    assert np.array_equal(p[:lane.hi - lane.lo], want_p[lane.lo:lane.hi])
    il.close()
    frames = synth.sequence(77, 0, n, H, W)
    for p in (0, 23, n - 2):
        wp, _, wc, wpsnr = oracle_results_flow(frames[p], frames[p + 1])
        np.testing.assert_allclose(want_p[p], wp, rtol=1e-10, atol=1e-12, err_msg=str(p))
        assert abs(want_psnr[p] - wpsnr) < 1e-9, p
    # waiting without the switch is refused
    seq = native.Sequence(native.default_context(), 3, 64, 64)
    with pytest.raises(native.GmeError):
        seq.wait()
    seq.close()


@pytest.mark.parametrize("bs,fd", [(16, 1), (12, 5)])
def test_pan240_sequence_vs_reference(golden, native, bs, fd, capsys, tmp_path):
    """The reference's real 51-frame sequence through the results.py flow (results.py:41-112) at the code
    default (bs 16, fd 1) and the slides' setting (bs 12, fd 5, docs/presentation/main.tex:382):
    parameters, compensated frames, psnr_records strings and some_data's summary equal the reference's."""
    import motion
    import results
    import sequence
    import utils
    g = golden("g9_pan240seq")
    frames = g["frames"]
    old = motion.BBME_BLOCK_SIZE
    motion.BBME_BLOCK_SIZE = bs
    try:
        rec = results.process_frames(list(frames), fd)
        want = json.loads(str(g["bs%d_fd%d_psnr_records_json" % (bs, fd)]))
        assert rec == want
        for streams, world in ((1, 1), (3, 1), (1, 2)):
            parts = []
            for r in range(world):
                sh = sequence.ShardedSequence(240, 320, 51, fd, rank=r, world=world, streams=streams)
                sh.load(frames)
                params = sh.estimate()
                sh.compensate(params)
                for k in range(sh.n_pairs):
                    i = sh.pair_start + k + fd
                    key = "bs%d_fd%d_i%d_" % (bs, fd, i)
                    np.testing.assert_allclose(params[k], g[key + "params"], rtol=1e-10, atol=1e-12, err_msg=key)
                    assert sha(sh.read_compensated(k)) == str(g[key + "comp_sha"]), key
                    field = motion.get_motion_field_affine((int(240 / bs), int(320 / bs), 2), params[k])
                    assert np.array_equal(field, g[key + "field"]), key
                parts.append(sh.n_pairs)
                sh.close()
            assert sum(parts) == 51 - fd
    finally:
        motion.BBME_BLOCK_SIZE = old
    path = tmp_path / "psnr_records.json"
    path.write_text(json.dumps(rec))
    capsys.readouterr()
    utils.some_data(str(path))
    assert capsys.readouterr().out == str(g["bs%d_fd%d_some_data_stdout" % (bs, fd)])


def test_configs0_standin_exhaustive(golden, native):
    """BASELINE configs[0] (frame 10 vs 13, bs 16 sw 16 exhaustive) on the decodable frames of the same video."""
    import bbme
    g = golden("g9_pan240seq")
    f = g["frames"]
    for pn in (0, 1):
        assert np.array_equal(bbme.get_motion_field(f[10], f[13], 16, 16, 0, pn), g["exh_10_13_pn%d" % pn]), pn


def test_unmasked_fit_vs_reference(golden, native):
    """motion.best_affine_parameters (motion.py:33-88) against the reference's output."""
    import motion
    import synth
    g, g3 = golden("g10_extra"), golden("g3_docframes")
    cases = {"small": (synth.frame(77, 3, 128, 192), synth.frame(77, 4, 128, 192)),
             "pan240": (g3["in_pan240_prev"], g3["in_pan240_cur"]),
             "synth720": (synth.frame(1234, 0, 480, 720), synth.frame(1234, 1, 480, 720))}
    for tag, (p, c) in cases.items():
        got = motion.best_affine_parameters(p, c)
        assert got.dtype == np.float64 and got.shape == (6,)
        np.testing.assert_allclose(got, g["bap_" + tag], rtol=1e-10, atol=1e-12, err_msg=tag)


def test_two_threads_share_the_module_api(native):
    """bbme.get_motion_field / motion.* share one default context (one stream, one scratch
    buffer) and ctypes releases the GIL; concurrent callers must still get the oracle's answers."""
    import bbme
    import motion
    import synth
    co = c_oracle()
    jobs = []
    for k in range(6):
        h, w = (96 + 16 * k, 160 + 32 * (k % 3))
        p, c = synth.frame(100 + k, 0, h, w), synth.frame(100 + k, 1, h, w)
        jobs.append((p, c, (k % 4), (k % 2)))
    want = [co.bbme(p, c, 16, 8, sp, pn) for p, c, sp, pn in jobs]
    want_gme = [oracle_gme(p, c)[0] for p, c, _, _ in jobs]
    errors = []

    def worker(tid):
        try:
            for rep in range(6):
                for k in range(tid, len(jobs), 2):
                    p, c, sp, pn = jobs[k]
                    assert np.array_equal(bbme.get_motion_field(p, c, 16, 8, sp, pn), want[k]), (tid, rep, k)
                    np.testing.assert_allclose(motion.global_motion_estimation(p, c), want_gme[k], rtol=1e-10, atol=1e-12)
        except Exception as e:      # noqa: BLE001 - reported below
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_bad_arguments_fail_loudly(native):
    """Zero / negative block sizes and wrong shapes give error codes or the reference's exception, never a signal."""
    import bbme
    import motion
    ctx = native.default_context()
    lib = ctx.lib
    seq = native.Sequence(ctx, 3, 64, 96)
    seq.synth(3, 0)
    params = np.zeros((2, 6))
    pp = params.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    for bs in (0, -4):
        assert lib.gme_seq_compensate(seq.handle, 1, bs, pp, None) == native.ERR_ARG
        assert lib.gme_seq_bbme(seq.handle, 1, bs, 2, 0, 0) == native.ERR_ARG
        assert lib.gme_seq_gme_begin(seq.handle, 1, bs, 3, 2, None) == native.ERR_ARG
    assert lib.gme_seq_compensate(seq.handle, 1, 128, pp, None) == native.ERR_GEOMETRY
    assert lib.gme_seq_compensate(seq.handle, 3, 16, pp, None) == native.ERR_ARG          # fd >= N
    assert b"frame_distance" in lib.gme_last_error()
    f = seq.read_frame(0)
    with pytest.raises(ZeroDivisionError):            # bbme.py:23: int(H / 0)
        bbme.get_motion_field(f, f, block_size=0)
    old = motion.BBME_BLOCK_SIZE
    motion.BBME_BLOCK_SIZE = 0
    try:
        with pytest.raises(ZeroDivisionError):
            motion.global_motion_estimation(f, f)
        with pytest.raises(ZeroDivisionError):
            motion.motion_compensation(f, f)
    finally:
        motion.BBME_BLOCK_SIZE = old
    seq.close()


@pytest.mark.parametrize("proc,pnorm", [(0, 0), (0, 1), (3, 1), (1, 0)])
def test_chunked_bbme_launches(native, monkeypatch, proc, pnorm):
    """launch_bbme splits long sequences into several launches (2^24 blocks each; GME_BBME_CHUNK_BLOCKS
    lowers the limit): per-chunk offsets of the fields, the box-sum table and the frame planes, and the
    persistent kernels' pair numbering restarting per chunk, 11 pairs (not a multiple of 8) in 4 chunks."""
    ctx = native.default_context()
    co = c_oracle()
    seq = native.Sequence(ctx, 12, 96, 176)
    seq.synth(17, 3)
    frames = [seq.read_frame(i) for i in range(12)]
    monkeypatch.setenv("GME_BBME_CHUNK_BLOCKS", str(66 * 3))          # 66 blocks per pair -> 3 pairs per launch
    monkeypatch.setenv("GME_SEA_PERSIST", "2")
    seq.bbme(1, 16, 8, proc, pnorm)
    mv = seq.read_mv()
    for p in range(11):
        assert np.array_equal(mv[p], co.bbme(frames[p], frames[p + 1], 16, 8, proc, pnorm)), p
    seq.close()


def test_chunked_plane_launches(native, monkeypatch):
    """grid.z / grid.y carry pairs or frames in k_pyrdown, k_compensate*, k_sse, k_affine_field, k_sqbox16
    (hardware limit 65535; GME_MAX_GRID_PAIRS lowers the chunk size): chunked == unchunked == oracle."""
    import motion
    import sequence
    import synth
    frames = synth.sequence(23, 0, 9, 112, 176)
    sh = sequence.ShardedSequence(112, 176, 9, 1)
    sh.load(frames)
    want, want_psnr = sh.estimate_and_compensate(exact_psnr=True)
    want_comp = [sh.read_compensated(k) for k in range(8)]
    want_mv = sh.motion_fields(16, 8, 0, 1)
    sh.close()
    monkeypatch.setenv("GME_MAX_GRID_PAIRS", "3")
    sh = sequence.ShardedSequence(112, 176, 9, 1)
    sh.load(frames)
    got, got_psnr = sh.estimate_and_compensate(exact_psnr=True)
    assert np.array_equal(got, want) and np.array_equal(got_psnr, want_psnr)
    for k in range(8):
        assert np.array_equal(sh.read_compensated(k), want_comp[k]), k
    assert np.array_equal(sh.motion_fields(16, 8, 0, 1), want_mv)
    sh.close()
    wp, _, wc, wpsnr = oracle_results_flow(frames[7], frames[8])
    np.testing.assert_allclose(got[7], wp, rtol=1e-10, atol=1e-12)
    assert np.array_equal(want_comp[7], wc) and abs(got_psnr[7] - wpsnr) < 1e-9
    field = motion.get_motion_field_affine((7, 11), wp)
    assert np.array_equal(field, c_oracle().affine_field(wp, 7, 11))


def test_cli_mains_run(golden, native, tmp_path, monkeypatch, capsys):
    """bbme.main (bbme.py:617-649) and results.main (results.py:14-112) executed on frame directories."""
    import bbme
    import results
    import synth
    from PIL import Image
    from test_oracle import _gme_inputs
    g8, g7 = golden("g8_next"), golden("g7_sequence")
    monkeypatch.chdir(tmp_path)
    # ---- bbme.main: frames[fi - 3] vs frames[fi]
    p, c = _gme_inputs(golden, "small")
    vdir = tmp_path / "clipA"
    vdir.mkdir()
    filler = synth.frame(1, 0, 128, 192)
    for i, f in enumerate((p, filler, filler, c)):
        Image.fromarray(f).save(str(vdir / ("%d.png" % i)))
    args = argparse.Namespace(path=str(vdir), fi=3, pnorm=0, block_size=10, search_window=4, searching_procedure=3)
    mf, mf_h = bbme.main(args)
    assert np.array_equal(mf, c_oracle().bbme(p, c, 10, 4, 3, 1))            # -pn is parsed but never forwarded: MSE
    assert np.array_equal(mf_h, g8["hier_small_sp3"])
    for name in ("3-res.png", "3h-res.png"):
        assert Image.open(str(tmp_path / "resources" / "images" / name)).size == (192, 128)
    # ---- results.main: resources/videos/<name>, results/<name>/...
    frames = synth.sequence(2000, 0, 6, 128, 192)
    rdir = tmp_path / "resources" / "videos" / "clipB"
    rdir.mkdir(parents=True)
    for i, f in enumerate(frames):
        Image.fromarray(f).save(str(rdir / ("frame%02d.png" % i)))
    rec = results.main(argparse.Namespace(path="clipB", fd="1"))
    want = json.loads(str(g8["psnr_records_json"]))
    assert rec == {k: want[k] for k in "12345"}
    out = tmp_path / "results" / "clipB"
    assert json.load(open(str(out / "psnr_records.json"))) == rec
    assert np.array_equal(np.array(Image.open(str(out / "compensated" / "-2.png"))), g7["fd1_i3_comp"])
    assert "frame shape: (128, 192)" in capsys.readouterr().out
    assert results.main(argparse.Namespace(path="clipB", fd=None)) == rec     # upstream crashes without -f; the default is 1


def test_rccl_self_gather_world_1(native):
    """gme_comm_* / gme_shard_gather (the C ABI's own RCCL path) with one rank: ncclCommInitRank,
    ncclAllGather of float64 rows on the context's stream, ncclAllReduce(max) as barrier."""
    import sequence
    ctx = native.Context(0)
    sequence.comm_init(ctx, 0, 1)
    rows = np.arange(7 * 7, dtype=np.float64).reshape(7, 7) * 0.37 - 3
    got = sequence.gather_parameters_rccl(ctx, rows, 7, 0, 1)
    assert got.dtype == np.float64 and np.array_equal(got, rows)
    assert sequence.gather_parameters_rccl(ctx, np.zeros((0, 6)), 0, 0, 1).shape == (0, 6)
    assert sequence.comm_max(ctx, 2.5) == 2.5
    sequence.comm_barrier(ctx)
    assert sequence.comm_info(ctx) == (0, 1)                # what ncclCommUserRank / ncclCommCount report
    # the communicator rides on the context's stream next to the kernels
    seq = native.Sequence(ctx, 3, 64, 96)
    seq.synth(1, 0)
    seq.bbme(1, 16, 8, 0, 0)
    assert np.array_equal(sequence.gather_parameters_rccl(ctx, rows[:3], 3, 0, 1), rows[:3])
    assert seq.read_mv().shape == (2, 4, 6, 2)
    seq.close()
    # the per-step exchange of a sharded block-matching run: summary rows all-gathered device to device
    # (gme_seq_mv_summary_gather), blocking and split-phase with two result slots, padded to n_max rows
    from helpers import mv_summary_rows
    seq = native.Sequence(ctx, 9, 96, 160)
    seq.synth(1234, 3)
    for proc, sw in ((0, 16), (3, 2), (2, 16)):
        seq.bbme(1, 16, sw, proc, 1)
        want = mv_summary_rows(seq.read_mv())
        assert np.array_equal(seq.mv_summary(), want), proc
        got = seq.mv_summary_gather(11, 1)
        assert got.shape == (1, 11, 6) and np.array_equal(got[0, :8], want) and not got[0, 8:].any(), proc
    shard = sequence.ShardedSequence(96, 160, 9, 1, ctx=ctx)
    assert np.array_equal(shard.unpad(got[:, :8]), want) and np.array_equal(shard.gather(want), want)
    seq.set_split_phase(True)
    seq.bbme(1, 16, 16, 0, 0)
    a = seq.mv_summary_gather(8, 1, slot=0)
    seq.wait()
    first = np.array(a)
    seq.synth(1234, 40)
    seq.bbme(1, 16, 16, 0, 0)
    b = seq.mv_summary_gather(8, 1, slot=1)                # queued while `a` is still being read
    seq.wait()
    assert np.array_equal(a, first) and np.array_equal(b[0], mv_summary_rows(seq.read_mv())) and not np.array_equal(a, b)
    seq.set_split_phase(False)
    with pytest.raises(IndexError):
        seq.mv_summary_gather(3, 1)                         # fewer rows than this rank has pairs
    seq.close()
    with pytest.raises(native.GmeError):
        sequence.comm_init(ctx, 0, 1)                       # one communicator per context
    sequence.comm_destroy(ctx)
    with pytest.raises(native.GmeError):
        sequence.comm_max(ctx, 1.0)
    ctx.close()


@pytest.mark.parametrize("pinned", [True, False])
def test_streamed_upload_equals_resident(native, pinned):
    """gme_seq_bbme_streamed: chunked upload on a copy stream overlapped with the search of the previous
    chunk (ragged chunks, frame distance 1 and 3, MAE / MSE with its per-frame table / diamond) gives
    the fields of the resident path and of the C oracle, from page-locked and from pageable frames."""
    import synth
    ctx = native.default_context()
    co = c_oracle()
    n, H, W = 30, 240, 368
    src = synth.sequence(55, 0, n, H, W)
    frames = native.pinned_empty((n, H, W)) if pinned else np.empty((n, H, W), np.uint8)
    frames[...] = src
    seq = native.Sequence(ctx, n, H, W)
    for fd, sw, proc, pn, chunk in ((1, 16, 0, 0, 7), (3, 8, 0, 1, 4), (1, 2, 3, 1, 11), (2, 16, 0, 1, 64), (1, 16, 0, 0, 1)):
        got = seq.bbme_streamed(frames, fd, 16, sw, proc, pn, chunk_frames=chunk).copy()
        assert got.shape == (n - fd, H // 16, W // 16, 2)
        for p in (0, 1, n // 2, n - fd - 1):
            assert np.array_equal(got[p], co.bbme(src[p], src[p + fd], 16, sw, proc, pn)), (fd, sw, proc, pn, chunk, p)
        seq.bbme(fd, 16, sw, proc, pn)                      # the frames stay resident
        assert np.array_equal(seq.read_mv(), got)
    # the streamed call keeps its per-frame table of box sums of squares for later calls; a later exhaustive
    # MSE search with a SMALLER frame distance reads the rows of frames [fd_small, fd_streamed) as `cur` too
    for chunk in (64, 5, 2):
        seq.bbme_streamed(frames, 3, 16, 16, 0, 1, chunk_frames=chunk)
        seq.bbme(1, 16, 16, 0, 1)
        got = seq.read_mv()
        for p in (0, 1, 2, 3, n - 2):
            assert np.array_equal(got[p], co.bbme(src[p], src[p + 1], 16, 16, 0, 1)), (chunk, p)
    part = seq.bbme_streamed(frames[:9], 1, 16, 16, 0, 0, chunk_frames=4)      # fewer frames than the sequence holds
    assert part.shape[0] == 8 and np.array_equal(part[7], co.bbme(src[7], src[8], 16, 16, 0, 0))
    with pytest.raises(IndexError):
        seq.bbme_streamed(frames[:1], 1, 16, 16, 0, 0)
    seq.close()


def test_roadmap_models_and_heuristics(native, tmp_path, capsys):
    """The authors' roadmap (recap_future_updates.md:9-14) on the device pipeline -- EXTENSIONS, self-consistency
    only: the model solves against least squares on the device's own inlier blocks, a warped scene where the
    similarity model must beat translation, the parameter heuristics and the unified CLI."""
    import gme_cli
    import motion
    import roadmap
    import synth
    from PIL import Image
    ctx = native.default_context()
    prev, cur = synth.frame(1234, 0, 480, 720), synth.frame(1234, 1, 480, 720)
    aff = roadmap.global_motion_estimation(prev, cur, "affine")
    np.testing.assert_allclose(aff, motion.global_motion_estimation(prev, cur), rtol=0, atol=0)     # the default path itself
    for model in ("translation", "similarity"):
        p = roadmap.global_motion_estimation(prev, cur, model)
        assert abs(p[0] - 5) < 0.2 and abs(p[3] + 3) < 0.2, (model, p)                             # the camera pan of the scene
    # level-2 solve against least squares on the stage the device read back
    seq = native.Sequence.from_frames(ctx, [prev, cur])
    params = roadmap.estimate_sequence(seq, 1, "similarity")[0]
    st = seq.gme_read_stage(2, 0)
    i, j = np.nonzero(~st["mask"])
    x, y = 4.0 * i, 4.0 * j
    dx, dy = st["gt"][i, j, 0].astype(float), st["gt"][i, j, 1].astype(float)
    D = np.concatenate([np.stack([np.ones_like(x), 0 * x, y, -x], 1), np.stack([0 * x, np.ones_like(x), x, y], 1)])
    th = np.linalg.lstsq(D, np.concatenate([dx, dy]), rcond=None)[0]              # (a0, b0, zoom, rotation)
    np.testing.assert_allclose(params, [th[0], -th[3], th[2], th[1], th[2], th[3]], rtol=1e-6, atol=1e-8)
    seq.close()
    # a zooming scene (x1.025 about the centre, small enough for the diamond walks to follow): the similarity model
    # reports the zoom -- column displacement 0.025 px per pixel = 0.1 per unit of the fit's y = 4 j at bs 16 -- and
    # (almost) no rotation, translation reports neither
    yy, xx = np.mgrid[0:240, 0:320].astype(np.float64)
    canvas = synth.canvas(7)

    def view(scale):
        sy = np.clip(np.rint(420 + (yy - 120) / scale), 0, 2047).astype(int)
        sx = np.clip(np.rint(560 + (xx - 160) / scale), 0, 4095).astype(int)
        return np.ascontiguousarray(canvas[sy, sx])
    a, b = view(1.0), view(1.025)
    sim = roadmap.global_motion_estimation(a, b, "similarity")
    assert 0.05 < sim[2] < 0.15 and sim[2] == sim[4] and abs(sim[1]) < 0.03 and sim[1] == -sim[5], sim
    tra = roadmap.global_motion_estimation(a, b, "translation")
    assert tra[1] == tra[2] == tra[4] == tra[5] == 0.0
    aff = roadmap.global_motion_estimation(a, b, "affine")
    assert abs(aff[2] - sim[2]) < 0.05 and abs(aff[4] - sim[4]) < 0.05, (aff, sim)
    # heuristics: the slides' block size for this frame size (docs/presentation/main.tex:426), a window that covers the pan
    s = roadmap.suggest_parameters(prev, cur)
    assert s["block_size"] == 24 and 8 <= s["search_window"] <= 16 and 0.1 <= s["outlier_fraction"] <= 0.5, s
    assert roadmap.suggest_parameters(a, a)["search_window"] <= 8          # static scene (the diamond clamp quirk moves the last block row/column by 1)
    # unified CLI on a frame directory
    d = tmp_path / "clip"
    d.mkdir()
    for k, f in enumerate((prev, cur)):
        Image.fromarray(f).save(str(d / ("%d.png" % k)))
    capsys.readouterr()
    out = gme_cli.main(["suggest", "-p", str(d), "-fi", "1"])
    assert out == s and "block_size: 24" in capsys.readouterr().out
