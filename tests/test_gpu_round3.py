"""Round 3: the streamed (host-frames) GME path, the split-phase upload, the pageable fallback of the split-phase
buffers, the CLI's --model / --suggest, the walk-search instances at bench size.  Needs an MI355X.
Same bars as test_gpu_parity.py: integer results bit-exact, parameters rtol 1e-10.
"""
import json
import os

import numpy as np
import pytest

from helpers import c_oracle, oracle_results_flow, sha

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    import _gme_native
    ctx = _gme_native.default_context()
    assert "gfx950" in ctx.info()["name"]
    return _gme_native


@pytest.mark.parametrize("bs,fd", [(16, 1), (12, 5)])
def test_estimate_stream_vs_reference(golden, native, bs, fd):
    """sequence.estimate_stream -- chunked upload of a video in host memory overlapped with the estimate of the previous
    chunks (results.py:41-59) -- on the reference's real 51-frame sequence: ragged chunks, 1 and 3 lanes, frames as a list,
    as a page-locked stack and as an ordinary NumPy stack; parameters, compensated frames and PSNR strings of ALL pairs
    against the reference's own output (g9), at the code default and at the slides' setting."""
    import motion
    import sequence
    g = golden("g9_pan240seq")
    frames = g["frames"]
    want_rec = json.loads(str(g["bs%d_fd%d_psnr_records_json" % (bs, fd)]))
    pinned = native.pinned_empty(frames.shape)
    pinned[...] = frames
    old = motion.BBME_BLOCK_SIZE
    motion.BBME_BLOCK_SIZE = bs
    try:
        for src, chunk, lanes in ((list(frames), 7, 3), (pinned, 16, 3), (np.array(frames), 50, 1), (pinned, 1000, 2), (list(frames), 1, 2)):
            P = 51 - fd
            comp = np.zeros((P, 240, 320), np.uint8)
            params, psnr = sequence.estimate_stream(src, fd, chunk_pairs=chunk, streams=lanes, compensated=comp)
            assert params.shape == (P, 6) and psnr.shape == (P,)
            for p in range(P):
                key = "bs%d_fd%d_i%d_" % (bs, fd, p + fd)
                np.testing.assert_allclose(params[p], g[key + "params"], rtol=1e-10, atol=1e-12, err_msg=key)
                assert sha(comp[p]) == str(g[key + "comp_sha"]), (key, chunk, lanes)
                assert str(complex(psnr[p], 0.0)) == want_rec[str(p + fd)], (key, chunk, lanes)
    finally:
        motion.BBME_BLOCK_SIZE = old


def test_estimate_stream_720_equals_resident_path(native):
    """The streamed path on 720x480 (the benched size): 70 synthetic frames in 4 chunks over 3 lanes equal the resident
    ShardedSequence path bit for bit (parameters included: same device sums, same host solves) and the C-oracle chain."""
    import sequence
    import synth
    frames = native.pinned_empty((70, 480, 720))
    frames[...] = synth.sequence(1234, 5, 70, 480, 720)
    p_s, psnr_s = sequence.estimate_stream(frames, 1, chunk_pairs=20, streams=3, exact_psnr=False)
    sh = sequence.ShardedSequence(480, 720, 70, 1, streams=3, interleave=True)
    sh.load(frames)
    p_r, psnr_r = sh.estimate_and_compensate()
    sh.close()
    assert np.array_equal(p_s, p_r) and np.array_equal(psnr_s, psnr_r)
    for p in (0, 19, 20, 68):
        wp, _, _, wpsnr = oracle_results_flow(frames[p], frames[p + 1])
        np.testing.assert_allclose(p_s[p], wp, rtol=1e-10, atol=1e-12)
        assert abs(psnr_s[p] - wpsnr) < 1e-9
    assert sequence.estimate_stream(frames[:1], 1)[0].shape == (0, 6)          # no pair at all


def test_split_phase_upload_and_pageable_result_buffers(native, monkeypatch):
    """(a) gme_seq_upload in split-phase mode returns with the copy queued; the stages queued behind it see the new
    frames.  (b) commit a627350: when page-locked memory runs out the split-phase result buffers fall back to ordinary
    memory -- the copies then hold the caller, the results are the same."""
    import motion
    import synth
    ctx = native.default_context()
    a, b = synth.sequence(7, 0, 6, 96, 160), synth.sequence(8, 3, 6, 96, 160)
    seq = native.Sequence(ctx, 6, 96, 160)
    want = {}
    for name, f in (("a", a), ("b", b)):
        seq.upload(0, f)
        want[name] = motion.estimate_sequence(seq, 1)
    for pageable in (False, True):
        if pageable:
            def no_pinned(shape, dtype=np.uint8):
                raise MemoryError("no page-locked memory left (test)")
            monkeypatch.setattr(native, "pinned_empty", no_pinned)
            seq.__dict__.pop("_pin", None)
        seq.set_split_phase(True)
        try:
            for name, f in (("a", a), ("b", b), ("a", a)):
                host = np.array(f)
                seq.upload(0, host)                                   # queued
                p0 = seq.gme_begin(1, 16)
                seq.wait()
                params = np.array(p0)
                for level in (1, 2):
                    params[:, 0] *= 2
                    params[:, 3] *= 2
                    sums = seq.gme_fit(level, params.astype(np.float64), 0.3)
                    seq.wait()
                    params = motion._solve_batch(sums)
                assert np.array_equal(params, want[name]), (pageable, name)
                if pageable:
                    assert seq._pin["p0"].base is None                           # an ordinary array, not a view of a page-locked block
        finally:
            seq.set_split_phase(False)
    seq.close()


def test_begin_fit_equals_staged_calls(native):
    """gme_seq_gme_begin_fit (first parameters -> projection -> level-1 fit on the device) hands back exactly what
    gme_seq_gme_begin + the host's float32 projection + gme_seq_gme_fit(1) do: first parameters and level-1 sums bit for
    bit, and the run continues into level 2 the same way -- blocking and split-phase, diamond and exhaustive level searches."""
    import motion
    import synth
    ctx = native.default_context()
    seq = native.Sequence(ctx, 9, 240, 368)
    seq.upload(0, synth.sequence(77, 2, 9, 240, 368))
    for proc, sw in ((3, 2), (0, 8)):
        p0 = np.array(seq.gme_begin(1, 16, proc, sw))
        proj = p0.copy()
        proj[:, 0] *= 2
        proj[:, 3] *= 2
        sums1 = np.array(seq.gme_fit(1, proj.astype(np.float64), 0.3))
        for split in (False, True):
            seq.invalidate_pyramids()
            seq.set_split_phase(split)
            q0, t1 = seq.gme_begin_fit(1, 16, 0.3, proc, sw)
            if split:
                seq.wait()
            assert np.array_equal(q0, p0) and np.array_equal(np.array(t1).view(np.uint64), sums1.view(np.uint64)), (proc, split)
            p = motion._solve_batch(t1)
            p[:, 0] *= 2
            p[:, 3] *= 2
            t2 = seq.gme_fit(2, p, 0.3)
            if split:
                seq.wait()
            t2 = np.array(t2)
            seq.set_split_phase(False)
        want = motion.estimate_sequence(seq, 1, proc, sw)
        assert np.array_equal(motion._solve_batch(t2), want), proc
    seq.close()


def test_set_frames_limits_the_stages(native):
    """gme_seq_set_frames: a sequence created for 12 frames acts as one of 5, then 9, then 12 -- block matching, the staged
    estimate and the compensation cover exactly the pairs of the frames in use and equal a fresh sequence of that length."""
    import motion
    import synth
    ctx = native.default_context()
    frames = synth.sequence(21, 0, 12, 96, 160)
    big = native.Sequence(ctx, 12, 96, 160)
    big.upload(0, frames)
    for n in (5, 9, 12, 2):
        big.set_frames(n)
        big.upload(0, frames[:n])
        ref = native.Sequence.from_frames(ctx, frames[:n])
        for seq in (big, ref):
            seq.bbme(1, 16, 16, 0, 1)
        assert big.read_mv().shape[0] == n - 1 and np.array_equal(big.read_mv(), ref.read_mv()), n
        p_big, p_ref = motion.estimate_sequence(big, 1), motion.estimate_sequence(ref, 1)
        assert p_big.shape == (n - 1, 6) and np.array_equal(p_big, p_ref), n
        assert np.array_equal(big.compensate(1, 16, p_big), ref.compensate(1, 16, p_ref)), n
        assert np.array_equal(big.read_compensated(n - 2), ref.read_compensated(n - 2)), n
        ref.close()
    for bad in (0, 13):
        with pytest.raises(IndexError):
            big.set_frames(bad)
    big.close()


def test_cli_results_model_and_suggest(golden, native, tmp_path, monkeypatch, capsys):
    """gme_cli results --model / --suggest (recap_future_updates.md:9-14, extensions): the affine default writes the
    reference's psnr_records (g7 flow); a non-affine model and the suggested constants run end to end and report what they chose."""
    import gme_cli
    import synth
    from PIL import Image
    monkeypatch.chdir(tmp_path)
    d = tmp_path / "resources" / "videos" / "clip"
    d.mkdir(parents=True)
    frames = synth.sequence(1234, 0, 6, 240, 320)
    for k, f in enumerate(frames):
        Image.fromarray(f).save(str(d / ("%d.png" % k)))
    base = gme_cli.main(["results", "-v", "clip", "-f", "1"])
    assert len(base) == 5 and base == gme_cli.main(["results", "-v", "clip", "-f", "1", "--model", "affine"])
    sim = gme_cli.main(["results", "-v", "clip", "-f", "1", "--model", "similarity"])
    tra = gme_cli.main(["results", "-v", "clip", "-f", "1", "--model", "translation"])
    assert set(sim) == set(base) == set(tra)
    for rec in (sim, tra):                                            # a pure pan: every model compensates it about as well
        for k in base:
            assert abs(complex(rec[k]).real - complex(base[k]).real) < 1.0, (k, rec[k], base[k])
    capsys.readouterr()
    import motion
    before = motion.BBME_BLOCK_SIZE, motion.MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE
    sug = gme_cli.main(["results", "-v", "clip", "-f", "1", "--suggest"])
    out = capsys.readouterr().out
    assert "suggested for this video: {'block_size': 12" in out and len(sug) == 5       # 240 / 20
    assert (motion.BBME_BLOCK_SIZE, motion.MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE) == before
    with pytest.raises(SystemExit):
        gme_cli.main(["results", "-v", "clip", "--model", "perspective"])


@pytest.mark.parametrize("proc,pnorm", [(1, 0), (1, 1), (2, 0), (2, 1)])
def test_walk_instances_at_bench_size(native, proc, pnorm):
    """bench.py's tss720 / tdl720: the three-step and 2-D log instances (k_walk16s<PNORM, PROC, FITS>, 0 SGPR spills) on 41
    pairs of 720x480 -- asserted through the launch plan -- every pair against the C oracle; sw 16 and sw 4 (every round
    fits the LDS window: the FITS instance, first window fetched ahead) and sw 24 (the instance with the global-memory path)."""
    ctx = native.default_context()
    co = c_oracle()
    seq = native.Sequence(ctx, 42, 480, 720)
    seq.synth(1234, 0)
    frames = [seq.read_frame(i) for i in range(42)]
    for sw, fits in ((16, "true"), (4, "true"), (24, "false")):     # FITS: every round fits the 48 x 64 LDS window (first step / sw <= 16)
        seq.bbme(1, 16, sw, proc, pnorm)
        assert ctx.last_bbme_info()["plan"].startswith("k_walk16s<%d,%d,%s>" % (pnorm, proc, fits)), ctx.last_bbme_info()
        mv = seq.read_mv()
        for p in range(41):
            assert np.array_equal(mv[p], co.bbme(frames[p], frames[p + 1], 16, sw, proc, pnorm)), (sw, p)
    seq.close()
