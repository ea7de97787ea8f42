"""CPU-side checks: the C-ABI library loads and exports what include/gme_hip.h declares,
host logic (sharding, gather over gloo with 2 ranks, table indices, 3x3 solve), and the
product path's refusal to run without its native library or a GPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import _gme_native
    header = open(os.path.join(REPO, "include", "gme_hip.h")).read()
    declared = set(re.findall(r"GME_API [^;(]*?\b(gme_\w+)\s*\(", header))
    assert len(declared) >= 25
    lib = _gme_native.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    # the ctypes signature table covers exactly the header
    assert declared == set(_gme_native.exported_symbols())


def test_no_device_means_loud_failure():
    """No GPU in the build container: the product path must raise, never fall back."""
    import _gme_native
    lib = _gme_native.load_library()
    if lib.gme_device_count() > 0:
        pytest.skip("a HIP device is present")
    import bbme
    f = np.zeros((32, 32), np.uint8)
    with pytest.raises(_gme_native.GmeError):
        bbme.get_motion_field(f, f)


def test_missing_library_is_an_error(monkeypatch):
    import _gme_native
    monkeypatch.setattr(_gme_native, "_lib", None)
    monkeypatch.setattr(_gme_native, "LIB_PATH", "/nonexistent/libgme_hip.so")
    with pytest.raises(_gme_native.GmeError, match="no CPU fallback"):
        _gme_native.load_library()


def test_loading_the_library_selects_dmabuf_ipc_unless_the_launcher_chose():
    """One rank per GPU over RCCL needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this pool; the HIP runtime reads it at its first
    call, which is after load_library().  Checked in fresh interpreters: unset -> "0", exported -> left alone."""
    code = ("import os, sys; sys.path.insert(0, %r); import _gme_native; _gme_native.load_library(); "
            "print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])" % os.path.join(REPO, "global-motion-estimation_amd"))
    for exported, want in ((None, "0"), ("1", "1")):
        env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
        if exported is not None:
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = exported
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == want


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "global-motion-estimation_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "gme_oracle" not in text.replace("oracle/gme_oracle.py affine_field", ""), f
                assert "libgme_oracle" not in text, f


def test_table_indices_follow_python_lists():
    import bbme
    assert bbme._table_index(-1, 4, "x") == 3 and bbme._table_index(2, 4, "x") == 2
    with pytest.raises(IndexError):
        bbme._table_index(4, 4, "x")
    with pytest.raises(IndexError):
        bbme._table_index(-5, 4, "x")
    a = np.arange(16, dtype=np.uint8).reshape(4, 4)
    b = a[::-1].copy()
    assert bbme.compute_dfd(a, b, 0) == np.float32(np.abs(a.astype(int) - b).sum())
    assert bbme.compute_dfd(a, b, 1) == np.float32(((a.astype(int) - b) ** 2).sum())
    with pytest.raises(AssertionError):
        bbme.compute_dfd(a, b[:2])
    assert len(bbme.searching_procedures) == 4 and len(bbme.pnorm_distances) == 2


def test_solve_matches_oracle_and_raises_on_singular(golden):
    import motion
    from oracle import gme_oracle
    g = golden("g4_gme")
    sums = np.concatenate([g["race_l2_F"].reshape(9), g["race_l2_Sx"], g["race_l2_Sy"]])
    got = motion._solve(sums)
    assert np.array_equal(got, gme_oracle.solve_parameters(g["race_l2_F"], g["race_l2_Sx"], g["race_l2_Sy"]))
    np.testing.assert_allclose(got, g["race_params"], rtol=1e-10, atol=1e-12)
    with pytest.raises(np.linalg.LinAlgError):
        motion._solve(np.zeros(15))
    p = np.array([1.5, 0, 0, -2.25, 0, 0], np.float32)
    assert motion.parameter_projection(p) is p and p[0] == 3.0 and p[3] == -4.5
    d = motion.affine_model(3, 4, np.array([1.0, .5, .25, -1, 0, 2]))
    assert d.shape == (2,) and d[0] == 1 + 1.5 + 1 and d[1] == 7


def test_batched_solve_equals_per_pair_solve(golden):
    """estimate_sequence solves all pairs with one stacked inv/matmul; it must equal the
    per-pair calls the reference makes (motion.py:262-264,280-282) bit for bit."""
    import motion
    g = golden("g4_gme")
    rng = np.random.default_rng(3)
    rows = []
    for tag in ("synth720", "race", "pan240", "dp", "small", "bs12"):
        for lvl in (1, 2):
            rows.append(np.concatenate([g["%s_l%d_F" % (tag, lvl)].reshape(9), g["%s_l%d_Sx" % (tag, lvl)],
                                        g["%s_l%d_Sy" % (tag, lvl)]]))
    for _ in range(500):
        a = rng.normal(size=(3, 3))
        rows.append(np.concatenate([(a @ a.T * rng.uniform(1e-3, 1e3)).reshape(9), rng.normal(size=6)]))
    rows = np.array(rows)
    batch = motion._solve_batch(rows)
    for k, r in enumerate(rows):
        assert np.array_equal(batch[k], motion._solve(r)), k
    assert motion._solve_batch(np.zeros((0, 15))).shape == (0, 6)
    with pytest.raises(np.linalg.LinAlgError):
        motion._solve_batch(np.zeros((2, 15)))


def test_shard_ranges_cover_pairs_once():
    import sequence
    for n in (0, 1, 5, 8, 1999):
        for world in (1, 2, 3, 8):
            spans = [sequence.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    assert sequence.shard_frames(1999, 1, 7, 8) == (1749, 251)
    assert sequence.shard_frames(3, 2, 1, 8) == (0, 0)
    with pytest.raises(ValueError):
        sequence.shard_range(4, 2, 2)


_GATHER = r'''
import os, sys
sys.path[:0] = [%(pkg)r, %(tests)r]
import numpy as np, torch.distributed as dist
import sequence
from helpers import gather_rows_torch, mv_summary_rows
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 7
a, b = sequence.shard_range(n, rank, world)
local = np.arange(n * 6, dtype=np.float64).reshape(n, 6)[a:b] * 1.5
full = gather_rows_torch(local, n, rank, world)
assert full.shape == (n, 6) and np.array_equal(full, np.arange(n * 6, dtype=np.float64).reshape(n, 6) * 1.5)
empty = gather_rows_torch(np.zeros((1 if rank == 0 else 0, 6)), 1, rank, world)
assert empty.shape == (1, 6)
# the row exchange of a sharded block-matching run (bench.py at N > 1): every rank summarises the fields of ITS
# pairs into 48-byte rows, the all-gather hands every rank all of them, in pair order
rng = np.random.default_rng(5)
fields = rng.integers(-16, 32, (n, 30, 45, 2)).astype(np.int32)             # the same on every rank (same seed)
rows = gather_rows_torch(mv_summary_rows(fields[a:b]), n, rank, world)
assert rows.shape == (n, 6) and np.array_equal(rows, mv_summary_rows(fields))
# the fixed-size blocks gme_seq_mv_summary_gather returns ([world][n_max][6], zero-padded) unpad to the same rows
shard = sequence.ShardedSequence.__new__(sequence.ShardedSequence)
shard.n_pairs_total, shard.world = n, world
longest, sizes = sequence.pad_and_trim(n, world)
blocks = np.zeros((world, longest, 6))
for r, (lo, hi) in enumerate(sizes):
    blocks[r, :hi - lo] = rows[lo:hi]
assert np.array_equal(shard.unpad(blocks), rows)
dist.barrier()
dist.destroy_process_group()
sys.stdout.write("rank" + str(rank) + " ok\n")
'''


def test_gather_over_gloo_world_2(tmp_path):
    script = tmp_path / "gather.py"
    script.write_text(_GATHER % {"pkg": os.path.join(REPO, "global-motion-estimation_amd"), "tests": os.path.join(REPO, "tests")})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank0 ok" in out.stdout and "rank1 ok" in out.stdout


_BENCH_COMM = r"""
import os, sys
sys.path[:0] = [%(repo)r, %(pkg)r, %(tests)r]
import numpy as np
import bench, sequence
class FakeCtx:                       # bench.Comm only syncs the context before a barrier
    def sync(self): pass
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = sys.argv[1]
if mode == "fallback":               # the C-ABI communicator is refused on every rank: the agreed fallback must be LABELLED
    def refuse(ctx, r, w, timeout_s=180.0):
        raise sequence.CommUnavailable("probe failed on 1 of 2 ranks (rank 1: OSError('no librccl'))")
    sequence.comm_init = refuse
    real = bench.Comm._torch_init
    bench.Comm._torch_init = lambda self, backend, r, w, l: real(self, "gloo" if backend == "nccl" else backend, r, w, l)
    os.environ.pop("GME_BENCH_BACKEND", None)
comm = bench.Comm(FakeCtx(), rank, world, 0)
rep = bench.rank_report(comm, 1000.0 * (rank + 1), 0.25 * (rank + 1), 5)
rows = comm.gather_rows(np.full((3 + rank, 7), float(rank)), 7)
assert rows.shape == (7, 7) and list(rows[:, 0]) == [0.0] * 3 + [1.0] * 4, rows
sys.stdout.write("rank%%d kind=%%s degraded=%%s per_rank=%%s gather_ms=%%s;\n" %% (rank, comm.kind, comm.degraded, rep["per_rank_pairs_per_s"], rep["gather_ms_per_step"]))
sys.stdout.flush()
comm.close()
"""


@pytest.mark.parametrize("mode", ["gloo", "fallback"])
def test_bench_line_fields_of_a_two_rank_launch(tmp_path, mode):
    """an N > 1 line describes itself -- the slowest and the fastest rank's pairs/s, the exchange's ms per
    step, and `degraded` when the rows did not travel over the C ABI's RCCL communicator (bench.py then exits 3 after
    printing).  Two gloo ranks on the CPU: bench.Comm + bench.rank_report, the gather of ragged shards."""
    script = tmp_path / "bench_comm.py"
    script.write_text(_BENCH_COMM % {"repo": REPO, "pkg": os.path.join(REPO, "global-motion-estimation_amd"), "tests": os.path.join(REPO, "tests")})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", GME_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29641", str(script), mode],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    for r in (0, 1):
        import re
        line = re.findall(r"rank%d kind=.*?;" % r, out.stdout, re.S)          # two processes share the pipe: lines may abut
        assert line, out.stdout + out.stderr
        assert "kind=torch.distributed/gloo" in line[0]
        assert "per_rank={'min': 1000.0, 'max': 2000.0}" in line[0], line[0]
        assert "gather_ms=%s;" % (50.0 * (r + 1)) in line[0], line[0]
        if mode == "gloo":
            assert "degraded=None" in line[0], line[0]
        else:
            assert "degraded=torch.distributed fallback (C-ABI RCCL communicator unavailable: probe failed on 1 of 2 ranks" in line[0], line[0]
    src = open(os.path.join(REPO, "bench.py")).read()
    assert 'out["degraded"] = comm.degraded' in src and "sys.exit(3)" in src


def test_synth_generator_self_checks():
    """SURVEY.md §8(d) constants."""
    import synth
    from helpers import sha
    assert int(synth.hash64(1234, 0)) == 0xdf34b78a642501be and int(synth.hash64(1234, 1)) == 0x9bbdb63ded02052d
    assert list(synth.canvas(1234)[0, :8]) == [140, 141, 145, 142, 143, 135, 134, 141]
    assert sha(synth.frame(1234, 0, 480, 720)) == "9736c2ac7184b594c41cb75edac231feac5775691bca78fc0af2cb8674ae7308"
    assert sha(synth.frame(1234, 1, 480, 720)) == "9652a5b6f736753131bdcc1961978ceb4238d311bb56ddba6e66d15ddffa7217"


def test_rescale_motion_field_and_some_data(golden, tmp_path, capsys):
    """bbme.rescale_motion_field (bbme.py:537-546) and utils.some_data (utils.py:138-164)."""
    import bbme
    import utils
    g = golden("g8_next")
    for tag in ("int", "float"):
        got = bbme.rescale_motion_field(g["resc_in_" + tag])
        assert got.dtype == np.int32 and np.array_equal(got, g["resc_out_" + tag]), tag
    assert np.array_equal(bbme.rescale_motion_field(g["resc_in_int"], scale=3), g["resc_out_int3"])
    path = tmp_path / "psnr_records.json"
    path.write_text(str(g["psnr_records_json"]))
    utils.some_data(str(path))
    assert capsys.readouterr().out == str(g["some_data_stdout"])


def test_frame_loaders_without_cv2(tmp_path):
    """utils.get_video_frames on an image directory, a .npy stack and a .y4m file."""
    import synth
    import utils
    from PIL import Image
    frames = synth.sequence(3, 0, 4, 48, 64)
    d = tmp_path / "clip"
    d.mkdir()
    for i, f in enumerate(frames):
        Image.fromarray(f).save(d / ("f%d.png" % (i + 9)))          # 9, 10, 11, 12: numeric, not lexical, order
    got = utils.get_video_frames(str(d))
    assert len(got) == 4 and all(np.array_equal(a, b) for a, b in zip(got, frames))
    np.save(tmp_path / "clip.npy", frames)
    assert np.array_equal(np.array(utils.get_video_frames(str(tmp_path / "clip.npy"))), frames)
    with open(tmp_path / "clip.y4m", "wb") as f:
        f.write(b"YUV4MPEG2 W64 H48 F30:1 Ip A1:1 C420jpeg\n")
        for fr in frames:
            f.write(b"FRAME\n" + fr.tobytes() + bytes(64 * 48 // 2))
    got = utils.get_video_frames(str(tmp_path / "clip.y4m"))
    assert len(got) == 4 and all(np.array_equal(a, b) for a, b in zip(got, frames))
    # needle diagram: BGR image of the frame's size; vectors drawn in red
    mf = np.zeros((3, 4, 2), np.int32)
    mf[1, 2] = (5, -3)
    img = utils.draw_motion_field(frames[0], mf)
    assert img.shape == (48, 64, 3) and img.dtype == np.uint8
    assert (img[:, :, 2] == 255).any() and utils.write_image(str(tmp_path / "o.png"), img)


_RDV = r'''
import os, sys
sys.path[:0] = [%(pkg)r]
import sequence
rank, world = int(sys.argv[1]), int(sys.argv[2])
blob = sequence.comm_exchange_id(lambda: bytes(range(128)), rank, world, timeout_s=60)
assert blob == bytes(range(128)), rank
sys.stdout.write("rank%%d got the id\n" %% rank)
'''


def test_rccl_id_rendezvous_and_padding(tmp_path):
    """The launcher side of gme_comm_init: rank 0 publishes the 128-byte RCCL id through a file keyed by
    MASTER_PORT and the ranks' common parent pid, the other ranks poll for it (started BEFORE rank 0
    here); and the fixed block size / trimming of the padded all-gather (gme_shard_gather)."""
    import sequence
    script = tmp_path / "rdv.py"
    script.write_text(_RDV % {"pkg": os.path.join(REPO, "global-motion-estimation_amd")})
    env = dict(os.environ, MASTER_PORT="29777")
    env.pop("GME_COMM_ID_FILE", None)
    late = [subprocess.Popen([sys.executable, str(script), str(r), "3"], env=env, stdout=subprocess.PIPE, text=True) for r in (1, 2)]
    import time
    time.sleep(0.5)
    first = subprocess.run([sys.executable, str(script), "0", "3"], env=env, capture_output=True, text=True, timeout=120)
    assert first.returncode == 0 and "rank0 got the id" in first.stdout, first.stdout + first.stderr
    for r, p in zip((1, 2), late):
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0 and ("rank%d got the id" % r) in out
    path = "/tmp/gme_rccl_%d/29777_%d.a0_0.id.0" % (os.getuid(), os.getpid())  # a private directory, keyed by port, parent, attempt
    assert os.path.exists(path) and (os.stat(path).st_mode & 0o777) == 0o600
    assert (os.stat(os.path.dirname(path)).st_mode & 0o777) == 0o700
    os.unlink(path)
    # rank 0 cannot make an id: it says so through the same file and the others give up at once (no 3-minute wait)
    os.environ["GME_COMM_ID_FILE"] = str(tmp_path / "failed.id")
    try:
        def broken():
            raise OSError("no librccl")
        with pytest.raises(OSError):
            sequence.comm_exchange_id(broken, 0, 2)
        t0 = time.time()
        sequence.Rendezvous._attempts.clear()              # "rank 1" is another process: its first attempt with this key
        with pytest.raises(RuntimeError):
            sequence.comm_exchange_id(broken, 1, 2, timeout_s=60)
        assert time.time() - t0 < 5
    finally:
        del os.environ["GME_COMM_ID_FILE"]
    assert sequence.pad_and_trim(7, 2) == (4, [(0, 3), (3, 7)])
    assert sequence.pad_and_trim(1999, 8)[0] == 250
    assert sequence.pad_and_trim(0, 4) == (1, [(0, 0)] * 4)
    # trimming the padded blocks back (what gather_parameters_rccl does with gme_shard_gather's output)
    longest, sizes = sequence.pad_and_trim(7, 2)
    full = np.arange(42, dtype=np.float64).reshape(7, 6)
    out = np.zeros((2, longest, 6))
    for r, (a, b) in enumerate(sizes):
        out[r, :b - a] = full[a:b]
    assert np.array_equal(np.concatenate([out[r, :b - a] for r, (a, b) in enumerate(sizes)]), full)


_AGREE = r"""
import os, sys
sys.path[:0] = [%(pkg)r]
import sequence
rank, world, scenario = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
state = {"comm": False, "entered_init": False}
def probe():
    if scenario == "probe_fails_on_2" and rank == 2:
        raise OSError("cannot open librccl.so")
def make_id():
    return bytes(range(128))
def init(blob):
    assert blob == bytes(range(128))
    state["entered_init"] = True
    if scenario == "init_fails_on_1" and rank == 1:
        raise RuntimeError("ncclCommInitRank failed")
    state["comm"] = True
def destroy():
    state["comm"] = False
rdv = sequence.Rendezvous(rank, world, timeout_s=60)
try:
    sequence.collective_init(rdv, probe, make_id, init, destroy)
    out = "up"
except sequence.CommUnavailable as e:
    out = "unavailable[%%s]" %% e
sys.stdout.write("rank%%d %%s comm=%%s entered_init=%%s\n" %% (rank, out, state["comm"], state["entered_init"]))
"""


_RETRY = r"""
import os, sys
sys.path[:0] = [%(pkg)r]
import sequence
rank, world = int(sys.argv[1]), int(sys.argv[2])
log = []
for attempt in (0, 1, 2):
    def probe():
        if attempt == 0 and rank == 1:
            raise OSError("cannot open librccl.so")
    def make_id():
        return bytes([attempt]) * 128
    def init(blob):
        assert blob == bytes([attempt]) * 128, "rank %%d read another attempt's id" %% rank
    def tag():
        return b"0000:05:00.0" if attempt == 1 else b"0000:%%02x:00.0" %% rank      # attempt 1: both ranks on one device
    rdv = sequence.Rendezvous(rank, world, timeout_s=60)
    try:
        sequence.collective_init(rdv, probe, make_id, init, lambda: None, tag)
        log.append("up")
    except sequence.CommUnavailable as e:
        log.append("unavailable[%%s]" %% e)
    left = sorted(f for f in os.listdir(os.path.dirname(rdv.base)) if f.startswith(os.path.basename(rdv.base) + ".") and ".ack_" not in f
                  and f.endswith(".%%d" %% rank))
    log.append("left=%%d" %% len(left))
sys.stdout.write("rank%%d %%s\n" %% (rank, " | ".join(log)))
"""


def test_comm_init_can_be_retried_with_the_same_key(tmp_path):
    """a failed bring-up used to leave its probe.* / init.* files behind, keyed only by (port, parent pid): a
    second comm_init in the same processes could read the first attempt's 'no:' marker or its ncclUniqueId.  Keys now
    carry an attempt nonce and a failed attempt removes its files (after an ack round).  Three attempts with ONE base:
    a probe failure, two ranks on one device (refused in phase 1, nobody enters init), then a good one whose id must be
    its own."""
    script = tmp_path / "retry.py"
    script.write_text(_RETRY % {"pkg": os.path.join(REPO, "global-motion-estimation_amd")})
    env = dict(os.environ, GME_COMM_ID_FILE=str(tmp_path / "launch"))
    env.pop("GME_COMM_ALLOW_SHARED_DEVICE", None)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in (1, 0)]
    for r, p in zip((1, 0), procs):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err
        parts = out.strip().split(" | ")
        assert parts[0].startswith("rank%d unavailable[probe failed on 1 of 2 ranks (rank 1: OSError" % r), out
        assert parts[1] == "left=0", out                                  # the failed attempt removed its files
        assert "ranks share a device (0000:05:00.0 on ranks [0, 1])" in parts[2] and parts[3] == "left=0", out
        assert parts[4] == "up", out
    assert not [f for f in os.listdir(tmp_path) if f.startswith("launch.") and ".ack_" not in f and ".a0_2." not in f], os.listdir(tmp_path)


@pytest.mark.parametrize("scenario", ["all_fine", "probe_fails_on_2", "init_fails_on_1"])
def test_comm_init_is_decided_by_all_ranks(tmp_path, scenario):
    """VERDICT r2 / ADVICE: the transport is never chosen per rank.  sequence.collective_init wraps ncclCommInitRank in
    two agreements over the file rendezvous: if any rank cannot load RCCL nobody enters the collective; if any rank's
    init fails, the others give their communicator back; either way EVERY rank raises CommUnavailable (and bench.py then
    takes the same fallback on every rank) or every rank is up.  A stale file of an earlier launch is never read."""
    script = tmp_path / "agree.py"
    script.write_text(_AGREE % {"pkg": os.path.join(REPO, "global-motion-estimation_amd")})
    base = str(tmp_path / "launch")
    stale = base + ".a0_0.probe.0"                         # an earlier launch with the same key died after saying "no"
    with open(stale, "wb") as f:
        f.write(b"no:stale failure marker")
    os.utime(stale, (1.0e9, 1.0e9))
    env = dict(os.environ, GME_COMM_ID_FILE=base)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "3", scenario], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in (2, 1, 0)]
    outs = {}
    for r, p in zip((2, 1, 0), procs):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err
        outs[r] = out.strip()
    if scenario == "all_fine":
        assert all(outs[r] == "rank%d up comm=True entered_init=True" % r for r in range(3)), outs
    elif scenario == "probe_fails_on_2":
        assert all("unavailable[probe failed on 1 of 3 ranks (rank 2: OSError" in outs[r] for r in range(3)), outs
        assert all("entered_init=False" in outs[r] for r in range(3)), outs      # nobody went into the collective
    else:
        assert all("unavailable[init failed on 1 of 3 ranks (rank 1: RuntimeError" in outs[r] for r in range(3)), outs
        assert all("comm=False" in outs[r] for r in range(3)), outs              # ranks 0 and 2 gave theirs back


def test_committed_profiles_belong_to_these_kernels():
    """bench.py prints `roofline.traffic` and the `issue` object only from a committed PMC summary taken with the SAME
    kernel sources (kernel_source_sha).  A change under csrc/ without re-running tools/final_run.sh ... prof would turn
    them into `null` in the driver's bench line: this test says so first.  One summary per BASELINE config and the walk
    searches, each naming the kernel the launch plan names."""
    sys.path.insert(0, REPO)
    import bench
    sha = bench.kernel_source_sha()
    assert set(bench.CONFIG_SOURCES) <= set(bench.CONFIGS) and all(
        os.path.exists(os.path.join(REPO, "global-motion-estimation_amd", "csrc", f)) for fs in list(bench.CONFIG_SOURCES.values()) + [bench.COMMON_SOURCES] for f in fs)
    plans = {"exh720": "k_exh_sea16p<3,5>", "exh720mse": "k_exh_mfma16<3,3,4>", "exh720mse_vec": "k_exh_sea16p_mse<3,5>", "exh1080": "k_exh_sea16p<5,7>",
             "exh1080mse_mfma": "k_exh_mfma16<5,2,4>",
             "exh1080mse": "k_exh_sea16p_mse<5,7>", "gme720": "k_walk16<1>", "gme1080exh": "k_exh_sea16p_mse<5,7>",
             "tss720": "k_walk16s<1,1,true>", "tdl720": "k_walk16s<1,2,true>", "dia720mse": "k_walk16<1>", "dia720": "k_walk16<0>",
             "gme1080": "k_walk16<1>", "seq1080": "k_walk16<1>", "gme720dev": "k_walk16<1>", "tss_bs4sw2": "k_walkq<4,1>",
             "gme_pan240_bs12fd5": "k_walkq<12,1>"}
    for config, kernel in plans.items():
        vals, psha, name = bench.committed_profile(config, kernel)
        # the whole of csrc/ as recorded, or -- after a change that cannot reach this config's kernels -- the files they are
        # compiled from (bench.CONFIG_SOURCES; `# config_source_sha:` in the summary)
        assert name and bench.profile_is_current(config, vals, psha), \
            "profiles/%s was taken with other kernel sources (%s != %s, config files %s != %s): re-run tools/final_run.sh <tag> prof" % (
                name, psha, sha, vals.get("_config_sha"), bench.kernel_source_sha(config))
        assert vals["_kernel"].split("<")[0] == kernel.split("<")[0], (config, vals.get("_kernel"))
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU"):
            assert vals.get(counter, 0) > 0, (config, counter)
        stats = os.path.join(REPO, "profiles", name.replace("_pmc_summary.txt", "_kernel_stats.csv"))
        assert os.path.exists(stats) and kernel.split("<")[0] in open(stats).read(), stats


def test_benched_kernel_instances_do_not_spill():
    """DESIGN.md said "the R = 5 body no longer spills to scratch" while the committed sources compiled to
    215 spilled VGPRs and 192 bytes of scratch per lane.  The Makefile now keeps the compiler's own resource remarks
    (build/*.remarks); every kernel instance bench.py's configs launch must show no VGPR spill and no scratch, and the
    table DESIGN.md quotes must be the build's (tools/resource_table.py --check)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import resource_table
    rows = {r["name"]: r for r in resource_table.kernels()}
    assert rows, "no build/*.remarks: make -C global-motion-estimation_amd/csrc"
    for name in resource_table.BENCHED:
        hits = [r for k, r in rows.items() if k == name or (name in ("k_fit_level", "k_compensate16", "k_pyrdown_lds", "k_sqbox16") and k.startswith(name))]
        assert hits, "no resource remarks for %s" % name
        for r in hits:
            assert r["vgpr_spill"] == 0 and r["scratch"] == 0, (r["name"], r)
    # what the launch bounds promise: 8 waves per SIMD for the R = 3 elimination kernels and the walk searches, >= 5 for the
    # 720x480 hostile-content MSE body (round 3 claimed 5 and had 4)
    m3 = rows["k_exh_mfma16<3, 3, 4>"]              # its speed hangs on waves per SIMD: VGPRs + AGPRs (the accumulators) of the unified file
    assert m3["vgpr"] + m3["agpr"] <= 96 and m3["waves"] >= 5, m3
    for name, waves in (("k_exh_sea16p<3, 5, 36>", 8), ("k_exh_sea16p_mse<3, 5, 36>", 8), ("k_walk16<1>", 8), ("k_exh_redo16<3, true>", 5),
                        ("k_exh_sea16p<5, 7, 38>", 6), ("k_exh_sea16p_mse<5, 7, 38>", 6), ("k_exh_redo16<5, true>", 4)):
        assert rows[name]["waves"] >= waves, (name, rows[name])
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "resource_table.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_pmc_summary_reads_only_the_newest_pass(tmp_path):
    """tools/pmc_summary.py: gpurun merges a call's files INTO the local gpurun_out/, so a pass directory re-used by a later
    profile run also holds the CSVs of earlier builds.  Only the newest CSV of each pass directory may enter the summary
    (round 3: three runs had been averaged into one file under one kernel_source_sha)."""
    import subprocess
    d = tmp_path / "cfg" / "pmc_FETCH_SIZE" / "box"
    d.mkdir(parents=True)
    head = "Kernel_Name,Counter_Name,Counter_Value,Grid_Size,Workgroup_Size,LDS_Block_Size,VGPR_Count,SGPR_Count\n"
    (d / "100_counter_collection.csv").write_text(head + "k_old(int),FETCH_SIZE,111,64,64,0,8,16\n")
    os.utime(str(d / "100_counter_collection.csv"), (1000, 1000))
    (d / "090_counter_collection.csv").write_text(head + "k_new(int),FETCH_SIZE,222,64,64,0,8,16\n" + "k_new(int),FETCH_SIZE,224,64,64,0,8,16\n")
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"), str(tmp_path / "cfg"), "k_"],
                         capture_output=True, text=True, check=True).stdout
    assert "k_new" in out and "mean=223" in out and "k_old" not in out, out
    assert "# kernel_source_sha: " in out


def test_stream_schedule_covers_every_pair_once():
    """StreamEstimator.schedule: chunks no larger than a lane holds, shrinking towards the end (the last chunk's estimate is
    the stretch no upload hides), never below min_chunk except for the remainder, every pair exactly once."""
    import sequence
    est = sequence.StreamEstimator.__new__(sequence.StreamEstimator)
    for cap, mn, lanes, P in ((512, 64, 2, 2048), (512, 64, 3, 2048), (128, 128, 2, 1000), (7, 7, 3, 50), (1000, 64, 2, 50), (16, 4, 1, 1), (64, 64, 2, 0)):
        est.chunk_pairs, est.min_chunk, est.lanes = cap, min(mn, cap), [None] * lanes
        ch = est.schedule(P)
        assert (not ch and P == 0) or (ch[0][0] == 0 and ch[-1][1] == P), (cap, mn, lanes, P, ch)
        assert all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
        sizes = [b - a for a, b in ch]
        assert all(0 < n <= cap for n in sizes) and sizes == sorted(sizes[:-1], reverse=True) + sizes[-1:], sizes
        assert all(n >= min(mn, cap) for n in sizes[:-1]), sizes
    est.chunk_pairs, est.min_chunk, est.lanes = 512, 64, [None, None]
    assert [b - a for a, b in est.schedule(2048)] == [512, 384, 288, 216, 162, 122, 91, 69, 64, 64, 64, 12]


def test_bench_helpers():
    """bench.py's sampling (first and last pair always in), content stacks and profile identity."""
    sys.path.insert(0, REPO)
    import bench
    s = bench.sample_pairs(2048, 64)
    assert s[0] == 0 and s[-1] == 2047 and len(s) == 64 and s == sorted(set(s))
    assert bench.sample_pairs(3, 64) == [0, 1, 2]
    f, h, w = bench.host_content("pan240seq", 120, 480, 720)
    assert f.shape == (120, 240, 320) and (h, w) == (240, 320) and np.array_equal(f[0], f[100]) and np.array_equal(f[51], f[49])
    f, h, w = bench.host_content("race", 5, 480, 720)
    assert f.shape == (5, 480, 720) and np.array_equal(f[0], f[2]) and not np.array_equal(f[0], f[1])
    assert bench.host_content("flat", 2, 32, 48)[0].min() == 128
    f, h, w = bench.host_content("pan240x2", 60, 480, 720)          # 51 distinct real frames at 640x480
    assert f.shape == (60, 480, 640) and (h, w) == (480, 640) and len({hash(x.tobytes()) for x in f[:51]}) == 51
    small = bench.host_content("pan240seq", 3, 480, 720)[0]
    assert np.array_equal(f[:3, 0::2, 0::2], small) and int(f[0, 1, 1]) == (int(small[0, :2, :2].astype(int).sum()) + 2) >> 2
    # the PMC summary of a config is read for the kernel the launch plan names, not for whatever comes first in the file
    vals, psha, name = bench.committed_profile("exh720", "k_exh_sea16p<3,5>")
    assert name and name.endswith("_exh720_pmc_summary.txt") and vals["_kernel"].startswith("k_exh_sea16p<") and vals["FETCH_SIZE"] > 0
    assert bench.committed_profile("exh720", "k_no_such_kernel<1>")[0] == {}
    host = bench.host_description()
    assert host["nproc"] >= 1 and host["numpy"] == np.__version__ and host["python"].count(".") == 2
    assert len(bench.kernel_source_sha()) == 16
    assert bench.byte_ops_per_pair(480, 720, 16, 16) == 2891044 * 256


def test_roadmap_model_solves_match_least_squares():
    """roadmap.solve_model (extension, recap_future_updates.md:9-12): from the normal-equation sums alone the
    translation / similarity / affine solves equal np.linalg.lstsq on the underlying block vectors."""
    import roadmap
    rng = np.random.default_rng(12)
    rows = []
    truth = []
    for trial in range(6):
        h, w = 15 + trial, 22 + 2 * trial
        i, j = np.mgrid[0:h, 0:w]
        x, y = (4.0 * i).ravel(), (4.0 * j).ravel()            # motion.py:254-255
        a0, b0, a, b = rng.normal(0, 3), rng.normal(0, 3), rng.normal(0, 0.01), rng.normal(0, 0.01)
        dx = np.rint(a0 - b * x + a * y + rng.normal(0, 0.3, x.size))     # x: row coordinate, dx: column displacement
        dy = np.rint(b0 + a * x + b * y + rng.normal(0, 0.3, x.size))
        wgt = 1.0 / (h * 16 * w * 16)
        A = np.stack([np.ones_like(x), x, y], axis=1)
        F = (A.T @ A) * wgt
        rows.append(np.concatenate([F.ravel(), (A.T @ dx) * wgt, (A.T @ dy) * wgt]))
        truth.append((x, y, dx, dy))
    rows = np.array(rows)
    aff = roadmap.solve_model(rows, "affine")
    tra = roadmap.solve_model(rows, "translation")
    sim = roadmap.solve_model(rows, "similarity")
    for k, (x, y, dx, dy) in enumerate(truth):
        A = np.stack([np.ones_like(x), x, y], axis=1)
        want = np.concatenate([np.linalg.lstsq(A, dx, rcond=None)[0], np.linalg.lstsq(A, dy, rcond=None)[0]])
        np.testing.assert_allclose(aff[k], want, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(tra[k], [dx.mean(), 0, 0, dy.mean(), 0, 0], rtol=1e-10, atol=1e-12)
        D = np.concatenate([np.stack([np.ones_like(x), 0 * x, y, -x], 1), np.stack([0 * x, np.ones_like(x), x, y], 1)])
        th = np.linalg.lstsq(D, np.concatenate([dx, dy]), rcond=None)[0]          # (a0, b0, zoom, rotation)
        np.testing.assert_allclose(sim[k], [th[0], -th[3], th[2], th[1], th[2], th[3]], rtol=1e-8, atol=1e-10)
        assert sim[k][2] == sim[k][4] and sim[k][1] == -sim[k][5]
    with pytest.raises(ValueError):
        roadmap.solve_model(rows, "perspective")
    with pytest.raises(np.linalg.LinAlgError):
        roadmap.solve_model(np.zeros((1, 15)), "similarity")
    with pytest.raises(np.linalg.LinAlgError):
        roadmap.solve_model(np.zeros((1, 15)), "translation")


def test_cli_lists_commands(capsys):
    import gme_cli
    with pytest.raises(SystemExit):
        gme_cli.main(["--help"])
    out = capsys.readouterr().out
    for word in ("bbme", "results", "suggest", "info"):
        assert word in out
    with pytest.raises(SystemExit):
        gme_cli.main(["bbme"])                     # -p and -fi are required, as upstream
    gme_cli.main(["info"])
    out = capsys.readouterr().out
    assert "three-step" in out and "similarity" in out and "device:" in out
