"""Whole-sequence (batch) form of the hot path and its multi-GPU sharding.

The reference processes a video as independent frame pairs (results.py:41-59): pair ``p``
is ``(frames[p], frames[p + fd])`` and no state is carried between pairs.  That is the
only parallel axis the path has, so:

* on one GPU the frames stay resident in HBM (``_gme_native.Sequence``) and every
  kernel launch covers all pairs of the shard;
* across GPUs (one process per GPU) contiguous pair ranges are dealt to the ranks, each
  rank needs its pairs' frames plus ``fd`` halo frames, and the only exchange is one
  all-gather of the per-pair ``float64[6]`` parameter vectors (48 B per pair) at the end
  -- RCCL over xGMI through the library's own communicator (gme_comm_*, gme_shard_gather).
"""
import ctypes
import os
import stat
import time
import types

import numpy as np

import _gme_native as _native
import motion


def shard_range(n_pairs, rank, world):
    """Contiguous pair range [start, stop) of `rank`."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return (rank * n_pairs) // world, ((rank + 1) * n_pairs) // world


def shard_frames(n_pairs, frame_distance, rank, world):
    """(first_frame, n_frames) a rank must hold: its pairs plus the `fd` halo."""
    start, stop = shard_range(n_pairs, rank, world)
    if stop == start:
        return start, 0
    return start, (stop - start) + frame_distance


# ---- the library's own RCCL communicator (gme_comm_*, include/gme_hip.h): no torch involved -------------
class CommUnavailable(RuntimeError):
    """Raised by comm_init on EVERY rank of the launch when at least one rank cannot bring the RCCL
    communicator up (no librccl, no id, ncclCommInitRank failed): the ranks can then take the same
    fallback, or all exit; none is left waiting inside a collective."""


def _process_start_time():
    """Wall-clock time this process was started (the ranks of one launch start within a second or so of
    each other; files older than that belong to an earlier launch)."""
    try:
        with open("/proc/self/stat") as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])          # field 22: starttime in clock ticks since boot
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return time.time() - (up - ticks / os.sysconf("SC_CLK_TCK"))
    except Exception:                                                      # noqa: BLE001
        return _IMPORT_TIME


_IMPORT_TIME = time.time()
_STALE_SLACK_S = 60.0        # ranks of one launch may start (and publish) this much apart; an earlier launch's files are older


def _rendezvous_base():
    """Path prefix of this launch's rendezvous files.  GME_COMM_ID_FILE names it outright; otherwise it lives in
    a directory only this user can enter (0700, checked) and is keyed by MASTER_PORT and the pid of the ranks'
    common parent (torch.distributed.run's agent, or the test's spawner)."""
    explicit = os.environ.get("GME_COMM_ID_FILE")
    if explicit:
        return explicit
    d = os.path.join(os.environ.get("GME_COMM_DIR", "/tmp"), "gme_rccl_%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError("%s is not a private directory of uid %d (mode %o, owner %d)" % (d, os.getuid(), st.st_mode & 0o7777, st.st_uid))
    try:                                         # day-old leftovers of earlier launches (ack files, crashed attempts)
        for name in os.listdir(d):
            path = os.path.join(d, name)
            st1 = os.lstat(path)
            if stat.S_ISREG(st1.st_mode) and st1.st_uid == os.getuid() and st1.st_mtime < time.time() - 86400:
                os.unlink(path)
    except OSError:
        pass
    return os.path.join(d, "%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))


class Rendezvous:
    """File-based exchange of small blobs between the ranks of one launch on one node: publish(key, blob) writes
    <base>.<key>.<rank> atomically (0600, O_EXCL | O_NOFOLLOW); collect(key) returns every rank's blob once all are
    there.  A file is accepted only if it is a regular file of this user and not older than this process (minus a few
    minute of start-up skew): what an earlier launch with the same port and parent left behind is never read."""

    _attempts = {}               # base -> rendezvous objects this process has made for it (ranks make them collectively)

    def __init__(self, rank, world, timeout_s=180.0, base=None):
        self.rank, self.world, self.timeout_s = int(rank), int(world), float(timeout_s)
        self.base = base or _rendezvous_base()
        self.not_before = _process_start_time() - _STALE_SLACK_S
        self._mine = []
        # Attempt nonce: a second bring-up in the same processes, or a worker group restarted by the same
        # torchrun agent (same port, same parent pid), must never read the first attempt's files -- a stale "no:" marker
        # or, worse, a stale ncclUniqueId (ranks entering ncclCommInitRank with different ids hang for good).  Every
        # rank of a launch makes its rendezvous objects in the same order, so the counter agrees across ranks.
        n = Rendezvous._attempts.get(self.base, 0)
        Rendezvous._attempts[self.base] = n + 1
        self.attempt = "a%s_%d" % (os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), n)

    def _path(self, key, rank):
        return "%s.%s.%s.%d" % (self.base, self.attempt, key, rank)

    def publish(self, key, blob):
        path = self._path(key, self.rank)
        tmp = "%s.tmp%d" % (path, os.getpid())
        for stale in (tmp, path):
            try:
                os.unlink(stale)
            except FileNotFoundError:
                pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(blob)
        os.replace(tmp, path)                    # atomic: readers see nothing or everything
        self._mine.append(path)

    def _read(self, key, rank):
        try:
            fd = os.open(self._path(key, rank), os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
        except (FileNotFoundError, OSError):
            return None
        with os.fdopen(fd, "rb") as f:
            st = os.fstat(f.fileno())
            if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or st.st_mtime < self.not_before:
                return None
            return f.read()

    def collect(self, key, ranks=None):
        ranks = list(range(self.world)) if ranks is None else list(ranks)
        got, t0 = {}, time.time()
        while True:
            for r in ranks:
                if r not in got:
                    blob = self._read(key, r)
                    if blob is not None:
                        got[r] = blob
            if len(got) == len(ranks):
                return [got[r] for r in ranks]
            if time.time() - t0 > self.timeout_s:
                missing = [r for r in ranks if r not in got]
                raise TimeoutError("rank %d: no '%s' from rank(s) %s at %s.* after %.0f s" % (self.rank, key, missing, self.base, self.timeout_s))
            time.sleep(0.01)

    def agree(self, key, ok, payload=b""):
        """Every rank says ok / not ok (with a payload or a reason); returns all payloads if every rank was ok,
        raises CommUnavailable on EVERY rank otherwise."""
        self.publish(key, (b"ok:" if ok else b"no:") + payload)
        blobs = self.collect(key)
        bad = ["rank %d: %s" % (r, b[3:].decode("utf-8", "replace")) for r, b in enumerate(blobs) if not b.startswith(b"ok:")]
        if bad:
            self.fail(key, "%s failed on %d of %d ranks (%s)" % (key, len(bad), self.world, "; ".join(bad)))
        return [b[3:] for b in blobs]

    def fail(self, key, message):
        """Raise CommUnavailable(message) -- called with the same verdict on every rank.  One more round first says that
        every rank has READ every file; then each rank removes its own.  The 4-byte ack files themselves stay (a rank
        that is done may exit before the slowest rank has read its ack; the attempt nonce keeps a later bring-up away
        from them, and _rendezvous_base sweeps day-old files out of the private directory)."""
        try:
            self.publish("ack_" + key, b"read")
            self.collect("ack_" + key)
            self._mine = [p for p in self._mine if ".ack_" not in os.path.basename(p)]
            self.cleanup()
        except TimeoutError:
            pass
        raise CommUnavailable(message)

    def cleanup(self):
        """Remove this rank's own files (call only once every rank is known to have read them)."""
        for path in self._mine:
            try:
                os.unlink(path)
            except OSError:
                pass
        self._mine = []


def comm_exchange_id(make_id, rank, world, timeout_s=180.0, rdv=None):
    """Rank 0 calls make_id() -> bytes and publishes them; every rank returns the same bytes.  If rank 0 cannot
    make one it says so through the same file and the others stop waiting at once."""
    rdv = rdv or Rendezvous(rank, world, timeout_s)
    if rank == 0:
        try:
            blob = make_id()
        except Exception:
            rdv.publish("id", b"FAILED")
            raise
        rdv.publish("id", blob)
        return blob
    blob = rdv.collect("id", [0])[0]
    if blob == b"FAILED":
        raise RuntimeError("rank 0 could not create an RCCL id")
    return blob


def collective_init(rdv, probe, make_id, init, destroy, device_tag=None):
    """The agreement around ncclCommInitRank, which cannot time out: (1) every rank loads RCCL (`probe`) and rank 0
    makes the id; all ranks learn whether all could, and nobody enters the collective otherwise; (2) every rank runs
    `init(id)`; all ranks learn whether all succeeded, and those that did `destroy()` their communicator
    otherwise.  Either every rank returns, or every rank raises CommUnavailable -- PROVIDED every rank that passed
    phase 1 really enters ncclCommInitRank in `init`: a rank whose `init` fails before the collective (device lost
    between probe and init, out of memory) leaves its peers inside a call that cannot time out, and only the
    launcher's own timeout ends the launch.  Everything that can fail locally therefore belongs in `probe`
    (comm_init: gme_comm_probe loads RCCL AND binds the context's device; the payload carries the device's PCI bus id
    and phase 1 refuses two ranks on one device, which RCCL would only refuse inside the collective)."""
    blob, err, tag = b"", None, b""
    try:
        probe()
        if device_tag is not None:
            tag = device_tag()
        if rdv.rank == 0:
            blob = make_id()
    except Exception as e:                        # noqa: BLE001 -- reported to every rank below
        err = e
    payload = len(tag).to_bytes(2, "little") + tag + blob
    blobs = rdv.agree("probe", err is None, payload if err is None else repr(err).encode())
    tags = [b[2:2 + int.from_bytes(b[:2], "little")] for b in blobs]
    shared = sorted({t for t in tags if t and tags.count(t) > 1})
    if shared and not os.environ.get("GME_COMM_ALLOW_SHARED_DEVICE"):
        # every rank sees the same tags, so every rank raises -- before anybody is inside ncclCommInitRank
        rdv.fail("probe", "probe: ranks share a device (%s): RCCL wants one device per rank" % ", ".join(
            "%s on ranks %s" % (t.decode("utf-8", "replace"), [r for r, x in enumerate(tags) if x == t]) for t in shared))
    uid = blobs[0][2 + len(tags[0]):]
    err = None
    try:
        init(uid)
    except Exception as e:                        # noqa: BLE001
        err = e
    try:
        rdv.agree("init", err is None, b"" if err is None else repr(err).encode())
    except CommUnavailable:
        if err is None:
            destroy()
        raise


def comm_init(ctx, rank, world, timeout_s=180.0):
    """Collective: give `ctx` an RCCL communicator spanning the `world` ranks of this node.  Raises
    CommUnavailable on every rank if any rank cannot (see collective_init)."""
    lib = ctx.lib

    def probe():
        _native._check(lib.gme_comm_probe(), lib)

    def make_id():
        buf = ctypes.create_string_buffer(128)
        _native._check(lib.gme_comm_unique_id(buf), lib)
        return buf.raw

    def init(blob):
        _native._check(lib.gme_comm_init(ctx.handle, blob, rank, world), lib)

    def device_tag():
        buf = ctypes.create_string_buffer(64)
        _native._check(lib.gme_device_bus_id(ctx.handle, buf, 64), lib)     # also binds the device: fails HERE, not in init
        return buf.value

    if world == 1:
        probe()
        init(make_id())
        comm_barrier(ctx)
        return
    rdv = Rendezvous(rank, world, timeout_s)
    collective_init(rdv, probe, make_id, init, lambda: comm_destroy(ctx), device_tag)
    comm_barrier(ctx)                            # every rank has read every file once this returns
    rdv.cleanup()


def comm_info(ctx):
    """(rank, world) as RCCL itself reports them for the context's communicator (ncclCommUserRank / ncclCommCount)."""
    r, n = ctypes.c_int(-1), ctypes.c_int(-1)
    _native._check(ctx.lib.gme_comm_info(ctx.handle, ctypes.byref(r), ctypes.byref(n)), ctx.lib)
    return r.value, n.value


def comm_destroy(ctx):
    _native._check(ctx.lib.gme_comm_destroy(ctx.handle), ctx.lib)


def comm_max(ctx, value):
    v = np.array([float(value)], dtype=np.float64)
    _native._check(ctx.lib.gme_comm_allreduce_max(ctx.handle, v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1), ctx.lib)
    return float(v[0])


def comm_barrier(ctx):
    comm_max(ctx, 0.0)


def pad_and_trim(n_pairs, world):
    """(rows of the largest shard, [(start, stop)] per rank): the fixed block size of the all-gather
    and how to cut the padding off again."""
    sizes = [shard_range(n_pairs, r, world) for r in range(world)]
    return max(1, max(b - a for a, b in sizes)), sizes


def gather_parameters_rccl(ctx, local, n_pairs, rank, world):
    """gather_parameters over the library's RCCL communicator (gme_shard_gather)."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    k = local.shape[1] if local.ndim == 2 else 6
    longest, sizes = pad_and_trim(n_pairs, world)
    out = np.empty((world, longest, k), dtype=np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    _native._check(ctx.lib.gme_shard_gather(ctx.handle, local.ctypes.data_as(dp), len(local), k, longest, out.ctypes.data_as(dp)),
                   ctx.lib)
    return np.concatenate([out[r, :b - a] for r, (a, b) in enumerate(sizes)], axis=0)


def psnr_from_sse(sse, height, width, exact=True):
    """utils.PSNR (utils.py:100-116) from sums of squared error: -1 where the frames are equal,
    else the real part of the reference's complex 20 log10(255 / sqrt(mse)).

    ``exact`` evaluates it per pair with cmath like the reference (results.py writes
    ``str(value)`` into psnr_records.json, so the last bit shows); the vectorised NumPy form
    differs from it by at most 2e-14 dB and is what the batch hot path uses."""
    mse = np.asarray(sse, dtype=np.float64) / (height * width)
    out = np.full(len(mse), -1.0)
    if exact:
        from cmath import log10, sqrt
        for k, m in enumerate(mse):
            if m != 0:
                out[k] = (20 * log10(255.0 / sqrt(m))).real
        return out
    nz = mse != 0
    out[nz] = 20.0 * np.log10(255.0 / np.sqrt(mse[nz]))
    return out


class _Lane:
    """One stream's part of a shard: its own context (HIP stream) and resident frames."""

    def __init__(self, ctx, seq, lo, hi):
        self.ctx, self.seq, self.lo, self.hi = ctx, seq, lo, hi    # local pair range [lo, hi)


class ShardedSequence:
    """This rank's slice of a video: frames in HBM, pair-level results on demand.

    ``streams`` > 1 cuts the slice once more into that many contiguous pair ranges, each with
    its own context (= HIP stream) on the same GPU and driven by its own host thread.  The
    staged estimate (motion.estimate_sequence) has the host solve 3x3 systems between device
    stages; with several streams one range's solves and launch latencies are covered by the
    other ranges' kernels.  Results do not depend on ``streams``.

    ``interleave=True`` drives the ranges from ONE host thread instead, through the split-phase calls
    of the C ABI (gme_seq_set_split_phase / gme_seq_wait): stage s of every range is queued, then each
    range's result is awaited and solved in turn while the other ranges' kernels run.  No host threads
    contend for the interpreter or the BLAS; estimate_and_compensate() is the call that uses it.
    """

    def __init__(self, height, width, n_frames, frame_distance=1, rank=0, world=1, ctx=None, streams=1, interleave=False):
        self.ctx = ctx or _native.default_context()
        self.interleave = bool(interleave)
        self.H, self.W, self.fd = int(height), int(width), int(frame_distance)
        self.n_frames_total = int(n_frames)
        self.n_pairs_total = max(0, self.n_frames_total - self.fd)
        self.rank, self.world = rank, world
        self.pair_start, self.pair_stop = shard_range(self.n_pairs_total, rank, world)
        self.first_frame, n_local = shard_frames(self.n_pairs_total, self.fd, rank, world)
        self.lanes = []
        n_loc = self.pair_stop - self.pair_start
        k = max(1, min(int(streams), n_loc))
        for j in range(k if n_local else 0):
            lo, hi = shard_range(n_loc, j, k)
            c = self.ctx if j == 0 else _native.Context(self.ctx.device)
            n_fr = (hi - lo) + self.fd if k > 1 else max(n_local, self.fd + 1)
            self.lanes.append(_Lane(c, _native.Sequence(c, n_fr, self.H, self.W), lo, hi))
        self.seq = self.lanes[0].seq if len(self.lanes) == 1 else None     # single-stream shards: the sequence itself
        self._pool = None

    @property
    def n_pairs(self):
        return self.pair_stop - self.pair_start

    def _each(self, fn):
        """fn(lane) for every lane, one host thread per lane; results in lane order."""
        if len(self.lanes) <= 1:
            return [fn(lane) for lane in self.lanes]
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(len(self.lanes))
        return list(self._pool.map(fn, self.lanes))

    def _lane_of(self, pair):
        for lane in self.lanes:
            if lane.lo <= pair < lane.hi:
                return lane, pair - lane.lo
        raise IndexError("pair %d is not in this shard" % pair)

    def close(self):
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
        for lane in self.lanes:
            lane.seq.close()
            if lane.ctx is not self.ctx:
                lane.ctx.close()
        self.lanes, self.seq = [], None

    def sync(self):
        for lane in self.lanes:
            lane.ctx.sync()

    def invalidate(self):
        """Mark everything derived from the frames stale (new video in the same buffers)."""
        for lane in self.lanes:
            lane.seq.invalidate_pyramids()

    def load(self, frames):
        """`frames` is the WHOLE video (uint8[N, H, W] or a list); only this rank's slice is uploaded:
        one copy per lane (its frames back to back), the lanes' copies in their own host threads so that
        one lane's upload runs beside another lane's kernels."""
        def run(lane):
            g0 = self.first_frame + lane.lo
            n = max(0, min(lane.seq.N, len(frames) - g0))
            if n == 0:
                return
            if isinstance(frames, np.ndarray) and frames.ndim == 3 and frames.dtype == np.uint8:
                lane.seq.upload(0, frames[g0:g0 + n])
            else:
                lane.seq.upload(0, np.stack([np.asarray(f, dtype=np.uint8) for f in frames[g0:g0 + n]]))
        self._each(run)

    def synth(self, seed, t0=0):
        """Generate this rank's slice of the synthetic sequence `seed` (video frame k = time t0 + k) on its own GPU."""
        for lane in self.lanes:
            lane.seq.synth(seed, t0 + self.first_frame + lane.lo)

    def motion_fields(self, block_size, search_window, procedure, pnorm):
        """bbme.get_motion_field for every local pair -> int32[P_local, h, w, 2]."""
        if not self.lanes:
            return np.zeros((0, int(self.H / block_size), int(self.W / block_size), 2), np.int32)

        def run(lane):
            lane.seq.bbme(self.fd, block_size, search_window, procedure, pnorm)
            return lane.seq.read_mv(0, lane.hi - lane.lo)
        return np.concatenate(self._each(run), axis=0)

    def estimate(self, procedure=3, search_window=2):
        """motion.global_motion_estimation for every local pair -> float64[P_local, 6]."""
        if not self.lanes:
            return np.zeros((0, 6))
        if self.interleave and len(self.lanes) > 1:
            return self._interleaved(procedure, search_window, False)[0]
        return np.concatenate(self._each(
            lambda lane: motion.estimate_sequence(lane.seq, self.fd, procedure, search_window)[:lane.hi - lane.lo]), axis=0)

    def _psnr(self, sse, exact=True):
        return psnr_from_sse(sse, self.H, self.W, exact)

    def compensate(self, params):
        """results.py:52-59,109 for every local pair -> PSNR(current, compensated) as float64[P_local]."""
        if not self.lanes:
            return np.zeros(0)
        params = np.asarray(params, dtype=np.float64)
        sse = self._each(lambda lane: lane.seq.compensate(self.fd, int(motion.BBME_BLOCK_SIZE),
                                                          params[lane.lo:lane.hi])[:lane.hi - lane.lo])
        return self._psnr(np.concatenate(sse))

    def estimate_and_compensate(self, procedure=3, search_window=2, exact_psnr=False):
        """estimate() then compensate() per stream without a join in between -> (params[P,6], psnr[P])."""
        if not self.lanes:
            return np.zeros((0, 6)), np.zeros(0)

        if os.environ.get("GME_DEVICE_SOLVE") == "1":
            got = self._device_solved(procedure, search_window, exact_psnr)
            if got is not None:
                return got                        # else: a pair sat on a rounding tie (or was singular): the host path below

        if self.interleave and len(self.lanes) > 1:
            return self._interleaved(procedure, search_window, True, exact_psnr)

        def run(lane):
            n = lane.hi - lane.lo
            p = motion.estimate_sequence(lane.seq, self.fd, procedure, search_window)[:n]
            return p, lane.seq.compensate(self.fd, int(motion.BBME_BLOCK_SIZE), p)[:n]
        parts = self._each(run)
        return np.concatenate([p for p, _ in parts], axis=0), self._psnr(np.concatenate([s for _, s in parts]), exact_psnr)

    def _device_solved(self, procedure, search_window, exact_psnr=False):
        """estimate_and_compensate() with the 3x3 solves on the device (GME_DEVICE_SOLVE=1, gme_seq_gme_device_solve): every
        lane's whole estimate is queued in one call, one wait per lane.  Parameters are within rtol 1e-10 of the host
        path's (LAPACK's last bits are not reproduced); model fields, masks, compensated frames and PSNR are bit-equal to
        it -- unless the library flags a pair (a model displacement within 1e-9 of a rounding tie, or a singular system):
        then nothing is returned and the caller runs the host path, which also raises upstream's LinAlgError."""
        frac = float(motion.MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE)
        bs = int(motion.BBME_BLOCK_SIZE)
        lanes = self.lanes
        for lane in lanes:
            lane.seq.set_split_phase(True)
        try:
            pending = [lane.seq.gme_device_solve(self.fd, bs, frac, procedure, search_window) for lane in lanes]
            params, sse, clean = [], [], True
            for lane, (p, e, f) in zip(lanes, pending):
                lane.seq.wait()
                n = lane.hi - lane.lo
                clean = clean and not np.any(f[:n])
                params.append(np.array(p[:n]))
                sse.append(np.array(e[:n]))
                lane.ctx.sync()                   # drains the stream and reports a walk that overran its guard
        finally:
            for lane in lanes:
                lane.seq.set_split_phase(False)
        if not clean:
            return None
        return np.concatenate(params, axis=0), self._psnr(np.concatenate(sse), exact_psnr)

    def _interleaved(self, procedure, search_window, compensate, exact_psnr=False):
        """estimate() / estimate_and_compensate() for several ranges from one host thread: every stage is queued on
        all streams before the first result is awaited, so range k's projection and 3x3 solves (motion.py:191-207,
        262-282, the same arithmetic as motion.estimate_sequence) run while the other ranges' searches do.
        -> (params[P, 6], psnr[P] or None)"""
        frac = float(motion.MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE)
        bs = int(motion.BBME_BLOCK_SIZE)
        lanes = self.lanes
        for lane in lanes:
            lane.seq.set_split_phase(True)
        try:
            # first parameters, their projection and the level-1 fit in one queued call (gme_seq_gme_begin_fit)
            pending = [lane.seq.gme_begin_fit(self.fd, bs, frac, procedure, search_window)[1] for lane in lanes]
            for k, lane in enumerate(lanes):
                lane.seq.wait()
                p = motion._solve_batch(pending[k])        # level 1 solved, projected in float64, level 2 asked for
                p[:, 0] = p[:, 0] * 2
                p[:, 3] = p[:, 3] * 2
                pending[k] = lane.seq.gme_fit(2, p, frac)
            params, sse = [None] * len(lanes), [None] * len(lanes)
            for k, lane in enumerate(lanes):
                lane.seq.wait()
                params[k] = motion._solve_batch(pending[k])
                if compensate:
                    sse[k] = lane.seq.compensate(self.fd, bs, params[k])
            out_sse = []
            for k, lane in enumerate(lanes):
                if compensate:
                    lane.seq.wait()
                    out_sse.append(np.array(sse[k][:lane.hi - lane.lo]))
                lane.ctx.sync()                   # drains the stream and reports a walk that overran its guard
        finally:
            for lane in lanes:
                lane.seq.set_split_phase(False)   # the other methods of the class use the blocking calls
        p_all = np.concatenate([p[:lane.hi - lane.lo] for p, lane in zip(params, lanes)], axis=0)
        return p_all, (self._psnr(np.concatenate(out_sse), exact_psnr) if compensate else None)

    def read_compensated(self, pair):
        lane, k = self._lane_of(pair)
        return lane.seq.read_compensated(k)

    def gather(self, local_rows):
        """All-gather of this shard's per-pair rows (float64[P_local, k]) -> float64[P_total, k] on every rank, over the
        context's RCCL communicator (comm_init first); the identity on a single rank."""
        if self.world == 1:
            return np.ascontiguousarray(local_rows, dtype=np.float64)
        return gather_parameters_rccl(self.ctx, local_rows, self.n_pairs_total, self.rank, self.world)

    def unpad(self, gathered):
        """float64[world, n_max, k] blocks of a fixed-size all-gather -> float64[P_total, k]."""
        _, sizes = pad_and_trim(self.n_pairs_total, self.world)
        return np.concatenate([gathered[r, :b - a] for r, (a, b) in enumerate(sizes)], axis=0)


class StreamEstimator:
    """results.py:41-59,109 for videos that still live in HOST memory: ``run(frames)`` -> (params float64[P, 6], psnr float64[P]).

    A video is cut into chunks of at most ``chunk_pairs`` pairs (plus the ``fd`` halo frames; see schedule()); ``streams`` lanes -- each its own
    context, HIP stream and (chunk_pairs + fd)-frame device sequence, allocated once here and reused by every run() --
    take the chunks in turn.  Everything a lane does is split-phase (gme_seq_set_split_phase): the upload of its chunk
    (on the device's shared upload stream), pyramids + dense field, level fits, compensation are queued and ONE host
    thread serves whichever lane has its result (gme_seq_poll), doing its projection / 3x3 solves (motion.py:191-207,
    262-282) and queueing its next stage.  While one lane's frames cross the link the other lanes' kernels run, so a
    video larger than HBM streams through at link speed.

    ``frames``: uint8[N, H, W] (page-locked memory from _gme_native.pinned_empty crosses at link speed and never holds
    the host thread) or a list of 2-D uint8 arrays (gathered chunk by chunk into a page-locked buffer per lane).
    ``compensated``: optional uint8[P, H, W] array that receives every compensated frame (results.py:59 writes them out).
    ``solve``: float64[n, 15] normal-equation sums -> float64[n, 6] parameters; default motion._solve_batch (the
    reference's two 3x3 systems), roadmap.solve_model for the other motion models.
    Results equal the resident path bit for bit (tests/test_gpu_round3.py)."""

    def __init__(self, height, width, frame_distance=1, chunk_pairs=512, streams=2, ctx=None, procedure=3, search_window=2,
                 min_chunk=64):
        self.H, self.W, self.fd = int(height), int(width), int(frame_distance)
        self.chunk_pairs = max(1, int(chunk_pairs))
        self.min_chunk = max(1, min(int(min_chunk), self.chunk_pairs))
        self.cap = self.chunk_pairs + self.fd                    # frames a lane holds
        self.procedure, self.search_window = procedure, search_window
        self.ctx = ctx or _native.default_context()
        self.lanes = []
        try:
            for j in range(max(1, int(streams))):
                c = self.ctx if j == 0 else _native.Context(self.ctx.device)
                seq = _native.Sequence(c, self.cap, self.H, self.W)
                seq.set_split_phase(True)
                self.lanes.append(types.SimpleNamespace(ctx=c, seq=seq, host=None, chunk=None, stage=0, pending=None, params=None))
        except BaseException:
            self.close(check=False)
            raise

    def close(self, check=True):
        """Release the lanes' device objects; with ``check`` a walk that overran its guard is reported (gme_sync)."""
        lanes, self.lanes = self.lanes, []
        err = None
        for lane in lanes:
            try:
                lane.seq.set_split_phase(False)
                if check:
                    lane.ctx.sync()
            except Exception as e:                                   # noqa: BLE001 -- release everything, then report
                err = err or e
            finally:
                lane.seq.close()
                if lane.ctx is not self.ctx:
                    lane.ctx.close()
        if err is not None and check:
            raise err

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close(check=exc[0] is None)

    def schedule(self, n_pairs):
        """Chunks [(first pair, end pair)] of a video of `n_pairs` pairs: as large as a lane holds while much is left (each
        chunk costs the host three round trips whatever its size), shrinking towards the end -- a quarter of what is left per
        lane, at least `min_chunk` -- because the estimate of the LAST chunk is the one stretch no upload hides (traced:
        with equal chunks of 1024 pairs the link idles 2.5 of 15.9 ms behind the last copy, with equal chunks of 128 the
        host's round trips leave gaps between the copies)."""
        out, p0 = [], 0
        while p0 < n_pairs:
            left = n_pairs - p0
            c = min(self.chunk_pairs, max(self.min_chunk, -(-left // (2 * max(1, len(self.lanes))))), left)
            out.append((p0, p0 + c))
            p0 += c
        return out

    def run(self, frames, compensated=None, exact_psnr=True, solve=None, on_chunk=None):
        """``on_chunk(p0, p1, comp, params, psnr)`` is called as each chunk finishes (chunks of different lanes may finish out
        of order) with its compensated frames uint8[p1 - p0, H, W] (one read for the chunk; the array is reused by the next
        chunk of the same lane), parameters and PSNR: what results.py writes per pair, without a whole-video array on the host."""
        fd, H, W, cap = self.fd, self.H, self.W, self.cap
        P = max(0, len(frames) - fd)
        params_out, sse_out = np.zeros((P, 6)), np.zeros(P, dtype=np.int64)
        if P == 0:
            return params_out, np.zeros(0)
        if tuple(np.asarray(frames[0]).shape) != (H, W):
            raise ValueError("frames of %r do not fit this estimator's %r" % (np.asarray(frames[0]).shape, (H, W)))
        chunks = self.schedule(P)
        stacked = isinstance(frames, np.ndarray) and frames.ndim == 3 and frames.dtype == np.uint8 and frames.flags.c_contiguous
        frac = float(motion.MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE)
        bs = int(motion.BBME_BLOCK_SIZE)
        solve = solve or motion._solve_batch

        def start(lane, chunk):
            lane.chunk, lane.stage = chunk, 1
            p0, p1 = chunk
            n = p1 - p0 + fd
            if stacked:
                src = frames[p0:p0 + n]
            else:
                if lane.host is None:
                    lane.host = _native.pinned_empty((cap, H, W))
                for k in range(n):
                    lane.host[k] = frames[p0 + k]
                src = lane.host[:n]
            lane.seq.set_frames(n)               # the stages cover this chunk's pairs only; buffers stay sized for `cap`
            lane.seq.upload(0, src)              # queued; `src` stays alive (lane.host / the caller's array)
            # Nothing else is queued on the lane yet: a kernel launch behind a cross-stream wait on a copy holds the calling
            # thread until that copy is done (measured: one gme_begin_fit call of 7.6 ms, the upload of two chunks), and the
            # one host thread must stay free for the other lanes.  Stage 1 ends when the upload's event has arrived.

        def advance(lane):
            """Take the lane's result (it has arrived) and queue its next stage -> True when the chunk is finished."""
            seq, (p0, p1) = lane.seq, lane.chunk
            n = p1 - p0
            seq.wait()
            if lane.stage == 1:                  # the chunk is on the device: dense field, first parameters, level-1 fit
                lane.pending = seq.gme_begin_fit(fd, bs, frac, self.procedure, self.search_window)[1]
                lane.stage = 2
                return False
            if lane.stage == 2:                  # level-1 sums are back: solve, project (float64), ask for level 2
                p = solve(lane.pending[:n])
                p[:, 0] = p[:, 0] * 2
                p[:, 3] = p[:, 3] * 2
                lane.pending = seq.gme_fit(2, p, frac)
                lane.stage = 3
                return False
            if lane.stage == 3:
                lane.params = solve(lane.pending[:n])
                lane.pending = seq.compensate(fd, bs, lane.params)
                lane.stage = 4
                return False
            params_out[p0:p1] = lane.params
            sse_out[p0:p1] = lane.pending[:n]
            if compensated is not None:
                seq.read_compensated_range(0, n, compensated[p0:p1])      # one wait for the chunk, not one per pair
            if on_chunk is not None:
                if compensated is not None:
                    comp = compensated[p0:p1]
                else:
                    if getattr(lane, "comp_host", None) is None:
                        lane.comp_host = np.empty((cap, H, W), np.uint8)
                    comp = seq.read_compensated_range(0, n, lane.comp_host[:n])
                on_chunk(p0, p1, comp, lane.params[:n], psnr_from_sse(lane.pending[:n], H, W, exact_psnr))
            lane.stage = 0
            return True

        todo = list(reversed(chunks))
        busy = []
        for lane in self.lanes:
            if todo:
                start(lane, todo.pop())
                busy.append(lane)
        while busy:
            # serve whichever lane has its result: a lane that waits for its frames must not hold up one whose sums are
            # back; the lanes furthest along go first, so that their next chunk's upload is queued as early as possible
            progressed = False
            for lane in sorted(busy, key=lambda l: -l.stage):
                if lane.seq.poll():
                    progressed = True
                    if advance(lane):
                        if todo:
                            start(lane, todo.pop())
                        else:
                            busy.remove(lane)
            if not progressed:
                time.sleep(2e-5)                 # nothing ready: look again shortly (blocking on ONE lane's event would sit out
                                                 # another lane's result: traced, a lane idled 3.5 ms behind its finished upload)
        for lane in self.lanes:
            lane.ctx.sync()                      # drains the lane and reports a walk that overran its guard
        return params_out, psnr_from_sse(sse_out, H, W, exact_psnr)


def estimate_stream(frames, frame_distance=1, chunk_pairs=512, streams=2, ctx=None, compensated=None, procedure=3, on_chunk=None,
                    search_window=2, exact_psnr=True, solve=None, min_chunk=64):
    """One-shot StreamEstimator: allocate the lanes, run `frames` through them, release them
    -> (params float64[P, 6], psnr float64[P]).  Setting the lanes up costs a few milliseconds each; callers with
    several videos of one size keep a StreamEstimator."""
    n_frames = len(frames)
    P = max(0, n_frames - int(frame_distance))
    if P == 0:
        return np.zeros((0, 6)), np.zeros(0)
    H, W = np.asarray(frames[0]).shape
    chunk_pairs = max(1, min(int(chunk_pairs), P))
    n_chunks = (P + chunk_pairs - 1) // chunk_pairs
    with StreamEstimator(H, W, frame_distance, chunk_pairs, max(1, min(int(streams), n_chunks)), ctx, procedure, search_window,
                         min_chunk=min_chunk) as est:
        return est.run(frames, compensated=compensated, exact_psnr=exact_psnr, solve=solve, on_chunk=on_chunk)
