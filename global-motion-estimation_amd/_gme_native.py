"""ctypes binding of libgme_hip.so (include/gme_hip.h) -- the only way the Python
surface reaches the GPU.  No PyTorch, no fallback: if the library or a gfx950 device
is missing every call raises.

Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C global-motion-estimation_amd/csrc``.
"""
import ctypes
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgme_hip.so")

GME_OK, ERR_ARG, ERR_GEOMETRY, ERR_HIP, ERR_STATE, ERR_NOMEM = 0, -1, -3, -4, -5, -6


class GmeError(RuntimeError):
    """HIP/runtime failure inside libgme_hip.so."""


_c_u8p = ctypes.POINTER(ctypes.c_uint8)
_c_i16p = ctypes.POINTER(ctypes.c_int16)
_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_i64p = ctypes.POINTER(ctypes.c_int64)
_c_f32p = ctypes.POINTER(ctypes.c_float)
_c_f64p = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p
_i = ctypes.c_int

_SIGNATURES = {
    "gme_last_error": (ctypes.c_char_p, []),
    "gme_device_count": (_i, []),
    "gme_create": (_vp, [_i]),
    "gme_destroy": (None, [_vp]),
    "gme_sync": (_i, [_vp]),
    "gme_stream": (_vp, [_vp]),
    "gme_device_info": (_i, [_vp, ctypes.c_char_p, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "gme_device_bus_id": (_i, [_vp, ctypes.c_char_p, _i]),
    "gme_last_bbme_info": (_i, [_vp, ctypes.c_char_p, _i, _c_i64p, _c_i64p, _c_i64p]),
    "gme_last_bbme_listed": (_i, [_vp, _c_i64p]),
    "gme_timer_start": (_i, [_vp]),
    "gme_timer_stop": (_i, [_vp, _c_f32p]),
    "gme_bbme_u8": (_i, [_vp, _c_u8p, _c_u8p, _i, _i, _i, _i, _i, _i, _i, _c_i32p]),
    "gme_pyrdown_u8": (_i, [_vp, _c_u8p, _i, _i, _i, _c_u8p]),
    "gme_affine_field": (_i, [_vp, _c_f64p, _i, _i, _c_i16p]),
    "gme_compensate_u8": (_i, [_vp, _c_u8p, _i, _i, _i, _c_i32p, _i, _i, _c_u8p]),
    "gme_sse_u8": (_i, [_vp, _c_u8p, _c_u8p, _i, _i, _i, _i, _c_i64p]),
    "gme_seq_create": (_vp, [_vp, _i, _i, _i]),
    "gme_seq_destroy": (None, [_vp]),
    "gme_seq_upload": (_i, [_vp, _i, _i, _c_u8p, _i, ctypes.c_int64]),
    "gme_seq_set_frames": (_i, [_vp, _i]),
    "gme_seq_synth": (_i, [_vp, ctypes.c_uint64, _i]),
    "gme_seq_read_frame": (_i, [_vp, _i, _i, _c_u8p]),
    "gme_seq_invalidate": (_i, [_vp]),
    "gme_seq_bbme": (_i, [_vp, _i, _i, _i, _i, _i]),
    "gme_seq_read_mv": (_i, [_vp, _i, _i, _c_i32p]),
    "gme_seq_bbme_streamed": (_i, [_vp, _c_u8p, _i, ctypes.c_int64, _i, _i, _i, _i, _i, _i, _i, _c_i32p]),
    "gme_host_alloc": (_vp, [ctypes.c_size_t]),
    "gme_host_free": (None, [_vp]),
    "gme_comm_probe": (_i, []),
    "gme_comm_info": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "gme_seq_mv_summary": (_i, [_vp, _c_f64p]),
    "gme_seq_mv_summary_gather": (_i, [_vp, _i, _c_f64p]),
    "gme_comm_unique_id": (_i, [ctypes.c_char_p]),
    "gme_comm_init": (_i, [_vp, ctypes.c_char_p, _i, _i]),
    "gme_comm_destroy": (_i, [_vp]),
    "gme_shard_gather": (_i, [_vp, _c_f64p, _i, _i, _i, _c_f64p]),
    "gme_comm_allreduce_max": (_i, [_vp, _c_f64p, _i]),
    "gme_seq_gme_begin": (_i, [_vp, _i, _i, _i, _i, _c_f32p]),
    "gme_seq_gme_fit": (_i, [_vp, _i, _c_f64p, ctypes.c_double, _c_f64p]),
    "gme_seq_gme_begin_fit": (_i, [_vp, _i, _i, _i, _i, ctypes.c_double, _c_f32p, _c_f64p]),
    "gme_seq_gme_read_stage": (_i, [_vp, _i, _i, _c_i32p, _c_i16p, _c_u8p, _c_i64p]),
    "gme_seq_compensate": (_i, [_vp, _i, _i, _c_f64p, _c_i64p]),
    "gme_solve_fit_sums": (_i, [_vp, _c_f64p, _i, _i, _i, _i, _c_f64p, _c_i32p]),
    "gme_seq_gme_device_solve": (_i, [_vp, _i, _i, _i, _i, ctypes.c_double, _c_f64p, _c_i64p, _c_i32p]),
    "gme_seq_read_compensated": (_i, [_vp, _i, _c_u8p]),
    "gme_seq_read_compensated_range": (_i, [_vp, _i, _i, _c_u8p]),
    "gme_seq_set_split_phase": (_i, [_vp, _i]),
    "gme_seq_wait": (_i, [_vp]),
    "gme_seq_poll": (_i, [_vp]),
}

_lib = None
_lib_lock = threading.Lock()


def exported_symbols():
    """Names include/gme_hip.h declares (used by the CPU-side symbol test)."""
    return sorted(_SIGNATURES)


def load_library():
    """dlopen libgme_hip.so and attach signatures; never touches a device."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise GmeError(
                "libgme_hip.so is not built (%s). Build it with `make -C %s`; this package has no "
                "CPU fallback." % (LIB_PATH, os.path.join(_HERE, "csrc")))
        # multi-process use (one rank per GPU, RCCL over xGMI): the host driver of this pool only supports dmabuf IPC, and
        # the runtime reads the switch when the first HIP call initialises it -- i.e. after this point.  A value the
        # launcher exported wins.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def _raise(rc, lib):
    msg = (lib.gme_last_error() or b"").decode("utf-8", "replace")
    if rc == ERR_ARG:
        raise IndexError(msg)            # bbme.py:27,60 raise IndexError on bad table indices
    if rc == ERR_GEOMETRY:
        raise AssertionError(msg)        # bbme.py:59
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    raise GmeError("libgme_hip: %s (code %d)" % (msg, rc))


def _check(rc, lib):
    if rc != GME_OK:
        _raise(rc, lib)


def as_frame(a, name="frame"):
    """2-D C-contiguous uint8 view/copy of `a` (the reference feeds grayscale uint8, utils.py:26-28)."""
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("%s must be a 2-D grayscale image, got shape %r" % (name, a.shape))
    if a.dtype != np.uint8:
        raise TypeError("%s must be uint8 (got %s): the device kernels are integer-exact on 8-bit luma" % (name, a.dtype))
    if a.strides[1] != 1 or a.strides[0] < a.shape[1]:
        a = np.ascontiguousarray(a)
    return a


def _p(a, t):
    return a.ctypes.data_as(t)


def _block_size(bs):
    """The reference divides the frame shape by the block size first (bbme.py:23-25, motion.py:303):
    a zero block size is a ZeroDivisionError there, before anything else is looked at."""
    bs = int(bs)
    if bs == 0:
        raise ZeroDivisionError("division by zero")
    return bs


class _Pinned:
    """Owner of one gme_host_alloc block; frees it when the last array view goes away."""

    def __init__(self, lib, nbytes):
        self.lib, self.ptr = lib, lib.gme_host_alloc(nbytes)
        if not self.ptr:
            raise MemoryError((lib.gme_last_error() or b"").decode())

    def __del__(self):
        try:
            if self.ptr:
                self.lib.gme_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.uint8):
    """np.empty in page-locked host memory (gme_host_alloc): frame loaders that decode into it let
    uploads run at link speed and overlap the kernels (Sequence.bbme_streamed, ShardedSequence.load)."""
    lib = load_library()
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    owner = _Pinned(lib, max(n, 1))
    buf = (ctypes.c_uint8 * max(n, 1)).from_address(owner.ptr)
    buf._gme_owner = owner                     # the array's .base chain keeps `buf`, which keeps the block
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class Context:
    """One device + one HIP stream (gme_ctx)."""

    def __init__(self, device=None):
        self.lib = load_library()
        if device is None:
            device = int(os.environ.get("GME_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = self.lib.gme_device_count()
            if n > 0:
                device %= n
        self.device = device
        self._sequences = weakref.WeakSet()
        self.handle = self.lib.gme_create(device)
        if not self.handle:
            raise GmeError("cannot open HIP device %d: %s" % (device, self.lib.gme_last_error().decode()))

    def close(self):
        if getattr(self, "handle", None):
            for seq in list(self._sequences):      # sequences hold device memory of this context
                seq.close()
            self.lib.gme_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- misc
    def sync(self):
        _check(self.lib.gme_sync(self.handle), self.lib)

    def info(self):
        name = ctypes.create_string_buffer(128)
        cu, clk = _i(0), _i(0)
        _check(self.lib.gme_device_info(self.handle, name, 128, ctypes.byref(cu), ctypes.byref(clk)), self.lib)
        return {"name": name.value.decode(), "cu_count": cu.value, "clock_khz": clk.value}

    def last_bbme_info(self):
        """{'plan': kernel / tile shape / schedule of the last block-matching call,
        'patches': candidate patches its elimination bound saw, 'surviving': those scored exactly,
        'listed': those each block's first upper bound left (what a single evaluation round would have scored),
        'redo_tiles': tiles handed to the brute-force redo kernel}."""
        plan = ctypes.create_string_buffer(192)
        n, k, r, l = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        _check(self.lib.gme_last_bbme_info(self.handle, plan, 192, ctypes.byref(n), ctypes.byref(k), ctypes.byref(r)), self.lib)
        _check(self.lib.gme_last_bbme_listed(self.handle, ctypes.byref(l)), self.lib)
        return {"plan": plan.value.decode(), "patches": n.value, "surviving": k.value, "listed": l.value, "redo_tiles": r.value}

    def solve_fit_sums(self, sums, h, w, project=False):
        """gme_solve_fit_sums: float64[P, 15] normal-equation sums -> (params float64[P, 6], flags int32[P])."""
        sums = np.ascontiguousarray(np.asarray(sums, dtype=np.float64).reshape(-1, 15))
        params = np.empty((len(sums), 6), np.float64)
        flags = np.empty(len(sums), np.int32)
        _check(self.lib.gme_solve_fit_sums(self.handle, _p(sums, _c_f64p), len(sums), int(bool(project)), int(h), int(w),
                                           _p(params, _c_f64p), _p(flags, _c_i32p)), self.lib)
        return params, flags

    def timer_start(self):
        _check(self.lib.gme_timer_start(self.handle), self.lib)

    def timer_stop(self):
        ms = ctypes.c_float(0)
        _check(self.lib.gme_timer_stop(self.handle, ctypes.byref(ms)), self.lib)
        return ms.value

    # ---- single-pair calls
    def bbme(self, prev, cur, block_size, search_window, procedure, pnorm):
        prev, cur = as_frame(prev, "previous"), as_frame(cur, "current")
        if prev.shape != cur.shape:
            raise AssertionError("previous and current differ in shape (bbme.py:59)")
        H, W = prev.shape
        block_size = _block_size(block_size)
        if cur.strides[0] != prev.strides[0]:
            cur = np.ascontiguousarray(cur)
            prev = np.ascontiguousarray(prev)
        mf = np.zeros((int(H / block_size), int(W / block_size), 2), dtype=np.int32)
        _check(self.lib.gme_bbme_u8(self.handle, _p(prev, _c_u8p), _p(cur, _c_u8p), H, W, prev.strides[0],
                                    block_size, search_window, procedure, pnorm, _p(mf, _c_i32p)), self.lib)
        return mf

    def pyrdown(self, img):
        img = as_frame(img, "image")
        H, W = img.shape
        out = np.empty(((H + 1) // 2, (W + 1) // 2), dtype=np.uint8)
        _check(self.lib.gme_pyrdown_u8(self.handle, _p(img, _c_u8p), H, W, img.strides[0], _p(out, _c_u8p)), self.lib)
        return out

    def affine_field(self, params, h, w):
        p = np.ascontiguousarray(np.asarray(params).astype(np.float64).reshape(6))
        out = np.zeros((h, w, 2), dtype=np.int16)
        _check(self.lib.gme_affine_field(self.handle, _p(p, _c_f64p), h, w, _p(out, _c_i16p)), self.lib)
        return out

    def compensate(self, frame, mf):
        frame = as_frame(frame)
        mf32 = np.ascontiguousarray(np.asarray(mf)[:, :, :2].astype(np.int32))
        H, W = frame.shape
        out = np.empty((H, W), dtype=np.uint8)
        _check(self.lib.gme_compensate_u8(self.handle, _p(frame, _c_u8p), H, W, frame.strides[0], _p(mf32, _c_i32p),
                                          mf32.shape[0], mf32.shape[1], _p(out, _c_u8p)), self.lib)
        return out

    def sse(self, a, b):
        a, b = as_frame(a, "original"), as_frame(b, "noisy")
        if a.shape != b.shape:
            raise ValueError("operands could not be broadcast together with shapes %r %r" % (a.shape, b.shape))
        v = ctypes.c_int64(0)
        _check(self.lib.gme_sse_u8(self.handle, _p(a, _c_u8p), _p(b, _c_u8p), a.shape[0], a.shape[1], a.strides[0],
                                   b.strides[0], ctypes.byref(v)), self.lib)
        return v.value


class Sequence:
    """N frames resident in HBM (gme_seq) and the per-pair results computed from them."""

    def __init__(self, ctx, n_frames, height, width):
        self.ctx, self.lib = ctx, ctx.lib
        self.N, self.H, self.W = int(n_frames), int(height), int(width)
        self.N_cap = self.N                      # frames the device buffers hold; N = frames in use (set_frames)
        self.handle = self.lib.gme_seq_create(ctx.handle, self.N, self.H, self.W)
        if not self.handle:
            raise GmeError("gme_seq_create failed: %s" % self.lib.gme_last_error().decode())
        ctx._sequences.add(self)
        self._gme = None

    @classmethod
    def from_frames(cls, ctx, frames):
        frames = [as_frame(f) for f in frames] if not isinstance(frames, np.ndarray) else frames
        if isinstance(frames, np.ndarray):
            if frames.ndim != 3 or frames.dtype != np.uint8:
                raise TypeError("frames must be uint8[N, H, W]")
            frames = np.ascontiguousarray(frames)
            seq = cls(ctx, frames.shape[0], frames.shape[1], frames.shape[2])
            seq.upload(0, frames)
            return seq
        seq = cls(ctx, len(frames), *frames[0].shape)
        for k, f in enumerate(frames):
            if f.shape != frames[0].shape:
                raise ValueError("all frames of a sequence must share one shape")
            seq.upload(k, np.ascontiguousarray(f)[None])
        return seq

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.ctx, "handle", None):  # the context may already be gone at interpreter exit
                self.lib.gme_seq_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, first, frames):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, H, W = frames.shape
        if (H, W) != (self.H, self.W):
            raise ValueError("frame shape %r does not match the sequence %r" % ((H, W), (self.H, self.W)))
        _check(self.lib.gme_seq_upload(self.handle, first, n, _p(frames, _c_u8p), W, H * W), self.lib)

    def set_frames(self, n_frames):
        """Use only the first `n_frames` frames from now on (1 <= n_frames <= N_cap): every stage call covers the pairs of
        frames [0, n_frames); the device buffers stay sized for N_cap.  Ends a staged GME run."""
        _check(self.lib.gme_seq_set_frames(self.handle, int(n_frames)), self.lib)
        self.N = int(n_frames)

    def synth(self, seed, t0=0):
        _check(self.lib.gme_seq_synth(self.handle, seed, t0), self.lib)

    def invalidate_pyramids(self):
        """Force the next gme_begin to rebuild pyramid levels 1 and 0 (done automatically after
        upload/synth; bench.py uses it so that every timed step pays for its pyramids)."""
        _check(self.lib.gme_seq_invalidate(self.handle), self.lib)

    def level_shape(self, level):
        h, w = self.H, self.W
        for _ in range(2 - level):
            h, w = (h + 1) // 2, (w + 1) // 2
        return h, w

    def read_frame(self, index, level=2):
        out = np.empty(self.level_shape(level), dtype=np.uint8)
        _check(self.lib.gme_seq_read_frame(self.handle, level, index, _p(out, _c_u8p)), self.lib)
        return out

    # ---- BBME over all pairs
    def bbme(self, frame_distance, block_size, search_window, procedure, pnorm):
        block_size = _block_size(block_size)
        _check(self.lib.gme_seq_bbme(self.handle, frame_distance, block_size, search_window, procedure, pnorm), self.lib)
        self._mv_shape = (self.N - frame_distance, int(self.H / block_size), int(self.W / block_size), 2)

    def bbme_streamed(self, frames, frame_distance, block_size, search_window, procedure, pnorm, chunk_frames=128):
        """bbme() for frames that still live in host memory (uint8[n, H, W], n <= N): chunked upload on a
        copy stream overlapped with the search of the previous chunk -> int32[n - fd, h, w, 2].

        The array returned is a page-locked buffer that belongs to the sequence and is OVERWRITTEN by the next
        bbme_streamed() call on it: copy it if it has to outlive that call ("Host frames, streamed")."""
        block_size = _block_size(block_size)
        frames = np.asarray(frames)
        if frames.ndim != 3 or frames.dtype != np.uint8 or frames.shape[1:] != (self.H, self.W):
            raise TypeError("frames must be uint8[n, %d, %d]" % (self.H, self.W))
        if frames.strides[2] != 1 or frames.strides[1] < self.W or frames.strides[0] < frames.strides[1] * self.H:
            frames = np.ascontiguousarray(frames)
        n = frames.shape[0]
        shape = (max(n - frame_distance, 0), int(self.H / block_size), int(self.W / block_size), 2)
        # page-locked result (the read-backs never hold the calling thread), kept with the sequence: pinning
        # 22 MB takes several milliseconds, more than the search of 2048 pairs.  The array returned is this
        # buffer: copy it if it has to outlive the next bbme_streamed() call on the same sequence.
        out = getattr(self, "_stream_out", None)
        if out is None or out.shape != shape:
            try:
                out = pinned_empty(shape, np.int32) if shape[0] else np.empty(shape, np.int32)
            except MemoryError:
                out = np.empty(shape, dtype=np.int32)
            self._stream_out = out
        _check(self.lib.gme_seq_bbme_streamed(self.handle, _p(frames, _c_u8p), frames.strides[1], frames.strides[0], n,
                                              frame_distance, block_size, search_window, procedure, pnorm, int(chunk_frames),
                                              _p(out, _c_i32p)), self.lib)
        self._mv_shape = out.shape
        return out

    def read_mv(self, first=0, count=None):
        pairs = self._mv_shape[0]
        count = pairs - first if count is None else count
        out = np.empty((count,) + self._mv_shape[1:], dtype=np.int32)
        if out.size:
            _check(self.lib.gme_seq_read_mv(self.handle, first, count, _p(out, _c_i32p)), self.lib)
        return out

    def mv_summary(self):
        """Per-pair summary rows of the last bbme() field -> float64[P, 6] = modal vector x, y (over [-64, 64)^2),
        its block count, sum of x, sum of y, checksum (include/gme_hip.h: gme_seq_mv_summary)."""
        out = np.empty((self._mv_shape[0], 6), dtype=np.float64)
        _check(self.lib.gme_seq_mv_summary(self.handle, _p(out, _c_f64p)), self.lib)
        return out

    def mv_summary_gather(self, n_max, world, slot=0):
        """mv_summary() of every rank, all-gathered device to device over the context's RCCL communicator
        -> float64[world, n_max, 6] (each rank's block zero-padded to n_max rows).  In split-phase mode the array
        is filled once wait() returns; `slot` picks one of several page-locked result buffers so that a caller
        can queue the next step while it still reads the previous one."""
        out = self._buffer("gathered%d" % slot, (int(world), int(n_max), 6), np.float64, by_frames=False)
        _check(self.lib.gme_seq_mv_summary_gather(self.handle, int(n_max), _p(out, _c_f64p)), self.lib)
        return out

    # ---- GME stages
    # -- split-phase calls (gme_seq_set_split_phase): gme_begin / gme_fit / compensate queue their work and hand back
    #    page-locked arrays that are filled once wait() returns; one host thread can then drive several sequences
    def set_split_phase(self, on=True):
        """Split-phase mode: upload / gme_begin / gme_fit / compensate / mv_summary_gather return once queued.  The arrays
        they hand back are page-locked buffers owned by the sequence, one per call kind, REUSED by the next call of that
        kind and valid only after wait(); the blocking helpers (motion.estimate_sequence) refuse a sequence in this mode."""
        _check(self.lib.gme_seq_set_split_phase(self.handle, int(bool(on))), self.lib)
        self._split = bool(on)                   # the page-locked buffers stay with the sequence for the next time

    def wait(self):
        _check(self.lib.gme_seq_wait(self.handle), self.lib)

    def poll(self):
        """True when wait() would return at once (the last split-phase result has arrived)."""
        rc = self.lib.gme_seq_poll(self.handle)
        if rc < 0:
            _raise(rc, self.lib)
        return rc == 1

    def _buffer(self, name, shape, dtype, by_frames=True):
        """Result / argument array of a staged call: ordinary memory, or (split-phase) one page-locked block per
        name and shape kept with the sequence -- the copy engine writes it while the caller is elsewhere."""
        if not getattr(self, "_split", False):
            return np.empty(shape, dtype=dtype)
        pin = self.__dict__.setdefault("_pin", {})
        shape = tuple(shape)
        a = pin.get(name)
        if a is None or a.shape[1:] != shape[1:] or a.shape[0] < shape[0] or a.dtype != np.dtype(dtype):
            # one block per name, sized for the sequence's full frame count: set_frames() changes the rows in use,
            # pinning memory anew for every chunk length would cost more than the chunk
            full = (max(shape[0], self.N_cap) if by_frames else shape[0],) + shape[1:]
            try:
                a = pinned_empty(full, dtype) if int(np.prod(full)) else np.empty(full, dtype=dtype)
            except MemoryError:                  # no page-locked memory left: the copies still work, they just hold the caller
                a = np.empty(full, dtype=dtype)
            pin[name] = a
        return a[:shape[0]]

    def gme_begin(self, frame_distance, bbme_block_size, procedure=3, search_window=2):
        pairs = self.N - frame_distance
        bbme_block_size = _block_size(bbme_block_size)
        p0 = self._buffer("p0", (max(pairs, 0), 6), np.float32)
        _check(self.lib.gme_seq_gme_begin(self.handle, frame_distance, bbme_block_size, procedure, search_window,
                                          _p(p0, _c_f32p)), self.lib)
        self._gme = (frame_distance, bbme_block_size, pairs)
        return p0

    def gme_begin_fit(self, frame_distance, bbme_block_size, outlier_fraction, procedure=3, search_window=2):
        """gme_begin + projection of the first parameters + gme_fit(level 1) without the trip to the host in between
        -> (first parameters float32[P, 6], level-1 sums float64[P, 15]); the level-2 search is queued behind."""
        pairs = self.N - frame_distance
        bbme_block_size = _block_size(bbme_block_size)
        p0 = self._buffer("p0", (max(pairs, 0), 6), np.float32)
        sums = self._buffer("sums1", (max(pairs, 0), 15), np.float64)
        _check(self.lib.gme_seq_gme_begin_fit(self.handle, frame_distance, bbme_block_size, procedure, search_window,
                                              float(outlier_fraction), _p(p0, _c_f32p), _p(sums, _c_f64p)), self.lib)
        self._gme = (frame_distance, bbme_block_size, pairs)
        return p0, sums

    def gme_fit(self, level, params_in, outlier_fraction):
        """-> sums float64[P, 15] = F (9) | Sx (3) | Sy (3).  level -1 fits the field of the
        last bbme() call against the full-resolution frame size."""
        pairs = self._gme[2] if level >= 0 else self._mv_shape[0]
        p = self._buffer("fit_in%d" % level, (pairs, 6), np.float64)
        p[...] = np.asarray(params_in, dtype=np.float64).reshape(pairs, 6)
        sums = self._buffer("sums%d" % level, (pairs, 15), np.float64)
        _check(self.lib.gme_seq_gme_fit(self.handle, level, _p(p, _c_f64p), float(outlier_fraction), _p(sums, _c_f64p)),
               self.lib)
        return sums

    def stage_shape(self, level):
        if level < 0:
            return self._mv_shape[1:3]
        h, w = self.level_shape(level)
        bs = 2 if level == 0 else self._gme[1]
        return int(h / bs), int(w / bs)

    def gme_read_stage(self, level, pair):
        h, w = self.stage_shape(level)
        gt = np.empty((h, w, 2), dtype=np.int32)
        model = np.empty((h, w, 2), dtype=np.int16)
        mask = np.empty((h, w), dtype=np.uint8)
        thr = ctypes.c_int64(0)
        _check(self.lib.gme_seq_gme_read_stage(self.handle, level, pair, _p(gt, _c_i32p), _p(model, _c_i16p),
                                               _p(mask, _c_u8p), ctypes.byref(thr)), self.lib)
        if level == 0:
            return {"gt": gt}
        return {"gt": gt, "model": model, "mask": mask.astype(bool), "thr": thr.value}

    # ---- compensation + squared error
    def compensate(self, frame_distance, block_size, params):
        pairs = self.N - frame_distance
        block_size = _block_size(block_size)
        p = self._buffer("comp_in", (pairs, 6), np.float64)
        p[...] = np.asarray(params, dtype=np.float64).reshape(pairs, 6)
        sse = self._buffer("sse", (pairs,), np.int64)
        _check(self.lib.gme_seq_compensate(self.handle, frame_distance, block_size, _p(p, _c_f64p), _p(sse, _c_i64p)),
               self.lib)
        return sse

    def gme_device_solve(self, frame_distance, bbme_block_size, outlier_fraction, procedure=3, search_window=2):
        """The whole estimate + compensation with the 3x3 solves on the device (gme_seq_gme_device_solve): one round trip
        -> (params float64[P, 6], sse int64[P], flags int32[P]); pairs with a non-zero flag must be redone by the staged calls."""
        pairs = self.N - frame_distance
        bbme_block_size = _block_size(bbme_block_size)
        params = self._buffer("dev_params", (max(pairs, 0), 6), np.float64)
        sse = self._buffer("dev_sse", (max(pairs, 0),), np.int64)
        flags = self._buffer("dev_flags", (max(pairs, 0),), np.int32)
        _check(self.lib.gme_seq_gme_device_solve(self.handle, frame_distance, bbme_block_size, procedure, search_window,
                                                 float(outlier_fraction), _p(params, _c_f64p), _p(sse, _c_i64p), _p(flags, _c_i32p)), self.lib)
        self._gme = (frame_distance, bbme_block_size, pairs)
        return params, sse, flags

    def read_compensated(self, pair):
        out = np.empty((self.H, self.W), dtype=np.uint8)
        _check(self.lib.gme_seq_read_compensated(self.handle, pair, _p(out, _c_u8p)), self.lib)
        return out

    def read_compensated_range(self, first, count, out=None):
        """Compensated frames of pairs first .. first+count-1 -> uint8[count, H, W] with one wait (into `out` if given)."""
        if out is None:
            out = np.empty((count, self.H, self.W), dtype=np.uint8)
        if out.shape != (count, self.H, self.W) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous uint8[%d, %d, %d]" % (count, self.H, self.W))
        _check(self.lib.gme_seq_read_compensated_range(self.handle, int(first), int(count), _p(out, _c_u8p)), self.lib)
        return out


_default = None
_default_lock = threading.Lock()


def default_context():
    """Process-wide context used by the module-level functions of bbme/motion/utils."""
    global _default
    with _default_lock:
        if _default is None:
            _default = Context()
        return _default
