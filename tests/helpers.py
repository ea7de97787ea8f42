"""Shared test plumbing: golden fixtures, the C oracle through ctypes, the NumPy oracle."""
import ctypes
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
ORACLE_SO = os.path.join(REPO, "oracle", "_build", "libgme_oracle.so")


class Golden:
    """Lazy access to tests/golden/*.npz (arrays only, allow_pickle stays False)."""

    def __init__(self):
        self._files = {}

    def __call__(self, name):
        if name not in self._files:
            self._files[name] = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        return self._files[name]


def _ptr(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


class COracle:
    """ctypes face of oracle/gme_oracle.c (built on demand with oracle/Makefile)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])
        self.lib = ctypes.CDLL(ORACLE_SO)
        self.lib.gmeo_sse.restype = ctypes.c_int64

    def bbme(self, prev, cur, bs, sw, proc, pnorm, allow_inexact=0):
        prev = np.ascontiguousarray(prev, np.uint8)
        cur = np.ascontiguousarray(cur, np.uint8)
        H, W = prev.shape
        mf = np.zeros((H // bs, W // bs, 2), np.int32)
        rc = self.lib.gmeo_bbme(_ptr(prev, ctypes.c_uint8), _ptr(cur, ctypes.c_uint8), H, W, bs, sw,
                                proc, pnorm, allow_inexact, _ptr(mf, ctypes.c_int32))
        if rc != 0:
            raise RuntimeError("gmeo_bbme rc=%d" % rc)
        return mf

    def pyrdown(self, src):
        src = np.ascontiguousarray(src, np.uint8)
        H, W = src.shape
        dst = np.empty(((H + 1) // 2, (W + 1) // 2), np.uint8)
        self.lib.gmeo_pyrdown(_ptr(src, ctypes.c_uint8), H, W, _ptr(dst, ctypes.c_uint8))
        return dst

    def affine_field(self, params, h, w):
        p = np.ascontiguousarray(np.asarray(params).astype(np.float64))
        out = np.empty((h, w, 2), np.int16)
        self.lib.gmeo_affine_field(_ptr(p, ctypes.c_double), h, w, _ptr(out, ctypes.c_int16))
        return out

    def fit_level(self, gt, params, frac, level_shape):
        gt = np.ascontiguousarray(gt, np.int32)
        h, w = gt.shape[:2]
        p = np.ascontiguousarray(np.asarray(params).astype(np.float64))
        model = np.empty((h, w, 2), np.int16)
        diff = np.empty((h, w), np.int32)
        thr = ctypes.c_int64(0)
        mask = np.empty((h, w), np.uint8)
        sums = np.empty(15, np.float64)
        rc = self.lib.gmeo_fit_level(_ptr(gt, ctypes.c_int32), h, w, _ptr(p, ctypes.c_double),
                                     ctypes.c_double(frac), int(level_shape[0]), int(level_shape[1]),
                                     _ptr(model, ctypes.c_int16), _ptr(diff, ctypes.c_int32),
                                     ctypes.byref(thr), _ptr(mask, ctypes.c_uint8),
                                     _ptr(sums, ctypes.c_double))
        assert rc == 0
        return dict(model=model, diff=diff, thr=thr.value, mask=mask.astype(bool),
                    F=sums[:9].reshape(3, 3).copy(), Sx=sums[9:12].copy(), Sy=sums[12:15].copy())

    def first_parameters(self, dense):
        dense = np.ascontiguousarray(dense, np.int32)
        out = np.empty(6, np.float32)
        self.lib.gmeo_first_parameters(_ptr(dense, ctypes.c_int32), dense.shape[0] * dense.shape[1],
                                       _ptr(out, ctypes.c_float))
        return out

    def compensate(self, frame, mf):
        frame = np.ascontiguousarray(frame, np.uint8)
        mf = np.ascontiguousarray(mf, np.int32)
        H, W = frame.shape
        out = np.empty_like(frame)
        self.lib.gmeo_compensate(_ptr(frame, ctypes.c_uint8), H, W, _ptr(mf, ctypes.c_int32),
                                 mf.shape[0], mf.shape[1], _ptr(out, ctypes.c_uint8))
        return out

    def sse(self, a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return int(self.lib.gmeo_sse(_ptr(a, ctypes.c_uint8), _ptr(b, ctypes.c_uint8), a.size))


_c = None


def c_oracle():
    global _c
    if _c is None:
        _c = COracle()
    return _c


def np_oracle():
    from oracle import gme_oracle
    return gme_oracle


def oracle_gme(prev, cur, procedure=3, sw=2, bs=16, frac=0.3):
    """motion.global_motion_estimation (motion.py:109-136) through the C oracle with a selectable BBME
    at levels 1-2 (BASELINE configs[3] fits the affine model to an exhaustive-search
    field; the reference itself hard-codes diamond = procedure 3) -> (params, [level-1, level-2 stage])."""
    co, o = c_oracle(), np_oracle()
    pp = [co.pyrdown(co.pyrdown(prev)), co.pyrdown(prev), prev]
    cp = [co.pyrdown(co.pyrdown(cur)), co.pyrdown(cur), cur]
    dense = co.bbme(pp[0], cp[0], 2, 2, 3, 1)
    params = co.first_parameters(dense)
    stages = []
    for lvl in (1, 2):
        params = o.project_parameters(params)
        gt = co.bbme(pp[lvl], cp[lvl], bs, sw, procedure, 1)
        st = co.fit_level(gt, params, frac, pp[lvl].shape)
        st["gt"] = gt
        st["params_in"] = np.array(params, copy=True)
        st["dense"] = dense
        stages.append(st)
        params = o.solve_parameters(st["F"], st["Sx"], st["Sy"])
    return params, stages


def oracle_results_flow(prev, cur, bs=16, procedure=3, sw=2):
    """results.py:50-59,109 for one pair through the C oracle -> (params, field int16, compensated, psnr.real)."""
    co, o = c_oracle(), np_oracle()
    params, _ = oracle_gme(prev, cur, procedure, sw, bs)
    field = co.affine_field(params, int(prev.shape[0] / bs), int(prev.shape[1] / bs))
    comp = co.compensate(prev, field.astype(np.int32))
    return params, field, comp, o.psnr(cur, comp)


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gather_rows_torch(local, n_pairs, rank, world, device=None):
    """The path's one exchange over torch.distributed (gloo in the CPU tests and rehearsals; nccl = RCCL as
    bench.py's collectively agreed fallback transport): all-gather of per-pair rows -> float64[n_pairs, k] on every
    rank, padded to the largest shard like gme_shard_gather.  Test / bench plumbing: the product path uses the
    library's own communicator (sequence.gather_parameters_rccl)."""
    import sequence
    local = np.ascontiguousarray(local, dtype=np.float64)
    if world == 1:
        return local
    import torch
    import torch.distributed as dist
    k = local.shape[1] if local.ndim == 2 else 6
    longest, sizes = sequence.pad_and_trim(n_pairs, world)
    buf = torch.zeros((longest, k), dtype=torch.float64)
    if len(local):
        buf[:len(local)] = torch.from_numpy(local)
    if device is not None:
        buf = buf.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return np.concatenate([o.cpu().numpy()[:b - a] for o, (a, b) in zip(out, sizes)], axis=0)


def mv_summary_rows(mf):
    """NumPy statement of gme_seq_mv_summary (include/gme_hip.h) for int32[P, h, w, 2] fields -> float64[P, 6]:
    modal vector x, y over the vectors inside [-64, 64)^2 (ties: smaller (x+64)*128 + (y+64)), its count,
    sum x, sum y, checksum sum_i ((i mod 251) + 1)(3 x_i + 5 y_i)."""
    mf = np.asarray(mf, dtype=np.int64)
    if mf.ndim == 3:
        mf = mf[None]
    out = np.zeros((mf.shape[0], 6))
    for p, f in enumerate(mf):
        x, y = f[..., 0].ravel(), f[..., 1].ravel()
        inside = (x >= -64) & (x < 64) & (y >= -64) & (y < 64)
        if inside.any():
            counts = np.bincount(((x[inside] + 64) * 128 + (y[inside] + 64)).astype(np.int64), minlength=128 * 128)
            b = int(np.argmax(counts))                       # first maximum = smallest bin
            out[p, :3] = (b // 128 - 64, b % 128 - 64, counts[b])
        wgt = np.arange(len(x), dtype=np.int64) % 251 + 1
        out[p, 3:] = (x.sum(), y.sum(), int((wgt * (3 * x + 5 * y)).sum()))
    return out
