#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X hot path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            # N=1 directly (no torch in the process)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over this rank's batch of frame pairs, frames already resident
in HBM.  Default workload = BASELINE.json configs[1]: 720x480 synthetic luma, bs=16, sw=16, exhaustive
search, MAE.  Frame pairs shard across ranks with no data-path collective ("weak" scaling: every rank
holds its own batch).  The path's one exchange is the all-gather of 48-byte per-pair rows: at N > 1
every step of a block-matching config ends with it (gme_seq_mv_summary_gather: summary kernel ->
ncclAllGather of the device rows -> one copy to the host, queued behind the search so the next step's
search runs behind the collective), and the GME sequence config gathers its parameter rows
(gme_shard_gather).  Barriers and the max-over-ranks time go through the same communicator
(gme_comm_*, include/gme_hip.h).

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      HBM view of the dominant kernel: algorithmic bytes / HIP-event kernel time, measured live
  issue         the same kernel against its real bound, the VALU issue rate: busy fraction and
                instructions per wave from the committed PMC profile of this command (labelled as such)
  parity        the HIP results of >= 64 sampled pairs (first and last included) against the C oracle
  content_sweep the same kernel on other content (real frames, noise, flat): pairs/s and the share of
                candidate patches the elimination bound left for exact evaluation; content_sweep_mse likewise
  secondary     (default command only) the other BASELINE configs at a handful of steps each, so that the one
                driver-run line carries configs[2] and configs[3] too: pairs/s, ms/step, sampled oracle parity
  cpu_baseline  the NumPy oracle (reference loop structure) on one host core over a bounded sample
"""
import argparse
import hashlib
import json
import os
import platform
import sys
import time
import types

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(REPO, "global-motion-estimation_amd"), REPO]

import numpy as np      # noqa: E402

DEFAULT_PAIRS = 2048

CONFIGS = {
    # name: (H, W, bs, sw, procedure, pnorm, seed, label)
    "exh720": (480, 720, 16, 16, 0, 0, 1234, "720x480 synthetic luma, bs=16 sw=16 exhaustive MAE (BASELINE configs[1])"),
    "exh720mse": (480, 720, 16, 16, 0, 1, 1234, "720x480 synthetic luma, bs=16 sw=16 exhaustive MSE"),
    "exh1080": (1080, 1920, 16, 32, 0, 0, 4321, "1920x1080 synthetic luma, bs=16 sw=32 exhaustive MAE"),
    "dia720": (480, 720, 16, 16, 3, 0, 1234, "720x480 synthetic luma, bs=16 diamond MAE"),
    "dia720mse": (480, 720, 16, 16, 3, 1, 1234, "720x480 synthetic luma, bs=16 diamond MSE"),
    "tss720": (480, 720, 16, 16, 1, 1, 1234, "720x480 synthetic luma, bs=16 sw=16 three-step search MSE (bbme.py:182-341)"),
    "tdl720": (480, 720, 16, 16, 2, 1, 1234, "720x480 synthetic luma, bs=16 sw=16 2-D logarithmic search MSE (bbme.py:344-433)"),
    # full GME: procedure/pnorm fields unused (the reference hard-codes diamond + MSE, motion.py:27,224)
    "gme720": (480, 720, 16, 2, -1, 1, 1234, "720x480 full multiscale affine GME (3-level pyramid + diamond BBME + "
               "outlier mask + compensate + PSNR), BASELINE configs[2]"),
    "exh1080mse": (1080, 1920, 16, 32, 0, 1, 4321, "1920x1080 synthetic luma, bs=16 sw=32 exhaustive MSE"),
    "gme1080exh": (1080, 1920, 16, 32, -2, 1, 4321, "1920x1080 synthetic, bs=16 sw=32 exhaustive MSE + affine fit "
                   "(BASELINE configs[3]: GME with exhaustive BBME at levels 1-2) + compensate"),
    # BASELINE configs[4]: the whole 2000-frame 1080p sequence, sharded over the ranks by pair range,
    # diamond GME + compensation per pair, one all-gather of the float64[6] rows at the end of a step
    "seq1080": (1080, 1920, 16, 2, -3, 1, 2000, "2000-frame 1920x1080 synthetic sequence sharded across the GPUs, "
                "diamond-search GME + compensate + PSNR, RCCL all-gather of per-pair parameters (BASELINE configs[4])"),
    "gme1080": (1080, 1920, 16, 2, -1, 1, 2000, "1920x1080 synthetic sequence, diamond-search GME + compensate, "
                "BASELINE configs[4] per-GPU shard"),
    "gme720dev": (480, 720, 16, 2, -1, 1, 1234, "720x480 full multiscale affine GME + compensate + PSNR (BASELINE configs[2]) with the OPT-IN "
                  "device solve (GME_DEVICE_SOLVE=1: 3x3 solves on the device, one host round trip per estimate; parameters within "
                  "rtol 1e-10, everything downstream bit-equal or flagged back to the host path)"),
    # exhaustive MSE on the other unit: sw 16 takes the matrix cores by default (k_exh_mfma16), sw 32 the vector unit's elimination kernel
    "exh720mse_vec": (480, 720, 16, 16, 0, 1, 1234, "720x480 synthetic luma, bs=16 sw=16 exhaustive MSE on the vector unit (GME_EXH_MFMA=0: the "
                      "elimination kernel k_exh_sea16p_mse, the default path until round 4)"),
    "exh1080mse_mfma": (1080, 1920, 16, 32, 0, 1, 4321, "1920x1080 synthetic luma, bs=16 sw=32 exhaustive MSE on the matrix cores (GME_EXH_MFMA=1: "
                        "opt-in at this window, content-independent)"),
    # the block sizes the reference itself runs besides 16
    "tss_bs4sw2": (480, 720, 4, 2, 1, 1, 1234, "720x480 synthetic luma, bs=4 sw=2 three-step search MSE: bbme.get_motion_field's own "
                   "defaults (bbme.py:15-18)"),
    "gme_pan240_bs12fd5": (240, 320, 12, 2, -1, 1, 0, "320x240 real frames (the reference's 51 pan240 frames, walked back and forth), full GME "
                           "+ compensate + PSNR at the slides' setting BBME_BLOCK_SIZE=12, frame distance 5 "
                           "(docs/presentation/main.tex:382; golden g9)"),
}
# per-config extras: frame content uploaded from the host instead of the synthetic generator, frame distance
EXTRA = {"gme_pan240_bs12fd5": {"content": "pan240seq", "fd": 5}, "gme720dev": {"env": {"GME_DEVICE_SOLVE": "1"}, "streams": 2},
         "exh720mse_vec": {"env": {"GME_EXH_MFMA": "0"}}, "exh1080mse_mfma": {"env": {"GME_EXH_MFMA": "1"}}}
INT8_MFMA_PEAK_TOPS = 5000.0     # MI355X_MICROARCH.md: dense bf16 ~2.5 PF, i8 = 2x bf16 per clock (v_mfma_i32_16x16x64_i8: 32768 ops / 16 cycles / SIMD)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
# measured on MI355X (tools/microbench/valu_rates2.hip, profiles/r01_valu_rates.txt):
# v_qsad_pk_u16_u8 issues one wave-instruction (64 lanes x 16 byte-abs-diffs) per ~16.3
# cycles per SIMD at ~2.35 GHz -> 1024 SIMDs * 1024 ops / 6.9 ns
QSAD_PEAK_OPS = 1024 * 1024 / 6.9e-9
PARITY_BUDGET_S = 25.0           # C-oracle time the parity gate may spend per bench line
# rough C-oracle seconds per pair (one core), to size the parity sample
ORACLE_S_PER_PAIR = {"exh720": 0.05, "exh720mse": 0.06, "exh1080": 1.3, "exh1080mse": 1.8, "dia720": 0.01,
                     "dia720mse": 0.01, "tss720": 0.01, "tdl720": 0.01, "gme720": 0.03, "gme1080": 0.15, "seq1080": 0.15,
                     "gme1080exh": 2.1, "tss_bs4sw2": 0.02, "gme_pan240_bs12fd5": 0.01, "gme720dev": 0.03,
                     "exh720mse_vec": 0.06, "exh1080mse_mfma": 1.8}
# the default line's "secondary" block: (config, pairs per step, steps, warmup, wall seconds of C-oracle parity on
# ORACLE_THREADS threads, pairs the parity gate checks at least).  seq1080 = BASELINE configs[4] at N = 1: the whole
# 2000-frame video, with the world-1 RCCL all-gather (gme_shard_gather) inside every step.
SECONDARY = [("gme720", 2048, 20, 3, 1.0, 8), ("gme720dev", 2048, 20, 3, 1.0, 8), ("exh720mse", 2048, 12, 3, 1.0, 8), ("exh720mse_vec", 2048, 12, 3, 1.0, 8),
             ("dia720mse", 2048, 20, 3, 0.5, 8),
             ("tss720", 2048, 20, 3, 0.5, 8), ("tdl720", 2048, 20, 3, 0.5, 8),
             ("tss_bs4sw2", 2048, 20, 3, 0.5, 8), ("gme_pan240_bs12fd5", 2048, 20, 3, 0.5, 8),
             ("exh1080mse", 512, 8, 2, 3.0, 8), ("exh1080mse_mfma", 512, 8, 2, 3.0, 8), ("gme1080exh", 512, 8, 2, 3.5, 8), ("seq1080", None, 8, 2, 2.0, 8)]


def algorithmic_bytes(H, W, bs, gme=False):
    if gme:
        return 3 * H * W + 48                               # 2 frames in, 1 compensated frame out, params
    return 2 * H * W + 8 * (H // bs) * (W // bs)


def byte_ops_per_pair(H, W, bs, sw):
    """Valid candidates x bs^2 (exact count of byte abs-diffs a search that evaluates every candidate needs)."""
    def valid(n):
        tot = 0
        for o in range(0, n - bs + 1, bs):
            tot += sum(1 for w in range(-sw, sw + bs) if 0 <= o + w <= n - bs)
        return tot
    return valid(H) * valid(W) * bs * bs


# Which files of csrc/ a config's launches come from (beyond the ones every config depends on): a change in one file only
# invalidates the committed profiles of the configs listed with it.  bbme_mfma.hip also holds the policy that sends exhaustive
# MSE to one unit or the other, so it belongs to every exhaustive-MSE config.
COMMON_SOURCES = ("Makefile", "gme_internal.h", "gme_api.hip", "bbme_kernels.hip", "synth_kernels.hip")
_SEA = ("bbme_sea_common.h", "bbme_fast.hip")
_GME = ("gme_kernels.hip", "bbme_walk16.hip")
CONFIG_SOURCES = {
    "exh720": ("bbme_sea.hip",) + _SEA, "exh1080": ("bbme_sea.hip",) + _SEA,
    "exh720mse": ("bbme_mfma.hip", "bbme_fast.hip"), "exh1080mse_mfma": ("bbme_mfma.hip", "bbme_fast.hip"),
    "exh720mse_vec": ("bbme_sea_mse.hip", "bbme_mfma.hip") + _SEA, "exh1080mse": ("bbme_sea_mse.hip", "bbme_mfma.hip") + _SEA,
    "gme1080exh": ("bbme_sea_mse.hip", "bbme_mfma.hip") + _SEA + _GME,
    "gme720": _GME, "gme720dev": _GME, "gme1080": _GME, "gme_pan240_bs12fd5": _GME, "seq1080": _GME + ("gme_comm.hip",),
    "tss720": ("bbme_walk16.hip",), "tdl720": ("bbme_walk16.hip",), "dia720": ("bbme_walk16.hip",), "dia720mse": ("bbme_walk16.hip",),
    "tss_bs4sw2": (),
}


def kernel_source_sha(config=None):
    """Identity of the kernels a committed profile belongs to: sha256 over csrc/ sources -- all of them, or (config given)
    the files that config's launches are compiled from (CONFIG_SOURCES + COMMON_SOURCES)."""
    h = hashlib.sha256()
    root = os.path.join(REPO, "global-motion-estimation_amd", "csrc")
    only = None if config is None else set(COMMON_SOURCES) | set(CONFIG_SOURCES[config])
    for name in sorted(os.listdir(root)):
        if (name.endswith((".hip", ".h")) or name == "Makefile") and (only is None or name in only):
            h.update(name.encode())
            h.update(open(os.path.join(root, name), "rb").read())
    return h.hexdigest()[:16]


def profile_is_current(config, vals, psha):
    """A committed profile counts for the tree when the whole of csrc/ hashes to what it recorded, or when the files its own
    config is compiled from do (`# config_source_sha:` line; vals["_config_sha"])."""
    if psha == kernel_source_sha():
        return True
    return bool(config in CONFIG_SOURCES and vals.get("_config_sha") and vals["_config_sha"] == kernel_source_sha(config))


def committed_profile(config, kernel=None):
    """Counters of the newest committed PMC summary of `bench.py --config <config>` (profiles/) for the kernel whose
    name starts with `kernel` (the launch plan's kernel; default: the busiest kernel of the file), its
    kernel_source_sha line and file name; ({}, None, None) if there is none."""
    prof_dir = os.path.join(REPO, "profiles")
    if not os.path.isdir(prof_dir):
        return {}, None, None
    cands = sorted(n for n in os.listdir(prof_dir) if n.endswith("_%s_pmc_summary.txt" % config))
    if not cands:
        return {}, None, None
    per_kernel, sha, csha = {}, None, None
    for line in open(os.path.join(prof_dir, cands[-1])):
        if line.startswith("# kernel_source_sha:"):
            sha = line.split(":", 1)[1].strip()
        if line.startswith("# config_source_sha:"):
            csha = line.split(":", 1)[1].strip()
        f = line.split()
        if not line.startswith("#") and len(f) >= 4 and f[-1].startswith("mean="):
            name = " ".join(f[:-3]).replace(" ", "")           # "k_exh_sea16p<3, 5, 36>" -> "k_exh_sea16p<3,5,36>"
            per_kernel.setdefault(name, {})[f[-3]] = float(f[-1].split("=")[1])
    if not per_kernel:
        return {}, sha, cands[-1]
    if kernel:
        base = kernel.split("<")[0]
        hits = [n for n in per_kernel if n.split("<")[0] == base]
        if not hits:
            return {}, sha, cands[-1]
        name = max(hits, key=lambda n: per_kernel[n].get("GRBM_GUI_ACTIVE", 0.0))
    else:
        name = max(per_kernel, key=lambda n: per_kernel[n].get("GRBM_GUI_ACTIVE", 0.0))
    vals = dict(per_kernel[name])
    vals["_kernel"] = name
    vals["_config_sha"] = csha
    return vals, sha, cands[-1]


def host_description():
    """What the CPU baseline ran on (nproc and CPU model of the box, interpreter, NumPy)."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return {"cpu_model": model, "nproc": os.cpu_count(), "cores_usable_by_this_process": usable,
            "python": platform.python_version(), "numpy": np.__version__, "machine": platform.machine()}


# ---------------------------------------------------------------------------------------------
# content other than the synthetic sequence (exhaustive configs): uint8[n, H, W] host stacks
# ---------------------------------------------------------------------------------------------
def upscale2(f):
    """uint8[n, h, w] -> uint8[n, 2h, 2w]: the frames themselves on the even grid, rounded means in between
    (real content at twice the size without flat 2x2 cells)."""
    f = f.astype(np.uint16)
    n, h, w = f.shape
    up = np.empty((n, 2 * h, 2 * w), np.uint16)
    right = np.concatenate([f[:, :, 1:], f[:, :, -1:]], axis=2)
    down = np.concatenate([f[:, 1:], f[:, -1:]], axis=1)
    diag = np.concatenate([down[:, :, 1:], down[:, :, -1:]], axis=2)
    up[:, 0::2, 0::2] = f
    up[:, 0::2, 1::2] = (f + right + 1) >> 1
    up[:, 1::2, 0::2] = (f + down + 1) >> 1
    up[:, 1::2, 1::2] = (f + right + down + diag + 2) >> 2
    return up.astype(np.uint8)


def host_content(kind, n, H, W):
    g = os.path.join(REPO, "tests", "golden")
    if kind == "noise":            # nothing correlates: the bound prunes (almost) nothing
        rng = np.random.default_rng(99)
        return rng.integers(0, 256, (n, H, W), dtype=np.uint8), H, W
    if kind == "flat":             # all costs tie at zero
        return np.full((n, H, W), 128, np.uint8), H, W
    if kind == "race":             # the reference's 720x480 doc frames: TWO distinct frames, alternating (L2-resident)
        z = np.load(os.path.join(g, "g3_docframes.npz"))
        pair = [z["in_race_prev"], z["in_race_cur"]]
        return np.stack([pair[i & 1] for i in range(n)]), 480, 720
    if kind in ("pan240seq", "pan240x2"):     # the reference's 51 real frames (320x240), walked back and forth
        f = np.load(os.path.join(g, "g9_pan240seq.npz"))["frames"]
        if kind == "pan240x2":                # ... upscaled to 640x480: 51 distinct real frames near the headline size
            f = upscale2(f)
        order = list(range(51)) + list(range(49, 0, -1))
        return np.stack([f[order[i % len(order)]] for i in range(n)]), f.shape[1], f.shape[2]
    raise KeyError(kind)


CONTENT_NOTE = {"race": "2 distinct real frames alternating (L2-resident)", "pan240seq": "51 distinct real frames, 320x240",
                "pan240x2": "51 distinct real frames upscaled x2 to 640x480", "noise": "uniform noise, nothing correlates",
                "flat": "constant frames, every cost ties"}


def sample_pairs(n_pairs, want):
    want = max(2, min(want, n_pairs))
    return sorted(set(int(round(x)) for x in np.linspace(0, n_pairs - 1, want)))


ORACLE_THREADS = max(1, min(8, (os.cpu_count() or 2) - 1))     # the C oracle is re-entrant and ctypes releases the GIL


def parity_sample_size(config, budget_s=PARITY_BUDGET_S, at_least=3):
    """Pairs the parity gate checks: what `budget_s` seconds of wall clock buy with ORACLE_THREADS oracle threads."""
    return int(max(at_least, min(64, budget_s * ORACLE_THREADS / ORACLE_S_PER_PAIR.get(config, 0.1))))


def oracle_map(fn, items):
    """fn over items on ORACLE_THREADS threads (the checker's own parallelism: nothing of the product path runs here)."""
    if ORACLE_THREADS == 1 or len(items) < 2:
        return [fn(x) for x in items]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(ORACLE_THREADS) as ex:
        return list(ex.map(fn, items))


def cpu_baseline_exhaustive(cfg, budget_s=14.0):
    """NumPy oracle on one core over whole block rows of pair (t=0, t=1) until `budget_s`."""
    H, W, bs, sw, proc, pnorm, seed, _ = cfg
    from oracle import gme_oracle
    import synth
    prev, cur = synth.frame(seed, 0, H, W), synth.frame(seed, 1, H, W)
    nbr = H // bs
    mf = np.zeros((nbr, W // bs, 2), np.int32)
    order = [nbr // 2] + [r for r in range(nbr) if r != nbr // 2]      # an interior row first
    done, t0 = [], time.perf_counter()
    for r in order:
        gme_oracle.search_exhaustive(prev, cur, mf, H, W, pnorm, bs, sw, block_rows=(r, r + 1))
        done.append(r)
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0

    def row_cands(r):   # weight rows by their candidate count so that edge rows do not skew the extrapolation
        return sum(1 for w in range(-sw, sw + bs) if 0 <= r * bs + w <= H - bs)
    frac = sum(row_cands(r) for r in done) / sum(row_cands(r) for r in range(nbr))
    return {"value": frac / el, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": "oracle/gme_oracle.py (NumPy, reference loop structure) on %d of %d block rows of pair t=0,1 "
                      "(%.1f%% of the pair's candidates) in %.1f s" % (len(done), nbr, 100 * frac, el),
            "rows_checked": done, "mf": mf}


def cpu_baseline_walk(cfg):
    """NumPy oracle, one core, one whole pair of a walk search (diamond / three-step / 2-D log)."""
    H, W, bs, sw, proc, pnorm, seed, _ = cfg
    from oracle import gme_oracle
    import synth
    prev, cur = synth.frame(seed, 0, H, W), synth.frame(seed, 1, H, W)
    t0 = time.perf_counter()
    mf = gme_oracle.get_motion_field(prev, cur, block_size=bs, search_window=sw, searching_procedure=proc, pnorm_distance=pnorm)
    el = time.perf_counter() - t0
    return {"value": 1.0 / el, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": "oracle/gme_oracle.py (NumPy, reference loop structure) on the whole pair t=0,1 in %.2f s" % el, "mf": mf}


def cpu_baseline_gme(cfg, frames):
    """NumPy oracle (reference loop structure), one core, one pair: motion.global_motion_estimation +
    get_motion_field_affine + compensate_frame + PSNR (results.py:50-59,109).  For the exhaustive-search
    GME of configs[3] only the two level searches are timed, on a bounded sample of block rows."""
    H, W, bs, sw, proc, pnorm, seed, _ = cfg
    from oracle import gme_oracle as o
    prev, cur = frames
    if proc == -2:
        pp, cp = o.get_pyramids(prev), o.get_pyramids(cur)
        total, parts = 0.0, []
        for lvl, budget in ((2, 9.0), (1, 5.0)):
            h, w = pp[lvl].shape
            nbr = h // bs
            mf = np.zeros((nbr, w // bs, 2), np.int32)
            order = [nbr // 2] + [r for r in range(nbr) if r != nbr // 2]
            done, t0 = 0, time.perf_counter()
            for r in order:
                o.search_exhaustive(pp[lvl], cp[lvl], mf, h, w, 1, bs, sw, block_rows=(r, r + 1))
                done += 1
                if time.perf_counter() - t0 > budget:
                    break
            el = time.perf_counter() - t0
            total += el * nbr / done                       # interior rows: an upper-ish estimate of the level
            parts.append("level %d: %d of %d block rows in %.1f s" % (lvl, done, nbr, el))
        return {"value": 1.0 / total, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
                "sample": "oracle/gme_oracle.py exhaustive MSE sw=%d level searches only, extrapolated from %s; pyramids, dense "
                          "field, fits and compensation (seconds) not included" % (sw, "; ".join(parts))}
    t0 = time.perf_counter()
    params = o.global_motion_estimation(prev, cur)
    field = o.affine_field((int(H / bs), int(W / bs)), params)
    comp = o.compensate_frame(prev, field)
    psnr = o.psnr(cur, comp)
    el = time.perf_counter() - t0
    return {"value": 1.0 / el, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": "oracle/gme_oracle.py (NumPy, reference loop structure): GME + model field + compensation + PSNR of "
                      "one pair in %.2f s" % el, "params": params, "psnr": psnr}


class Comm:
    """Barrier / max / gather across the ranks of one node.  world == 1: nothing.  world > 1: the
    library's RCCL communicator (gme_comm_*: ncclAllReduce / ncclAllGather on the context's stream).
    Whether that communicator is used is decided by ALL ranks together (sequence.comm_init raises
    CommUnavailable on every rank if any rank cannot bring it up); only then do all ranks take
    torch.distributed as the transport instead.  GME_BENCH_BACKEND=gloo is the CPU rehearsal."""

    def __init__(self, ctx, rank, world, local):
        self.ctx, self.rank, self.world, self.kind, self.dist, self.torch = ctx, rank, world, "none", None, None
        self.rccl_reports = None
        self.degraded = None                       # set when the agreed fallback transport carries the rows instead of the C ABI's RCCL
        if world == 1 and not os.environ.get("GME_BENCH_FORCE_DIST"):
            return
        backend = os.environ.get("GME_BENCH_BACKEND", "rccl")
        # RCCL may print a version banner on stdout when it initialises; stdout must carry the one JSON
        # line only, so fd 1 is parked on stderr until the communicator exists
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "rccl":
                import sequence
                try:
                    sequence.comm_init(ctx, rank, world)
                    self.kind = "rccl (C ABI: gme_comm_*)"
                    r, n = sequence.comm_info(ctx)
                    self.rccl_reports = {"rank": r, "ranks": n}
                    if (r, n) != (rank, world):
                        raise SystemExit("bench.py: RCCL reports rank %d of %d, the launcher said %d of %d" % (r, n, rank, world))
                    return
                except sequence.CommUnavailable as e:      # raised on EVERY rank: all of them take the same fallback
                    print("bench.py rank %d: C-ABI RCCL communicator unavailable on this launch (%s); all ranks fall back to "
                          "torch.distributed" % (rank, e), file=sys.stderr)
                    backend = "nccl"
                    self.degraded = "torch.distributed fallback (C-ABI RCCL communicator unavailable: %s)" % e
                # anything else (a rendezvous timeout, a HIP error) is not agreed between the ranks: die non-zero
            self._torch_init(backend, rank, world, local)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    def _torch_init(self, backend, rank, world, local):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        dist.barrier()
        self.kind = "torch.distributed/" + backend
        self.device = torch.device("cuda", local) if backend == "nccl" else None

    @property
    def rccl(self):
        return self.kind.startswith("rccl")

    def barrier(self):
        self.ctx.sync()
        if self.rccl:
            import sequence
            sequence.comm_barrier(self.ctx)
        elif self.dist is not None:
            self.dist.barrier()
            if self.torch.cuda.is_available():
                self.torch.cuda.synchronize()

    def min(self, value):
        return -self.max(-value)

    def max(self, value):
        if self.rccl:
            import sequence
            return sequence.comm_max(self.ctx, value)
        if self.dist is not None:
            t = self.torch.tensor([value], dtype=self.torch.float64, device=self.device or "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        return value

    def gather_rows(self, rows, n_pairs_total):
        import sequence
        if self.rccl:
            return sequence.gather_parameters_rccl(self.ctx, rows, n_pairs_total, self.rank, self.world)
        if self.dist is not None:
            sys.path.insert(0, os.path.join(REPO, "tests"))
            from helpers import gather_rows_torch
            return gather_rows_torch(rows, n_pairs_total, self.rank, self.dist.get_world_size(), self.device)
        return rows

    def close(self):
        if self.rccl:
            import sequence
            sequence.comm_destroy(self.ctx)
        elif self.dist is not None:
            self.dist.destroy_process_group()
        self.kind = "none"


def hostile_1080(ctx, n=48):
    """the hostile floor of BASELINE configs[3]'s geometry in the driver's line -- `n` pairs of 1920x1080 uniform
    noise, bs 16, sw 32, both norms: nothing correlates, the elimination kernel hands (almost) every tile to the brute-force
    redo kernel (k_exh_redo16<5, .>).  First and last pair against the C oracle."""
    import _gme_native as native
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import c_oracle
    H, W, bs, sw = 1080, 1920, 16, 32
    fr, _, _ = host_content("noise", n + 1, H, W)
    s2 = native.Sequence.from_frames(ctx, fr)
    co = c_oracle()
    out = {}
    for pn, name in ((0, "mae"), (1, "mse")):
        for _ in range(2):
            s2.invalidate_pyramids()
            s2.bbme(1, bs, sw, 0, pn)
        ctx.sync()
        ctx.timer_start()
        for _ in range(4):
            s2.invalidate_pyramids()
            s2.bbme(1, bs, sw, 0, pn)
        ms = ctx.timer_stop() / 4
        inf = ctx.last_bbme_info()
        chk = [0, n - 1]
        got = [s2.read_mv(p, 1)[0] for p in chk]
        want = oracle_map(lambda p: co.bbme(fr[p], fr[p + 1], bs, sw, 0, pn), chk)
        out[name] = {"frame": "%dx%d" % (W, H), "content": CONTENT_NOTE["noise"], "pairs": n, "pairs_per_s": n / (ms * 1e-3),
                     "kernel": inf["plan"].split(" grid")[0], "tiles_redone_by_brute_force": inf["redo_tiles"],
                     "surviving_fraction": inf["surviving"] / inf["patches"] if inf["patches"] else None,
                     "parity_ok_sampled": bool(all(np.array_equal(g, w) for g, w in zip(got, want))), "pairs_checked_vs_c_oracle": len(chk)}
    s2.close()
    return out


def rank_report(comm, rate_local, gather_s, steps):
    """What an N > 1 line says about its ranks: the slowest and the fastest rank's own pairs/s (collective: every rank
    calls it) and the host-timed cost of the step's exchange on this rank (None where the exchange is queued on the stream)."""
    distributed = comm.kind != "none"
    return {"per_rank_pairs_per_s": {"min": comm.min(rate_local), "max": comm.max(rate_local)} if distributed else None,
            "gather_ms_per_step": 1e3 * gather_s / max(steps, 1) if gather_s is not None else None}


def measure(opt, ctx, comm, rank, world):
    """One bench line: set the workload up, time `opt.steps` steps, check parity; -> dict on rank 0, None elsewhere.
    `opt`: config, content, pairs, steps, warmup, cpu_baseline, content_sweep, pcie, parity_budget_s."""
    import _gme_native as native
    cfg = CONFIGS[opt.config]
    H, W, bs, sw, proc, pnorm, seed, label = cfg
    gme = proc < 0
    extra = EXTRA.get(opt.config, {})
    fd = extra.get("fd", 1)
    B = opt.pairs if opt.pairs is not None else DEFAULT_PAIRS
    distributed = comm.kind != "none"
    env_before = {k: os.environ.get(k) for k in extra.get("env", {})}
    os.environ.update(extra.get("env", {}))
    import motion as _motion
    bs_before = _motion.BBME_BLOCK_SIZE
    if gme:
        _motion.BBME_BLOCK_SIZE = bs               # motion.py:9 is read at call time (the authors set 12 / 24 / 32 for their figures)
    # GME runs cut the resident pairs into `streams` ranges, each on its own HIP stream and host
    # thread: one range's host-side 3x3 solves are covered by the other ranges' kernels
    # (not for the exhaustive-search GME of configs[3]: its kernels run for tens of ms, the host gaps do
    # not matter and concurrent persistent kernels only contend: 17.5 k pairs/s on one stream, 15 k on three)
    # The ranges are driven by ONE host thread through the split-phase calls (ShardedSequence(interleave=True)):
    # gme720 1 stream 499-513 k pairs/s; 2 / 3 / 4 interleaved streams 582 / 588 / 596 k; with a host thread per
    # stream (GME_BENCH_INTERLEAVE=0) 2 / 3 / 4 streams gave 452-471 / 499 / 385-407 k on the same boxes.
    # Round 3, final kernels, same box, four rounds: 2 / 3 / 4 / 6 ranges 619 / 647 / 655 / 634 k (means) -> 4.
    streams = int(os.environ.get("GME_BENCH_STREAMS", str(extra.get("streams", 1 if proc == -2 else 4))))
    interleave = os.environ.get("GME_BENCH_INTERLEAVE", "1") == "1"      # one host thread over all streams (split-phase calls)
    shard = seq = None
    if proc == -3:
        import sequence
        n_frames = int(os.environ.get("GME_BENCH_FRAMES", "2000"))
        shard = sequence.ShardedSequence(H, W, n_frames, 1, rank=rank, world=world, ctx=ctx, streams=streams, interleave=interleave)
        shard.synth(seed)                          # each rank generates its own slice (pairs + 1 halo frame)
        B = shard.n_pairs                          # pairs of THIS rank; the step covers the whole sequence
    elif gme:
        import sequence
        if extra.get("content"):
            frames, H, W = host_content(extra["content"], B + fd, H, W)
            shard = sequence.ShardedSequence(H, W, B + fd, fd, ctx=ctx, streams=streams, interleave=interleave)
            shard.load(frames)
            del frames
        else:
            shard = sequence.ShardedSequence(H, W, B + 1, 1, ctx=ctx, streams=streams, interleave=interleave)
            shard.synth(seed, rank * B)            # rank r holds frames t = r*B .. r*B+B (halo of fd=1 included)
    elif opt.content == "synthetic":
        seq = native.Sequence(ctx, B + 1, H, W)
        seq.synth(seed, rank * B)
    else:
        frames, H, W = host_content(opt.content, B + 1, H, W)
        label = label.replace("synthetic luma", "%s content" % opt.content).replace("720x480", "%dx%d" % (W, H))
        seq = native.Sequence.from_frames(ctx, frames)
        del frames
    if shard is not None:
        shard.sync()
    ctx.sync()

    last = {}
    finish = None
    if proc == -3:
        def step():
            shard.invalidate()
            params, psnr = shard.estimate_and_compensate()
            rows = np.concatenate([params, psnr[:, None]], axis=1)
            last["local_rows"] = rows
            t_g = time.perf_counter()
            last["rows"] = comm.gather_rows(rows, shard.n_pairs_total)     # the path's one exchange: 56 B per pair
            last["gather_s"] = last.get("gather_s", 0.0) + time.perf_counter() - t_g
    elif gme:
        def step():
            # motion.global_motion_estimation + results.py:52-59,109 for every resident pair;
            # frames change between videos, so the pyramids are rebuilt inside the step
            shard.invalidate()
            if proc == -2:
                last["params"], last["psnr"] = shard.estimate_and_compensate(0, sw)
            else:
                last["params"], last["psnr"] = shard.estimate_and_compensate()
    elif distributed and comm.rccl:
        # N > 1: every step ends with the path's one exchange -- the all-gather of one 48-byte summary row per pair
        # (modal vector, its count, vector sums, checksum), device to device over RCCL on the search's own stream.  It
        # is queued behind the search and awaited one step later, so the next search runs behind the collective.
        seq.set_split_phase(True)
        pend = {"k": 0, "buf": None}

        def step():
            seq.invalidate_pyramids()
            seq.bbme(1, bs, sw, proc, pnorm)
            if pend["buf"] is not None:
                seq.wait()                         # rows of the previous step (the event behind its copy)
                last["gathered"] = pend["buf"]
            pend["buf"] = seq.mv_summary_gather(B, world, slot=pend["k"] & 1)
            pend["k"] += 1

        def finish():
            if pend["buf"] is not None:
                seq.wait()
                last["gathered"] = pend["buf"]
                pend["buf"] = None
    elif distributed:
        def step():                                # the agreed fallback transport / the gloo rehearsal: rows via the host
            seq.invalidate_pyramids()
            seq.bbme(1, bs, sw, proc, pnorm)
            local = seq.mv_summary()
            t_g = time.perf_counter()
            rows = comm.gather_rows(local, world * B)
            last["gather_s"] = last.get("gather_s", 0.0) + time.perf_counter() - t_g
            last["gathered"] = rows.reshape(world, B, 6)
    else:
        def step():
            # per-frame auxiliary tables (the MSE identity's box sums of squares) are derived from the
            # frames: a new batch has to rebuild them, so every step does
            seq.invalidate_pyramids()
            seq.bbme(1, bs, sw, proc, pnorm)       # asynchronous launches on the context's stream

    for _ in range(opt.warmup):
        step()
    if finish:
        finish()
    if shard is not None:
        shard.sync()
    last.pop("gather_s", None)
    comm.barrier()
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        step()
    if finish:
        finish()
    kernel_ms = ctx.timer_stop() / max(opt.steps, 1)      # HIP events on the launch stream; synchronises
    if shard is not None:
        shard.sync()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    if gme:
        kernel_ms = 1e3 * elapsed / max(opt.steps, 1)     # several streams: the whole step on the wall clock
    # N > 1: the slowest and the fastest rank (their own pairs over their own clock between the two barriers), and what the
    # step's exchange costs: host-timed around the gather where it blocks; where it is queued behind the search (RCCL,
    # block matching) five gathers are timed alone after the loop
    rep = rank_report(comm, B * opt.steps / elapsed, last.get("gather_s"), opt.steps)
    per_rank, gather_ms = rep["per_rank_pairs_per_s"], rep["gather_ms_per_step"]
    if seq is not None and distributed and comm.rccl:
        comm.barrier()
        t_g = time.perf_counter()
        for k in range(5):
            seq.mv_summary_gather(B, world, slot=k & 1)
            seq.wait()
        gather_ms = 1e3 * (time.perf_counter() - t_g) / 5
    elapsed = comm.max(elapsed)
    if seq is not None and distributed and comm.rccl:
        seq.set_split_phase(False)

    def release():
        _motion.BBME_BLOCK_SIZE = bs_before
        for k, v in env_before.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        if seq is not None:
            seq.close()
        if shard is not None:
            shard.close()

    if rank != 0:
        release()
        return None
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import c_oracle, mv_summary_rows, oracle_results_flow

    # ---- parity gate printed with the number: sampled pairs (first and last included) against the C oracle
    n_s = parity_sample_size(opt.config, opt.parity_budget_s, getattr(opt, "parity_min_pairs", 3))
    parity = {"checker": "oracle/gme_oracle.c (C restatement pinned on the reference's goldens)"}
    t_par = time.perf_counter()
    if not gme:
        info = ctx.last_bbme_info()
        co = c_oracle()
        idx = sample_pairs(B, n_s)
        got = [(seq.read_mv(p, 1)[0], seq.read_frame(p), seq.read_frame(p + 1)) for p in idx]       # device reads: one thread
        want = oracle_map(lambda t: co.bbme(t[1], t[2], bs, sw, proc, pnorm), got)
        bad = [p for p, g, w in zip(idx, got, want) if not np.array_equal(g[0], w)]
        parity.update({"pairs_checked": len(idx), "first": idx[0], "last": idx[-1], "mismatching_pairs": bad, "ok": not bad,
                       "what": "motion-vector field, bit-exact"})
        gpath = os.path.join(REPO, "tests", "golden", "g2_synth720.npz")
        if (opt.config in ("exh720", "exh720mse", "dia720", "dia720mse", "tss720", "tdl720") and opt.content == "synthetic"
                and os.path.exists(gpath)):
            key = "mf_sp%d_pn%d" % (proc, pnorm)
            g2 = np.load(gpath)
            if key in g2.files:
                parity["pair0_equals_reference_golden"] = bool(np.array_equal(seq.read_mv(0, 1)[0], g2[key]))
                parity["ok"] = parity["ok"] and parity["pair0_equals_reference_golden"]
        if distributed:
            # the gathered rows: rank 0's own block against its own fields, and sampled pairs OWNED BY OTHER RANKS
            # against the C oracle on frames regenerated by the host generator (rank r holds t = r*B .. r*B+B)
            import synth
            g = np.asarray(last["gathered"])
            own_ok = bool(np.array_equal(g[0], mv_summary_rows(seq.read_mv(0, B))))
            others, bad_rows = [], []
            if opt.content == "synthetic":
                for r in range(1, world):
                    for j in sorted({0, B // 2, B - 1}):
                        want = mv_summary_rows(co.bbme(synth.frame(seed, r * B + j, H, W), synth.frame(seed, r * B + j + 1, H, W),
                                                       bs, sw, proc, pnorm))[0]
                        others.append([r, j])
                        if not np.array_equal(g[r, j], want):
                            bad_rows.append([r, j])
            parity["gathered_rows"] = {"shape": list(g.shape), "own_block_equals_own_fields": own_ok,
                                       "pairs_of_other_ranks_checked_vs_oracle": others, "mismatching": bad_rows,
                                       "row": "modal vector x, y, its block count, sum x, sum y, checksum (gme_seq_mv_summary)"}
            parity["ok"] = parity["ok"] and own_ok and not bad_rows
    else:
        info = shard.lanes[0].ctx.last_bbme_info()
        n_local = shard.n_pairs
        idx = sample_pairs(n_local, n_s)
        params = last["local_rows"][:, :6] if proc == -3 else last["params"]
        psnr = last["local_rows"][:, 6] if proc == -3 else last["psnr"]
        bad = []
        worst = 0.0
        inputs = []
        for p in idx:                                   # device reads: one thread
            lane, k = shard._lane_of(p)
            inputs.append((lane.seq.read_frame(k), lane.seq.read_frame(k + fd), shard.read_compensated(p)))
        wants = oracle_map(lambda t: oracle_results_flow(t[0], t[1], bs, 0 if proc == -2 else 3, sw if proc == -2 else 2), inputs)
        for p, t, (wp, _, wc, wpsnr) in zip(idx, inputs, wants):
            ok = (np.allclose(params[p], wp, rtol=1e-10, atol=1e-12) and np.array_equal(t[2], wc)
                  and abs(psnr[p] - wpsnr) < 1e-9)
            worst = max(worst, float(np.max(np.abs(params[p] - wp))))
            if not ok:
                bad.append(p)
        parity.update({"pairs_checked": len(idx), "first": idx[0], "last": idx[-1], "mismatching_pairs": bad, "ok": not bad,
                       "what": "parameters (rtol 1e-10), compensated frame (bit-exact), PSNR (1e-9 dB)",
                       "max_abs_param_diff": worst})
        if proc == -3 and world > 1:            # gathered rows of the other ranks: the frames come from the host generator
            import synth
            rows = last["rows"]
            other = [p for p in sample_pairs(shard.n_pairs_total, 5) if not (shard.pair_start <= p < shard.pair_stop)]
            for p in other:
                wp, _, _, wpsnr = oracle_results_flow(synth.frame(seed, p, H, W), synth.frame(seed, p + 1, H, W), bs)
                if not (np.allclose(rows[p, :6], wp, rtol=1e-10, atol=1e-12) and abs(rows[p, 6] - wpsnr) < 1e-9):
                    bad.append(p)
            parity.update({"gathered_pairs_of_other_ranks_checked": other, "mismatching_pairs": bad, "ok": not bad})
        if opt.config == "gme720":
            g4 = np.load(os.path.join(REPO, "tests", "golden", "g4_gme.npz"))
            comp_sha = hashlib.sha256(shard.read_compensated(0).tobytes()).hexdigest()
            parity["pair0_equals_reference_golden"] = bool(np.allclose(last["params"][0], g4["synth720_params"], rtol=1e-10, atol=1e-12)
                                                           and comp_sha == str(g4["synth720_comp_sha"]))
            parity["ok"] = parity["ok"] and parity["pair0_equals_reference_golden"]
    parity["seconds"] = round(time.perf_counter() - t_par, 2)

    total_pairs = (shard.n_pairs_total if proc == -3 else world * B) * opt.steps
    value = total_pairs / elapsed
    abytes = algorithmic_bytes(H, W, bs, gme) * B
    achieved = abytes / (kernel_ms * 1e-3) / 1e9
    if proc == -3:
        exchange = "all-gather of float64[7] rows (6 parameters + PSNR) per pair at the end of a step (gme_shard_gather)"
    elif distributed and not gme:
        exchange = ("all-gather of one float64[6] summary row per pair at the end of every step, device to device, queued behind "
                    "the search (gme_seq_mv_summary_gather)" if comm.rccl else
                    "all-gather of one float64[6] summary row per pair at the end of every step, through the host (gme_seq_mv_summary)")
    else:
        exchange = "none in this run"
    out = {
        "metric": "frame-pairs/s + achieved HBM GB/s, 720x480 bs=16 sw=16 exhaustive"
                  if opt.config == "exh720" else "frame-pairs/s, " + opt.config,
        "value": value, "unit": "frame-pairs/s", "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup,
        "ms_per_step": 1e3 * elapsed / opt.steps, "higher_is_better": True,
        "scaling": "strong" if proc == -3 else "weak",
        "vs_baseline": None, "dtype": "u8", "data": extra.get("content") or ("synthetic" if opt.content == "synthetic" else opt.content),
        "config": {"workload": label, "pairs_per_step_per_gpu": B, "frame_distance": fd,
                   "sharding": "frame pairs across ranks, no data-path collective",
                   "streams_per_gpu": len(shard.lanes) if shard is not None else 1,
                   "collective": comm.kind, "exchange": exchange, "rccl_reports": comm.rccl_reports,
                   "per_rank_pairs_per_s": per_rank, "gather_ms_per_step": gather_ms,
                   "multi_gpu_measured_by_builder": False},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": info["plan"] if not gme else "whole step (all kernels + host solves)",
                     "kernel_ms_per_launch": kernel_ms, "algorithmic_bytes_per_launch": abytes},
        "parity": parity,
    }
    if comm.degraded:
        out["degraded"] = comm.degraded             # `value` then is NOT the product path's exchange; bench.py exits non-zero
    if distributed and not gme:
        out["roofline"]["note"] = "kernel_ms_per_launch spans the step's search, summary kernel and all-gather (HIP events on the stream)"
    if gme:
        # the dominant kernel of the GME step is the level-2 block search (diamond: k_walk16<1>; configs[3]: the
        # exhaustive MSE kernel): time that launch alone with HIP events on one lane's stream and rate it against the
        # bytes it must touch (both frames once + the field)
        lane0 = shard.lanes[0]
        n0 = lane0.hi - lane0.lo
        p_proc, p_sw = (0, sw) if proc == -2 else (3, 2)
        lane0.seq.bbme(fd, bs, p_sw, p_proc, 1)
        lane0.ctx.sync()
        lane0.ctx.timer_start()
        for _ in range(3):
            lane0.seq.bbme(fd, bs, p_sw, p_proc, 1)
        k_ms = lane0.ctx.timer_stop() / 3
        kinfo = lane0.ctx.last_bbme_info()
        kbytes = algorithmic_bytes(H, W, bs, False) * n0
        out["roofline"].update({
            "kernel": kinfo["plan"] + " (level-2 search of the step, timed alone on %d pairs)" % n0,
            "kernel_ms_per_launch": k_ms, "algorithmic_bytes_per_launch": kbytes,
            "achieved": kbytes / (k_ms * 1e-3) / 1e9, "frac": kbytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "whole_step": {"ms": kernel_ms, "algorithmic_bytes": abytes, "achieved_GBps": achieved, "frac": achieved / HBM_PEAK_GBS,
                           "note": "all kernels + host solves, %d streams" % len(shard.lanes)}})
        info = kinfo
    if info.get("patches"):
        out["elimination"] = {"patches": info["patches"], "surviving": info["surviving"],
                              "surviving_fraction": info["surviving"] / info["patches"],
                              "listed_fraction_before_ordered_rounds": info["listed"] / info["patches"],
                              "tiles_redone_by_brute_force": info["redo_tiles"]}

    # ---- HBM traffic and issue statistics of the dominant kernel from the committed rocprofv3 PMC passes of
    # this same command (profiles/): valid only for the kernels they were taken from (kernel_source_sha)
    plan_kernel = info["plan"].split(" ")[0] if info.get("plan") else None
    vals, psha, pname = committed_profile(opt.config, plan_kernel)
    switches = sorted(k for k in os.environ if k.startswith("GME_") and k not in ("GME_DEVICE",) and k not in extra.get("env", {}))
    if proc == 0:
        ops = byte_ops_per_pair(H, W, bs, sw) * B / (kernel_ms * 1e-3)
        out["brute_force_equivalent"] = {
            "note": "byte abs-diffs a search that evaluates EVERY candidate would need, per second; the elimination kernel "
                    "never evaluates most of them, so this is not a utilisation figure (it may exceed the QSAD issue peak)",
            "value": ops, "unit": "byte-abs-diff/s", "qsad_issue_peak": QSAD_PEAK_OPS}
    if vals and opt.content == "synthetic":
        reason = None
        if not profile_is_current(opt.config, vals, psha):
            reason = "kernel sources changed since %s was taken" % pname
        elif switches:
            reason = "runtime switches set: " + ",".join(switches)
        elif (B != DEFAULT_PAIRS and proc != -3) or world != 1:      # seq1080: its default is the 2000-frame video (a GME_BENCH_FRAMES
            reason = "profile is of the default single-GPU command"       # override counts as a runtime switch above)
        if reason is None and "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            out["roofline"]["traffic"] = int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
            out["roofline"]["traffic_source"] = ("committed profile profiles/%s, kernel %s: (2 x FETCH_SIZE + WRITE_SIZE) KB per launch "
                                                 "(mean over that kernel's launches of this command%s), x2 = the guide's gfx950 FETCH_SIZE "
                                                 "correction" % (pname, vals["_kernel"], "; [grid=N]: the launches of that grid size only, "
                                                 "i.e. the level-2 search of one pair range" if "[grid=" in vals["_kernel"] else ""))
        else:
            out["roofline"]["traffic_source"] = "null: " + (reason or "no FETCH_SIZE/WRITE_SIZE in the committed profile")
        if reason is None and all(k in vals for k in ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
                                                      "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")):
            cyc = vals["GRBM_GUI_ACTIVE"] / 8.0                 # summed over the 8 XCDs
            wave_tiles = info["patches"] / (64 * ((2 * sw + 16 + 15) // 16)) if info.get("patches") else None
            out["issue"] = {
                "bound": "valu_issue", "source": "committed profile profiles/%s (same kernel sources, same command)" % pname,
                "kernel": vals["_kernel"],
                "achieved": vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc), "peak": 1.0, "unit": "VALU-busy fraction of SIMD cycles",
                "frac": vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc),
                "salu_per_valu_inst": vals["SQ_INSTS_SALU"] / vals["SQ_INSTS_VALU"],
                "lds_conflict_share": vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"] if vals["SQ_LDS_IDX_ACTIVE"] else None,
                "insts_per_wave_tile": None if not wave_tiles else {"valu": vals["SQ_INSTS_VALU"] / wave_tiles,
                                                                    "salu": vals["SQ_INSTS_SALU"] / wave_tiles,
                                                                    "lds": vals["SQ_INSTS_LDS"] / wave_tiles}}
    elif opt.content == "synthetic":
        out["roofline"]["traffic_source"] = "null: no committed PMC profile of this config for kernel %s" % plan_kernel

    # ---- the same kernel on other content: real frames, noise (no pruning), flat (all ties); and the MSE kernel
    if proc == 0 and world == 1 and opt.content == "synthetic" and opt.content_sweep:
        def sweep_of(pn, kinds):
            sweep = {}
            nsw = B if H * W <= 720 * 480 else min(B, 512)      # the headline's own batch (rounds 2-3 swept 512 pairs: 6-7 % lower rates)
            for kind in kinds:
                if kind in ("race", "pan240x2") and (H, W) != (480, 720):
                    continue
                fr, h2, w2 = host_content(kind, nsw + 1, H, W)
                s2 = native.Sequence.from_frames(ctx, fr)
                for _ in range(4):                              # a few untimed passes: first touches, clocks back up after the
                    s2.invalidate_pyramids()                    # host-side checks of the previous entry
                    s2.bbme(1, bs, sw, proc, pn)
                ctx.sync()
                ctx.timer_start()
                for _ in range(8):
                    s2.invalidate_pyramids()
                    s2.bbme(1, bs, sw, proc, pn)
                ms = ctx.timer_stop() / 8
                inf = ctx.last_bbme_info()
                co = c_oracle()
                chk = sample_pairs(nsw, 6)
                ok = all(np.array_equal(s2.read_mv(p, 1)[0], co.bbme(fr[p], fr[p + 1], bs, sw, proc, pn)) for p in chk)
                sweep[kind] = {"frame": "%dx%d" % (w2, h2), "content": CONTENT_NOTE[kind], "pairs": nsw,
                               "pairs_per_s": nsw / (ms * 1e-3), "kernel": inf["plan"].split(" grid")[0],
                               "surviving_fraction": inf["surviving"] / inf["patches"] if inf["patches"] else None,
                               "listed_fraction_before_ordered_rounds": inf["listed"] / inf["patches"] if inf["patches"] else None,
                               "tiles_redone_by_brute_force": inf["redo_tiles"],
                               "parity_ok_sampled": bool(ok)}
                s2.close()
            return sweep
        # real content = the reference's 51 distinct pan240 frames (320x240, and upscaled x2 to 640x480); the two-frame `race`
        # pair of rounds 2-3 stays available as --content race
        out["content_sweep"] = sweep_of(pnorm, ("pan240x2", "pan240seq", "noise", "flat"))
        if opt.config == "exh720":
            out["content_sweep_mse"] = sweep_of(1, ("pan240x2", "noise"))
            out["content_sweep_1080p_noise"] = hostile_1080(ctx)

    if proc >= 0 and world == 1 and opt.pcie:
        # host-buffer (PCIe-inclusive) rate, NOT `value`: frames cross to the device, the fields
        # come back (SURVEY.md §8(d) "end-to-end number including H2D/D2H")
        n_e2e = B
        chunk = int(os.environ.get("GME_BENCH_CHUNK", "128"))
        host_frames = native.pinned_empty((n_e2e + 1, H, W))  # what a frame loader that decodes into page-locked memory hands over
        for i in range(n_e2e + 1):
            host_frames[i] = seq.read_frame(i)
        seq2 = native.Sequence(ctx, n_e2e + 1, H, W)          # its own sequence: upload, search, read back
        seq2.bbme_streamed(host_frames, 1, bs, sw, proc, pnorm, chunk)     # first touch (staging buffer, page-locked result) outside the timing
        ctx.sync()
        t_e = time.perf_counter()
        mv2 = seq2.bbme_streamed(host_frames, 1, bs, sw, proc, pnorm, chunk)
        t_e = time.perf_counter() - t_e
        mv2 = mv2.copy()                                       # the next call reuses the page-locked result buffer
        same = bool(np.array_equal(mv2[-1], seq.read_mv(B - 1, 1)[0]) and np.array_equal(mv2[0], seq.read_mv(0, 1)[0]))
        pageable = np.array(host_frames)                       # an ordinary NumPy stack: the HIP runtime stages it
        t_p = time.perf_counter()
        mv3 = seq2.bbme_streamed(pageable, 1, bs, sw, proc, pnorm, chunk)
        t_p = time.perf_counter() - t_p
        same = same and bool(np.array_equal(mv3, mv2))
        ctx.sync()
        t_c = time.perf_counter()                              # the copy alone (same bytes, page-locked source): the ceiling
        seq2.upload(0, host_frames)
        t_c = time.perf_counter() - t_c
        seq2.close()
        gbs = (n_e2e + 1) * H * W / t_e / 1e9
        out["pcie_inclusive"] = {"value": n_e2e / t_e, "unit": "frame-pairs/s", "equals_resident_result": same,
                                 "host_to_device_GBps": gbs, "chunk_frames": chunk,
                                 "copy_only": {"GBps": (n_e2e + 1) * H * W / t_c / 1e9, "pairs_per_s_if_nothing_else": n_e2e / t_c},
                                 "from_pageable_memory": n_e2e / t_p,
                                 "note": "gme_seq_bbme_streamed: %d frames from page-locked host memory uploaded in chunks on a copy stream "
                                         "while the previous chunk is searched, + read-back of %d fields; each frame crosses the link once "
                                         "(from_pageable_memory: the same call on an ordinary NumPy array)" % (n_e2e + 1, n_e2e)}
    if proc == -1 and world == 1 and opt.pcie:
        # the same flow from frames in HOST memory (results.py:41-59 hands the path host arrays): sequence.StreamEstimator
        # uploads chunk k + 1 while chunk k is estimated and compensated; NOT `value`
        import sequence
        chunk = int(os.environ.get("GME_BENCH_CHUNK", "512"))
        lanes_e = int(os.environ.get("GME_BENCH_STREAM_LANES", "2"))
        host_frames = native.pinned_empty((B + 1, H, W))
        for lane in shard.lanes:
            for k in range(lane.hi - lane.lo + 1):
                host_frames[lane.lo + k] = lane.seq.read_frame(k)
        with sequence.StreamEstimator(H, W, 1, chunk, lanes_e, ctx=ctx) as est:
            est.run(host_frames, exact_psnr=False)                       # first touch
            t_e = time.perf_counter()
            p_e, psnr_e = est.run(host_frames, exact_psnr=False)
            t_e = time.perf_counter() - t_e
        seq2 = native.Sequence(ctx, B + 1, H, W)
        seq2.upload(0, host_frames)
        t_c = time.perf_counter()                                        # the copy alone (same bytes, page-locked source): the ceiling
        seq2.upload(0, host_frames)
        t_c = time.perf_counter() - t_c
        seq2.close()
        out["pcie_inclusive"] = {"value": B / t_e, "unit": "frame-pairs/s", "chunk_pairs": chunk, "lanes": lanes_e,
                                 "equals_resident_result": bool(np.array_equal(p_e, last["params"]) and np.array_equal(psnr_e, last["psnr"])),
                                 "host_to_device_GBps": (B + 1) * H * W / t_e / 1e9,
                                 "copy_only": {"GBps": (B + 1) * H * W / t_c / 1e9, "pairs_per_s_if_nothing_else": B / t_c},
                                 "fraction_of_copy_ceiling": t_c / t_e,
                                 "note": "sequence.StreamEstimator: %d frames from page-locked host memory in chunks of at most %d pairs (shrinking towards the end) over %d lanes "
                                         "(split-phase uploads on one shared upload stream, estimate + compensation + PSNR per chunk); "
                                         "parameters and PSNR read back, compensated frames stay on the device" % (B + 1, chunk, lanes_e)}
    if proc == -3:
        rows = last["rows"]
        out["sequence"] = {"pairs_total": int(shard.n_pairs_total), "gathered_rows": int(rows.shape[0]),
                           "mean_psnr_db": float(np.mean(rows[:, 6])),
                           "median_params": [float(x) for x in np.median(rows[:, :6], axis=0)]}
    if world == 1 and opt.cpu_baseline and opt.content == "synthetic":
        if proc == 0:
            cb = cpu_baseline_exhaustive(cfg)
            rows_c, ref_mf = cb.pop("rows_checked"), cb.pop("mf")
            mv0 = seq.read_mv(0, 1)[0]
            cb["matches_gpu"] = bool(all(np.array_equal(ref_mf[r], mv0[r]) for r in rows_c))
        elif proc > 0:
            cb = cpu_baseline_walk(cfg)
            cb["matches_gpu"] = bool(np.array_equal(cb.pop("mf"), seq.read_mv(0, 1)[0]))
        else:
            lane = shard.lanes[0]
            cb = cpu_baseline_gme(cfg, (lane.seq.read_frame(0), lane.seq.read_frame(1)))
            if "params" in cb:
                p0 = last["local_rows"][0, :6] if proc == -3 else last["params"][0]
                ps0 = last["local_rows"][0, 6] if proc == -3 else last["psnr"][0]
                cb["matches_gpu"] = bool(np.allclose(cb.pop("params"), p0, rtol=1e-10, atol=1e-12) and abs(cb.pop("psnr") - ps0) < 1e-9)
        cb["host"] = host_description()
        out["cpu_baseline"] = cb
        if proc == 0:
            # BASELINE.md §3 "optimised CPU": the integer C oracle (gcc -O3, one core) on the same
            # pair, so the GPU figure is not only measured against interpreter overhead
            co = c_oracle()
            p0, p1 = seq.read_frame(0), seq.read_frame(1)
            t_c = time.perf_counter()
            mf_c = co.bbme(p0, p1, bs, sw, proc, pnorm)
            t_c = time.perf_counter() - t_c
            out["cpu_baseline_c"] = {"value": 1.0 / t_c, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
                                     "sample": "oracle/gme_oracle.c (integer C, gcc -O3) on the whole pair t=0,1 in %.2f s" % t_c,
                                     "matches_gpu": bool(np.array_equal(mf_c, seq.read_mv(0, 1)[0]))}
    if not gme and info.get("plan", "").startswith("k_exh_mfma16"):
        # the matrix-core search: the roofline that binds is the int8 MFMA rate, not HBM.  ALGORITHMIC ops = 2 x 256 byte
        # products per valid candidate (the kernel issues about twice that: Toeplitz operands carry zeros, tiles are padded)
        NT = (2 * sw + 16) // 16
        hbm = dict(out["roofline"])
        ops = 2.0 * byte_ops_per_pair(H, W, bs, sw) * B
        issued = 2.0 * 16 * 16 * 64 * (NT * NT * 8) * (H // 16) * (W // 16) * B
        out["roofline"] = {"bound": "mfma", "achieved": ops / (kernel_ms * 1e-3) / 1e12, "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                           "frac": ops / (kernel_ms * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS, "traffic": hbm.get("traffic"),
                           "traffic_source": hbm.get("traffic_source"), "kernel": hbm["kernel"], "kernel_ms_per_launch": kernel_ms,
                           "algorithmic_ops_per_launch": ops, "issued_mfma_ops_per_launch": issued,
                           "issued_frac": issued / (kernel_ms * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS,
                           "note": "kernel_ms_per_launch spans the step's two kernels (k_sqbox16 table + k_exh_mfma16); v_mfma_i32_16x16x64_i8, exact int32 sums",
                           "hbm": {k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")}}
    release()
    return out


def secondary_block(ctx, comm):
    """The other BASELINE configs at a handful of steps each (same code path as `--config <name>`, fewer steps, a
    smaller parity sample, no CPU baseline): the driver's one default run then carries configs[2] / configs[3] numbers."""
    block = {}
    t0 = time.perf_counter()
    for name, pairs, steps, warmup, par_s, par_min in SECONDARY:
        opt = types.SimpleNamespace(config=name, content="synthetic", pairs=pairs, steps=steps, warmup=warmup,
                                    cpu_baseline=False, content_sweep=False, pcie=False, parity_budget_s=par_s, parity_min_pairs=par_min)
        t1 = time.perf_counter()
        comm1 = comm
        try:
            if CONFIGS[name][4] == -3:
                # BASELINE configs[4] at N = 1: the step's all-gather goes through the library's RCCL communicator
                # (gme_shard_gather, world 1) exactly as at N > 1
                os.environ["GME_BENCH_FORCE_DIST"] = "1"
                try:
                    comm1 = Comm(ctx, 0, 1, ctx.device)
                finally:
                    os.environ.pop("GME_BENCH_FORCE_DIST", None)
            d = measure(opt, ctx, comm1, 0, 1)
        except Exception as e:                      # noqa: BLE001 -- one config must not take the headline line down
            block[name] = {"error": repr(e)}
            continue
        finally:
            if comm1 is not comm:
                comm1.close()
        entry = {"workload": d["config"]["workload"], "pairs_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                 "pairs_per_step": d["config"]["pairs_per_step_per_gpu"], "steps": steps, "warmup": warmup, "streams": d["config"]["streams_per_gpu"],
                 "kernel": d["roofline"]["kernel"], "kernel_ms_per_launch": d["roofline"]["kernel_ms_per_launch"],
                 "hbm_frac_of_dominant_kernel": d["roofline"].get("hbm", d["roofline"])["frac"],
                 "parity_ok": d["parity"]["ok"], "pairs_checked_vs_c_oracle": d["parity"]["pairs_checked"],
                 "first_and_last_pair_checked": [d["parity"]["first"], d["parity"]["last"]],
                 "collective": d["config"]["collective"], "seconds": round(time.perf_counter() - t1, 1)}
        if "whole_step" in d["roofline"]:
            entry["whole_step_hbm_frac"] = d["roofline"]["whole_step"]["frac"]
        if "elimination" in d:
            entry["surviving_fraction"] = d["elimination"]["surviving_fraction"]
        if d["roofline"]["bound"] == "mfma":
            entry["int8_mfma_frac"] = d["roofline"]["frac"]
        block[name] = entry
    block["seconds"] = round(time.perf_counter() - t0, 1)
    return block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=None,
                    help="frame pairs resident and processed per step per GPU (default %d: launch ramps, tails and "
                         "the host-side solves of the GME stages cost the same per step whatever the batch)" % DEFAULT_PAIRS)
    ap.add_argument("--config", default="exh720", choices=sorted(CONFIGS))
    ap.add_argument("--content", default="synthetic", choices=["synthetic", "race", "pan240seq", "pan240x2", "noise", "flat"],
                    help="frame content of the exhaustive / walk configs (uploaded from the host unless synthetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-content-sweep", action="store_true")
    ap.add_argument("--no-pcie", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configs the default line also times")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d does not match WORLD_SIZE=%d: launch it with torch.distributed.run "
                 "--nproc-per-node %d (or plainly for --gpus 1)" % (args.gpus, world, args.gpus))
    if CONFIGS[args.config][4] < 0 and args.content != "synthetic":
        sys.exit("--content applies to the block-matching configs only")

    import _gme_native as native
    ndev = native.load_library().gme_device_count()
    ctx = native.Context(local % max(ndev, 1))          # rehearsals may put several ranks on one card
    comm = Comm(ctx, rank, world, local % max(ndev, 1))
    opt = types.SimpleNamespace(config=args.config, content=args.content, pairs=args.pairs, steps=args.steps, warmup=args.warmup,
                                cpu_baseline=not args.no_cpu_baseline, content_sweep=not args.no_content_sweep,
                                pcie=not args.no_pcie, parity_budget_s=PARITY_BUDGET_S)
    out = measure(opt, ctx, comm, rank, world)
    if rank == 0:
        default_command = (args.config == "exh720" and args.content == "synthetic" and world == 1 and args.pairs is None
                           and comm.kind == "none")
        if default_command and not args.no_secondary:
            out["secondary"] = secondary_block(ctx, comm)
        print(json.dumps(out), flush=True)
    degraded = comm.degraded
    comm.close()
    if degraded:                                    # every rank knows (the fallback is agreed collectively): the line is
        sys.exit(3)                                 # printed, but a launch that did not use the C ABI's RCCL is not a pass


if __name__ == "__main__":
    main()
