#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X hot path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            # N=1 directly
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over this rank's batch of synthetic frame pairs,
frames already resident in HBM.  Default workload = BASELINE.json configs[1]:
720x480 synthetic luma, bs=16, sw=16, exhaustive search, MAE.  Frame pairs shard across
ranks with no data-path collective ("weak" scaling: every rank holds its own batch);
torch.distributed (RCCL) is used for the barriers and the max-over-ranks time only.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      HBM view of the dominant kernel: algorithmic bytes / HIP-event kernel time
  cpu_baseline  the NumPy oracle (reference loop structure) timed on one host core over a
                bounded sample of the same workload (rank 0, N=1 only)
plus "valu": the same kernel against the measured v_qsad_pk_u16_u8 issue rate, which is
what actually bounds exhaustive search.
"""
import argparse
import json
import os
import sys
import time

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
}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
# measured on MI355X (tools/microbench/valu_rates2.hip, profiles/r01_valu_rates.txt):
# v_qsad_pk_u16_u8 issues one wave-instruction (64 lanes x 16 byte-abs-diffs) per ~16.3
# cycles per SIMD at ~2.35 GHz -> 1024 SIMDs * 1024 ops / 6.9 ns
QSAD_PEAK_OPS = 1024 * 1024 / 6.9e-9


def algorithmic_bytes(H, W, bs, gme=False):
    if gme:
        return 3 * H * W + 48                               # 2 frames in, 1 compensated frame out, params
    return 2 * H * W + 8 * (H // bs) * (W // bs)


def byte_ops_per_pair(H, W, bs, sw):
    """Valid candidates x bs^2 (exact count of byte abs-diffs the exhaustive search needs)."""
    def valid(n):
        tot = 0
        for o in range(0, n - bs + 1, bs):
            tot += sum(1 for w in range(-sw, sw + bs) if 0 <= o + w <= n - bs)
        return tot
    return valid(H) * valid(W) * bs * bs


def cpu_baseline(cfg, budget_s=14.0):
    """NumPy oracle on one core over whole block rows of pair (t=0, t=1) until `budget_s`."""
    H, W, bs, sw, proc, pnorm, seed, _ = cfg
    from oracle import gme_oracle
    import synth
    prev, cur = synth.frame(seed, 0, H, W), synth.frame(seed, 1, H, W)
    nbr, nbc = H // bs, W // bs
    mf = np.zeros((nbr, nbc, 2), np.int32)
    order = [nbr // 2] + [r for r in range(nbr) if r != nbr // 2]      # an interior row first
    done, t0 = [], time.perf_counter()
    for r in order:
        if proc == 0:
            gme_oracle.search_exhaustive(prev, cur, mf, H, W, pnorm, bs, sw, block_rows=(r, r + 1))
        else:
            strip = slice(r * bs, (r + 1) * bs)
            raise NotImplementedError(strip)
        done.append(r)
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    # weight rows by their candidate count so that edge rows do not skew the extrapolation
    def row_cands(r):
        return sum(1 for w in range(-sw, sw + bs) if 0 <= r * bs + w <= H - bs)
    frac = sum(row_cands(r) for r in done) / sum(row_cands(r) for r in range(nbr))
    return {"value": frac / el, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": "oracle/gme_oracle.py (NumPy, reference loop structure) on %d of %d block rows of pair t=0,1 "
                      "(%.1f%% of the pair's candidates) in %.1f s" % (len(done), nbr, 100 * frac, el),
            "rows_checked": done, "mf": mf}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=None,
                    help="frame pairs resident and processed per step per GPU (default %d: launch ramps, tails and "
                         "the host-side solves of the GME stages cost the same per step whatever the batch)" % DEFAULT_PAIRS)
    ap.add_argument("--config", default="exh720", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
    cfg = CONFIGS[args.config]
    H, W, bs, sw, proc, pnorm, seed, label = cfg

    import torch
    dist = None
    if world > 1 or os.environ.get("GME_BENCH_FORCE_DIST"):      # the latter rehearses RCCL with one rank
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        local = local % max(ndev, 1)               # rehearsals may put several ranks on one card
        torch.cuda.set_device(local)
        backend = os.environ.get("GME_BENCH_BACKEND", "nccl")      # nccl = RCCL; gloo only for rehearsal
        # RCCL prints a version banner on stdout when it initialises; stdout must carry the one
        # JSON line only, so park fd 1 on stderr until the communicator exists
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    def barrier():
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    import _gme_native as native
    ctx = native.Context(local)
    B = args.pairs if args.pairs is not None else DEFAULT_PAIRS
    gme = proc < 0
    # GME runs cut the resident pairs into `streams` ranges, each on its own HIP stream and host
    # thread: one range's host-side 3x3 solves are covered by the other ranges' kernels
    # (not for the exhaustive-search GME of configs[3]: its kernels run for tens of ms, the host gaps do
    # not matter and concurrent persistent kernels only contend: 17.5 k pairs/s on one stream, 15 k on three)
    streams = int(os.environ.get("GME_BENCH_STREAMS", "1" if proc == -2 else "3"))
    shard = None
    if proc == -3:
        import sequence
        n_frames = int(os.environ.get("GME_BENCH_FRAMES", "2000"))
        shard = sequence.ShardedSequence(H, W, n_frames, 1, rank=rank, world=world, ctx=ctx, streams=streams)
        shard.synth(seed)                          # each rank generates its own slice (pairs + 1 halo frame)
        B = shard.n_pairs                          # pairs of THIS rank; the step covers the whole sequence
    elif gme:
        import sequence
        shard = sequence.ShardedSequence(H, W, B + 1, 1, ctx=ctx, streams=streams)
        shard.synth(seed, rank * B)                # rank r holds frames t = r*B .. r*B+B (halo of fd=1 included)
    else:
        seq = native.Sequence(ctx, B + 1, H, W)
        seq.synth(seed, rank * B)
    if shard is not None:
        shard.sync()
    ctx.sync()

    if proc == -3:
        last = {}
        gather_dev = torch.device("cuda", local) if (dist is not None and dist.get_backend() == "nccl") else None

        def step():
            shard.invalidate()
            params, psnr = shard.estimate_and_compensate()
            rows = np.concatenate([params, psnr[:, None]], axis=1)
            if dist is not None:                   # the path's one exchange: 56 B per pair over RCCL
                rows = sequence.gather_parameters(rows, shard.n_pairs_total, rank, dist.get_world_size(), gather_dev)
            last["rows"] = rows
    elif gme:
        last = {}

        def step():
            # motion.global_motion_estimation + results.py:52-59,109 for every resident pair;
            # frames change between videos, so the pyramids are rebuilt inside the step
            shard.invalidate()
            if proc == -2:
                last["params"], last["psnr"] = shard.estimate_and_compensate(0, sw)
            else:
                last["params"], last["psnr"] = shard.estimate_and_compensate()
    else:
        def step():
            # per-frame auxiliary tables (box sums for the pruning bound / the MSE identity) are
            # derived from the frames: a new batch has to rebuild them, so every step does
            seq.invalidate_pyramids()
            seq.bbme(1, bs, sw, proc, pnorm)       # asynchronous launches on the context's stream

    for _ in range(args.warmup):
        step()
    if shard is not None:
        shard.sync()
    ctx.sync()
    barrier()
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    kernel_ms = ctx.timer_stop() / max(args.steps, 1)      # HIP events on the launch stream; synchronises
    if shard is not None:
        shard.sync()
    ctx.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if gme:
        kernel_ms = 1e3 * elapsed / max(args.steps, 1)     # several streams: the whole step on the wall clock
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # parity gate printed with the number: rank 0's first pair is the golden pair (seed 1234, t=0,1)
    parity = None
    mv = None if gme else seq.read_mv(0, 1)[0]
    gpath = os.path.join(REPO, "tests", "golden", "g2_synth720.npz")
    if rank == 0 and args.config in ("exh720", "exh720mse", "dia720", "dia720mse") and os.path.exists(gpath):
        parity = bool(np.array_equal(mv, np.load(gpath)["mf_sp%d_pn%d" % (proc, pnorm)]))
    if rank == 0 and args.config == "gme720":
        g4 = np.load(os.path.join(REPO, "tests", "golden", "g4_gme.npz"))
        import hashlib
        comp_sha = hashlib.sha256(shard.read_compensated(0).tobytes()).hexdigest()
        parity = bool(np.allclose(last["params"][0], g4["synth720_params"], rtol=1e-10, atol=1e-12)
                      and comp_sha == str(g4["synth720_comp_sha"]))

    if rank == 0:
        total_pairs = (shard.n_pairs_total if proc == -3 else world * B) * args.steps
        value = total_pairs / elapsed
        abytes = algorithmic_bytes(H, W, bs, gme) * B
        if proc == -3:
            rows = last["rows"]
            out_extra = {"pairs_total": int(shard.n_pairs_total), "gathered_rows": int(rows.shape[0]),
                         "mean_psnr_db": float(np.mean(rows[:, 6])),
                         "median_params": [float(x) for x in np.median(rows[:, :6], axis=0)]}
        achieved = abytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "frame-pairs/s + achieved HBM GB/s, 720x480 bs=16 sw=16 exhaustive, 1->8 GPU"
                      if args.config == "exh720" else "frame-pairs/s, " + args.config,
            "value": value, "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if proc == -3 else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": label, "pairs_per_step_per_gpu": B, "frame_distance": 1,
                       "sharding": "frame pairs across ranks, no data-path collective",
                       "streams_per_gpu": len(shard.lanes) if shard is not None else 1},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_exh_sea16p<3, 6>" if (proc, pnorm, bs, sw) == (0, 0, 16, 16) else
                                   ("whole step (all kernels + host solves)" if gme else "see DESIGN.md"),
                         "kernel_ms_per_launch": kernel_ms,
                         "algorithmic_bytes_per_launch": abytes},
            "parity_first_pair_vs_reference_golden": parity,
        }
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this same
        # command (profiles/): (2 x FETCH_SIZE + WRITE_SIZE) KB, the x2 being the guide's gfx950
        # FETCH_SIZE correction, which the L2 miss count (TCC_MISS x 128 B) confirms here.
        prof = os.path.join(REPO, "profiles", "r01_final_exh720_pmc_summary.txt")
        if args.config == "exh720" and B == DEFAULT_PAIRS and os.path.exists(prof):    # the profile is of the default command
            vals = {}
            for line in open(prof):
                f = line.split()
                for name in ("FETCH_SIZE", "WRITE_SIZE"):
                    if name in f and not line.startswith("#"):
                        vals[name] = float(f[-1].split("=")[1])
            if len(vals) == 2:
                out["roofline"]["traffic"] = int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
                out["roofline"]["traffic_unit"] = "bytes per launch, from profiles/r01_final_exh720_pmc_summary.txt"
        if proc == 0:
            ops = byte_ops_per_pair(H, W, bs, sw) * B / (kernel_ms * 1e-3)
            # brute-force-equivalent rate: the exhaustive search's nominal byte abs-diffs per second.
            # With exact successive elimination most candidates are never evaluated, so this may
            # exceed the instruction-issue ceiling of a brute-force kernel (frac > 1).
            out["valu"] = {"bound": "v_qsad_pk_u16_u8 issue (brute-force equivalent)", "achieved": ops,
                           "peak": QSAD_PEAK_OPS, "unit": "byte-abs-diff/s", "frac": ops / QSAD_PEAK_OPS}
        if proc >= 0 and world == 1:
            # host-buffer (PCIe-inclusive) rate, NOT `value`: frames cross to the device, the fields
            # come back (SURVEY.md §8(d) "end-to-end number including H2D/D2H")
            n_e2e = B                                              # same launch size as the timed steps: a profile of
                                                                   # this command averages like-sized launches only
            host_frames = np.stack([seq.read_frame(i) for i in range(n_e2e + 1)])
            seq2 = native.Sequence(ctx, n_e2e + 1, H, W)          # its own small sequence: upload, search, read back
            seq2.upload(0, host_frames[:2])                        # first touch of the buffers outside the timing
            ctx.sync()
            t_e = time.perf_counter()
            seq2.upload(0, host_frames)
            seq2.bbme(1, bs, sw, proc, pnorm)
            _ = seq2.read_mv(0, n_e2e)
            t_e = time.perf_counter() - t_e
            seq2.close()
            out["pcie_inclusive"] = {"value": n_e2e / t_e, "unit": "frame-pairs/s",
                                     "note": "upload of %d frames from pageable host memory + search + read-back of %d fields; "
                                             "each frame crosses once" % (n_e2e + 1, n_e2e)}
        if proc == -3:
            out["sequence"] = out_extra
        if world == 1 and not args.no_cpu_baseline and proc == 0:
            cb = cpu_baseline(cfg)
            rows, ref_mf = cb.pop("rows_checked"), cb.pop("mf")
            cb["matches_gpu"] = bool(all(np.array_equal(ref_mf[r], mv[r]) for r in rows))
            out["cpu_baseline"] = cb
            # BASELINE.md §3 "optimised CPU": the integer C oracle (gcc -O3, one core) on the same
            # pair, so the GPU figure is not only measured against interpreter overhead
            try:
                sys.path.insert(0, os.path.join(REPO, "tests"))
                from helpers import c_oracle
                import synth
                p0, p1 = synth.frame(seed, 0, H, W), synth.frame(seed, 1, H, W)
                co = c_oracle()
                t_c = time.perf_counter()
                mf_c = co.bbme(p0, p1, bs, sw, proc, pnorm)
                t_c = time.perf_counter() - t_c
                out["cpu_baseline_c"] = {"value": 1.0 / t_c, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
                                         "sample": "oracle/gme_oracle.c (integer C, gcc -O3) on the whole pair t=0,1 in %.2f s" % t_c,
                                         "matches_gpu": bool(np.array_equal(mf_c, mv))}
            except Exception as e:      # the C oracle is optional test infrastructure
                out["cpu_baseline_c"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
