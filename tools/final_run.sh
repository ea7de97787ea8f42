# The GPU calls that regenerate everything DESIGN.md / profiles/ quote for a round (each part fits one gpurun call of <= 20 min):
#   bash tools/final_run.sh <tag> tests      GPU tests, the default bench line (with secondary block, sweeps, CPU baseline), rehearsals, scale_run
#   bash tools/final_run.sh <tag> benches    one bench line per config (CPU baseline included) + hostile / real content lines
#   bash tools/final_run.sh <tag> prof "<config:kernel-substring> ..."     tools/profile_configs.sh (bench + rocprofv3 kernel stats + PMC passes)
#       round 4: "exh720:k_ exh720mse:k_ exh720mse_vec:k_ gme720:k_ tss720:k_ tdl720:k_ dia720mse:k_ exh1080:k_ exh1080mse:k_ gme1080exh:k_" (8.5 GPU-minutes)
#       and "seq1080:k_ gme1080:k_ dia720:k_ tss_bs4sw2:k_ gme_pan240_bs12fd5:k_ gme720dev:k_ exh1080mse_mfma:k_" (5); tests/test_host.py checks that
#       they belong to the committed kernels
# Copy gpurun_out/<tag>/*_{bench.json,kernel_stats.csv,pmc_summary.txt} into profiles/ afterwards (tools/collect_profiles.sh).
TAG=${1:-r04_final}; PART=${2:-tests}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
line() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
e, p, c = d.get("elimination", {}), d.get("pcie_inclusive", {}), d.get("cpu_baseline", {})
print("%-16s %9.0f pairs/s %8.3f ms/step  parity %s/%s  surviving %s  redo %s  pcie %s  cpu %s  traffic %s" % (
    sys.argv[1], d["value"], d["ms_per_step"], d["parity"]["ok"], d["parity"]["pairs_checked"],
    round(e["surviving_fraction"], 4) if e else None, e.get("tiles_redone_by_brute_force"),
    round(p["value"]) if p else None, round(c["value"], 4) if c else None, d["roofline"].get("traffic")))
PY
}
if [ "$PART" = tests ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
  timeout -k 10 600 python3 bench.py > $O/${TAG}_default_bench.json 2> $O/default_bench.err || { echo "default bench failed"; tail -5 $O/default_bench.err; }
  line default $O/${TAG}_default_bench.json
  python3 - <<PY
import json
d = json.loads(open("$O/${TAG}_default_bench.json").read())
for k, v in d.get("secondary", {}).items():
    print("  secondary", k, v if not isinstance(v, dict) else {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("pairs_per_s", "ms_per_step", "parity_ok", "pairs_checked_vs_c_oracle", "seconds", "error", "surviving_fraction")})
for k in ("content_sweep", "content_sweep_mse"):
    for c, v in d.get(k, {}).items():
        print("  ", k, c, round(v["pairs_per_s"]), v["surviving_fraction"], v["tiles_redone_by_brute_force"], v["parity_ok_sampled"])
print("  cpu_baseline", {k: v for k, v in d["cpu_baseline"].items() if k != "sample"})
print("  pcie", {k: v for k, v in d.get("pcie_inclusive", {}).items() if k != "note"})
PY
  echo "== rehearsals: 2 gloo ranks on one GPU (rows through the host); 1 rank over the C ABI's RCCL (device-to-device gather per step)"
  GME_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 2>$O/gloo2.err | tail -1 > $O/${TAG}_gloo2_bench.json
  GME_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-content-sweep --no-pcie 2>$O/rccl1.err > $O/${TAG}_rccl1_bench.json
  for c in exh720 seq1080; do
    GME_BENCH_FORCE_DIST=1 GME_BENCH_FRAMES=400 MASTER_ADDR=127.0.0.1 MASTER_PORT=29545 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-content-sweep --no-pcie 2>/dev/null > $O/rccl1_$c.json; line rccl1_$c $O/rccl1_$c.json
  done
  python3 - <<PY
import json
for f in ("gloo2", "rccl1"):
    try:
        d = json.loads(open("$O/${TAG}_%s_bench.json" % f).read())
        print(f, round(d["value"]), "ms/step", round(d["ms_per_step"], 3), d["config"]["collective"], d["config"]["rccl_reports"], "parity", d["parity"]["ok"], d["parity"].get("gathered_rows"))
    except Exception as e:
        print(f, "failed", e)
PY
  bash tools/gloo2_seq1080.sh $O/gloo2_seq1080 2>&1 | head -1; cp $O/gloo2_seq1080/gloo2_seq1080_bench.json $O/${TAG}_gloo2_seq1080_bench.json 2>/dev/null
  bash tools/rccl_dup_rehearsal.sh $O/rccl_dup 2>&1 | head -14
  bash tools/scale_run.sh exh720 20 3 $O/scale 2>&1 | tail -3
elif [ "$PART" = benches ]; then
  for c in exh720mse exh720mse_vec exh1080 exh1080mse exh1080mse_mfma dia720 dia720mse tss720 tdl720 gme720 gme720dev gme1080 gme1080exh seq1080 tss_bs4sw2 gme_pan240_bs12fd5; do
    timeout -k 10 400 python3 bench.py --config $c --no-secondary 2>$O/${c}_bench.err > $O/${TAG}_${c}_bench.json || { echo "$c failed"; tail -3 $O/${c}_bench.err; continue; }
    line $c $O/${TAG}_${c}_bench.json
  done
  GME_BENCH_STREAMS=1 timeout -k 10 300 python3 bench.py --config gme720 --no-secondary --no-cpu-baseline --no-pcie 2>/dev/null > $O/${TAG}_gme720_1stream_bench.json; line gme720_1stream $O/${TAG}_gme720_1stream_bench.json
  GME_EXH_BRUTE=1 timeout -k 10 300 python3 bench.py --no-secondary --no-cpu-baseline --no-content-sweep --no-pcie 2>/dev/null > $O/${TAG}_exh720_brute_bench.json; line exh720_brute $O/${TAG}_exh720_brute_bench.json
  GME_BENCH_STREAMS=1 GME_DEVICE_SOLVE=1 timeout -k 10 300 python3 bench.py --config gme720 --no-secondary --no-cpu-baseline --no-pcie 2>/dev/null > $O/${TAG}_gme720_1stream_devsolve_bench.json; line gme720_1stream_devsolve $O/${TAG}_gme720_1stream_devsolve_bench.json
  GME_SEA_QUOTA=0 timeout -k 10 300 python3 bench.py --content pan240x2 --no-secondary --no-cpu-baseline --no-pcie --no-content-sweep 2>/dev/null > $O/${TAG}_exh720_pan240x2_noc2_bench.json; line exh720_pan240x2_noc2 $O/${TAG}_exh720_pan240x2_noc2_bench.json
  for c in noise flat race pan240x2 pan240seq; do
    for cfg in exh720 exh720mse exh720mse_vec; do
      timeout -k 10 300 python3 bench.py --config $cfg --content $c --no-secondary --no-cpu-baseline --no-pcie 2>/dev/null > $O/${TAG}_${cfg}_${c}_bench.json; line ${cfg}_$c $O/${TAG}_${cfg}_${c}_bench.json
    done
  done
else
  bash tools/profile_configs.sh $TAG "$3"
fi
