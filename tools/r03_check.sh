# One GPU call of round 3: GPU tests, the default bench line (with its secondary block), the multi-rank rehearsals.
TAG=${1:-r03b}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout -k 10 600 python3 bench.py > $O/default_bench.json 2> $O/default_bench.err || { echo "default bench failed"; tail -5 $O/default_bench.err; }
python3 - <<PY
import json
d = json.loads(open("$O/default_bench.json").read())
print("exh720", round(d["value"]), "parity", d["parity"]["ok"], "traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_source", "")[:60])
for k, v in d.get("secondary", {}).items():
    print("  secondary", k, v if not isinstance(v, dict) else {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("pairs_per_s", "ms_per_step", "parity_ok", "pairs_checked_vs_c_oracle", "seconds", "error", "surviving_fraction")})
for k in ("content_sweep", "content_sweep_mse"):
    for c, v in d.get(k, {}).items():
        print("  ", k, c, round(v["pairs_per_s"]), v["surviving_fraction"], v["tiles_redone_by_brute_force"], v["parity_ok_sampled"])
print("  cpu_baseline host", d["cpu_baseline"].get("host"))
PY
echo "== rehearsal: 2 gloo ranks on one GPU (rows through the host), then 1 rank over the C ABI's RCCL (device-to-device gather per step)"
GME_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 2>$O/gloo2.err | tail -1 > $O/gloo2.json
GME_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-content-sweep --no-pcie 2>$O/rccl1.err > $O/rccl1.json
python3 - <<PY
import json
for f in ("gloo2", "rccl1"):
    try:
        d = json.loads(open("$O/%s.json" % f).read())
        print(f, round(d["value"]), "ms/step", round(d["ms_per_step"], 3), d["config"]["collective"], d["config"]["rccl_reports"], "parity", d["parity"]["ok"], d["parity"].get("gathered_rows"))
    except Exception as e:
        print(f, "failed", e)
PY
bash tools/scale_run.sh exh720 20 3 $O/scale 2>&1 | tail -3
