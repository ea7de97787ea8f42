# usage (GPU box): bash tools/default_rerun.sh <tag>   -- the default bench line alone (with its secondary block), into gpurun_out/<tag>/
TAG=${1:-r04_final}; O=gpurun_out/$TAG
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $O
t0=$(date +%s)
timeout -k 10 600 python3 bench.py > $O/${TAG}_default_bench.json 2> $O/default_bench.err || { echo "default bench failed"; tail -5 $O/default_bench.err; }
echo "default bench: $(( $(date +%s) - t0 )) s"
python3 - <<PY
import json
d = json.loads(open("$O/${TAG}_default_bench.json").read())
print("default", round(d["value"]), d["ms_per_step"], d["parity"]["ok"], d["roofline"]["frac"], d["roofline"].get("traffic"))
print({k: (round(v["pairs_per_s"]) if isinstance(v, dict) and "pairs_per_s" in v else v) for k, v in d["secondary"].items()})
PY
