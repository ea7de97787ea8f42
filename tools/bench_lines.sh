# usage (GPU box): bash tools/bench_lines.sh <tag> "<config> ..."   -- one bench line per config into gpurun_out/<tag>/<tag>_<config>_bench.json
TAG=$1; O=gpurun_out/$TAG
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $O
for c in $2; do
  timeout -k 10 400 python3 bench.py --config $c --no-secondary 2>$O/${c}_bench.err > $O/${TAG}_${c}_bench.json || { echo "$c failed"; tail -3 $O/${c}_bench.err; continue; }
  python3 - <<PY
import json
d = json.loads(open("$O/${TAG}_${c}_bench.json").read())
print("$c", round(d["value"]), round(d["ms_per_step"], 3), d["parity"]["ok"], "traffic", d["roofline"].get("traffic"), "frac", d["roofline"].get("frac"), (d["roofline"].get("whole_step") or {}).get("frac"))
PY
done
