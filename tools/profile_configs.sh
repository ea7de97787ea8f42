#!/bin/bash
# usage (on the GPU box): bash tools/profile_configs.sh <tag> "<config:kernel-substring> ..."
# For every config: tools/gpu_profile.sh (bench line, rocprofv3 kernel stats, separate PMC passes), then the
# summaries that get committed under profiles/: <tag>_<config>_{bench.json,kernel_stats.csv,pmc_summary.txt}.
TAG=$1
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/$TAG
for item in $2; do
  c=${item%%:*}; k=${item#*:}
  O=gpurun_out/$TAG/$c
  bash tools/gpu_profile.sh $TAG/$c --config $c > gpurun_out/$TAG/${c}_profile.log 2>&1 || { echo "$c: profile run failed"; exit 1; }
  python3 tools/pmc_summary.py $O "$k" > gpurun_out/$TAG/${TAG}_${c}_pmc_summary.txt 2>/dev/null
  cp $O/bench.json gpurun_out/$TAG/${TAG}_${c}_bench.json
  f=$(ls $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/$TAG/${TAG}_${c}_kernel_stats.csv
  python3 - <<PY
import json
d = json.loads(open("$O/bench.json").read())
print("$c", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "parity", d["parity"]["ok"], d.get("elimination"))
PY
done
