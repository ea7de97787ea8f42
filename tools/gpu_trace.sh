#!/bin/bash
# tools/gpu_trace.sh <tag> <bench args...>: bench line + rocprofv3 kernel stats (+ optional PMC) on the GPU box
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err; cut -c1-330 $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 || echo trace failed
python3 - <<PY
import csv,glob,re
for f in glob.glob("$OUT/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(k_\w+(<\d+>)?|__amd\w+)",r["Name"]); n=m.group(1) if m else r["Name"][:30]
        print("%-28s calls=%-4s avg_us=%9.1f total_ms=%8.2f pct=%s"%(n,r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6,r["Percentage"]))
PY
if [ -n "$PMC" ]; then
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/bench.py --no-cpu-baseline "$@" --steps 3 --warmup 1 > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT "$PMC"
fi
