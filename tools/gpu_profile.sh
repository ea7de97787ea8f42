#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats, PMC passes.
# Usage: tools/gpu_profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== bench" && timeout -k 10 400 python3 $ROOT/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json
echo "== kernel trace" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-content-sweep --no-pcie --no-secondary "$@" > $OUT/trace.log 2>&1 || echo "trace failed"
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  echo "== pmc $C" && timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/bench.py --no-cpu-baseline --no-content-sweep --no-pcie --no-secondary "$@" --steps 3 --warmup 1 > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed"
done
find $OUT -name "*.csv" | head -30
