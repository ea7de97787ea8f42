#!/bin/bash
# usage: bash tools/pmc_ab.sh <tag> "<lib names under tools/microbench>" "<config[:content]> ..." <kernel substring>
# Same-box issue counters (two SQ passes) of library variants: per dispatch of the named kernel the vector / scalar / LDS
# instruction counts, busy and wait cycles -- what moved when a variant is faster or slower.
TAG=$1; LIBS=$2; CFGS=$3; KERN=${4:-k_exh_sea16p}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
L=global-motion-estimation_amd/lib/libgme_hip.so
cp $L /tmp/keep.so; trap 'cp /tmp/keep.so $ROOT/$L' EXIT
cd /tmp && export TMPDIR=/tmp
for v in $LIBS; do
  cp $ROOT/tools/microbench/libgme_$v.so $ROOT/$L
  for cc in $CFGS; do
    c=${cc%%:*}; content=""; [ "$cc" != "$c" ] && content="--content ${cc#*:}"
    O=$ROOT/gpurun_out/$TAG/${v}_${cc/:/_}; mkdir -p $O
    for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
      N=$(echo $C | tr ' ' '_' | cut -c1-30)
      timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -- python3 $ROOT/bench.py --config $c $content --no-cpu-baseline --no-content-sweep --no-pcie --no-secondary --steps 3 --warmup 1 > $O/pmc_$N.log 2>&1 || echo "pmc failed $v $cc"
    done
    echo "== $v $cc"; python3 $ROOT/tools/pmc_summary.py $O "$KERN" | grep -v "^#" | awk '{print $(NF-2), $NF}' | tr '\n' ' '; echo
  done
done
