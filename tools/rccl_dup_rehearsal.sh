#!/bin/bash
# Rehearsal of the all-ranks-agree bring-up on ONE card with two ranks of `bench.py --gpus 2` on device 0, twice:
#  (1) as launched: phase 1 of sequence.comm_init sees the same PCI bus id from both ranks and refuses -- both ranks report
#      CommUnavailable before anybody is inside ncclCommInitRank (round 4);
#  (2) GME_COMM_ALLOW_SHARED_DEVICE=1: the check is skipped and RCCL itself refuses ("Duplicate GPU detected",
#      ncclInvalidUsage) inside ncclCommInitRank on both ranks -- the REAL failure: both must report CommUnavailable (phase 2).
# Either way nobody may hang and the launch must end: the torch.distributed fallback meets the same refusal and exits
# non-zero -- that is the expected end here.
# usage: bash tools/rccl_dup_rehearsal.sh <outdir>
O=${1:-gpurun_out/rccl_dup}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $O
for mode in phase1 rccl; do
  [ $mode = rccl ] && export GME_COMM_ALLOW_SHARED_DEVICE=1
  t0=$(date +%s)
  timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
      bench.py --gpus 2 --steps 3 --warmup 1 > $O/stdout_$mode.log 2> $O/stderr_$mode.log
  rc=$?
  t1=$(date +%s)
  echo "rccl_dup_rehearsal[$mode]: exit code $rc after $((t1 - t0)) s (124 / 137 = hung and killed by timeout)"
  grep -c "C-ABI RCCL communicator unavailable" $O/stderr_$mode.log | sed 's/^/  ranks that reported CommUnavailable: /'
  grep -i "duplicate\|invalid usage\|ncclInvalid\|unavailable\|share a device" $O/stderr_$mode.log | cut -c1-300 | sort | uniq -c | sort -rn | sed -n 1,6p
  tail -2 $O/stdout_$mode.log | cut -c1-300
done
exit 0
