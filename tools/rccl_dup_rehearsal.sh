#!/bin/bash
# Rehearsal of the all-ranks-agree bring-up under a REAL RCCL failure (on the GPU box, one card): two ranks of
# `bench.py --gpus 2` on device 0.  RCCL refuses two ranks on one device, so ncclCommInitRank fails on both: both ranks
# must report CommUnavailable (sequence.comm_init, phase 2), nobody may hang, and the launch must end (the
# torch.distributed fallback meets the same refusal and exits non-zero -- that is the expected end here).
# usage: bash tools/rccl_dup_rehearsal.sh <outdir>
O=${1:-gpurun_out/rccl_dup}; mkdir -p $O
cd ${GRAFT_REPO_ROOT:-/root/repo}
t0=$(date +%s)
timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
    bench.py --gpus 2 --steps 3 --warmup 1 > $O/stdout.log 2> $O/stderr.log
rc=$?
t1=$(date +%s)
echo "rccl_dup_rehearsal: exit code $rc after $((t1 - t0)) s (124 / 137 = hung and killed by timeout)"
grep -c "C-ABI RCCL communicator unavailable" $O/stderr.log | sed 's/^/  ranks that reported CommUnavailable: /'
grep -i "duplicate\|invalid usage\|ncclInvalid\|unavailable" $O/stderr.log | cut -c1-300 | sort | uniq -c | sort -rn | sed -n 1,8p
tail -2 $O/stdout.log | cut -c1-400
exit 0
