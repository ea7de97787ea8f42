#!/bin/bash
# usage: bash tools/scale_run.sh [config] [steps] [warmup] [out_dir]
# The scaling series of bench.py on ONE node: N = 1, 2, 4, 8 (as far as the node has GPUs), one process per GPU,
# launched the way the driver does.  Each N is its own `python -m torch.distributed.run` started from this shell --
# nothing here touches a GPU before the ranks exist (no exec from a process that has initialised HIP).
# Prints one line per N and leaves the JSON lines in <out_dir>/scale_<config>_N<k>.json.  Efficiency is for the reader
# (or the driver) to compute from the per-N values; a 1-GPU box yields the N=1 line only.
CONFIG=${1:-exh720}; STEPS=${2:-20}; WARMUP=${3:-3}
cd "$(dirname "$0")/.."
OUT=${4:-gpurun_out/scale}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NGPU=$(python3 -c "import sys; sys.path.insert(0, 'global-motion-estimation_amd'); import _gme_native as n; print(n.load_library().gme_device_count())")
echo "devices visible: $NGPU"
PORT=29650
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && break
  PORT=$((PORT + 1))
  F="$OUT/scale_${CONFIG}_N${N}.json"
  if [ "$N" = 1 ]; then
    timeout -k 10 900 python3 bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --config $CONFIG --no-secondary > "$F" 2> "$F.err"
  else
    timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $N --steps $STEPS --warmup $WARMUP --config $CONFIG > "$F" 2> "$F.err"
  fi
  rc=$?
  [ $rc -ne 0 ] && { echo "N=$N failed (rc $rc): $(tail -3 "$F.err")"; break; }
  python3 - "$F" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("N=%d  %.0f %s  %.3f ms/step  collective=%s  rccl_reports=%s  parity=%s" % (
    d["n_gpus"], d["value"], d["unit"], d["ms_per_step"], d["config"]["collective"], d["config"].get("rccl_reports"), d["parity"]["ok"]))
PY
done
