# usage (GPU box): bash tools/mfma_sweep.sh <outdir> "<tile> ..." "<config> ..."   -- pairs/s of the matrix-core MSE search per tile shape
O=${1:-gpurun_out/mfma_sweep}; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_mfma.py -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for c in $3; do
  for t in $2; do
    GME_EXH_MFMA=1 GME_MFMA_TILE=$t timeout -k 10 300 python3 bench.py --config $c --no-secondary --no-cpu-baseline --no-pcie --no-content-sweep > $O/${c}_$t.json 2> $O/${c}_$t.err || { echo "$c $t failed"; tail -3 $O/${c}_$t.err; exit 1; }
    python3 - $O/${c}_$t.json $c $t <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], round(d["value"]), "pairs/s", round(d["ms_per_step"], 3), "ms", d["parity"]["ok"], d["parity"]["pairs_checked"])
PY
  done
done
