TAG=${1:-r03d}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
echo "== MSE bound resolution A/B (base = floor(LBx/2^14) vs mseq = floor(LBx/32))"
bash tools/ab.sh "base mseq" "exh720mse" 2 2>&1 | tail -4
bash tools/ab.sh "base mseq" "exh1080mse" 1 "--pairs 512 --steps 5" 2>&1 | tail -2
bash tools/ab.sh "base mseq" "exh720mse" 1 "--content pan240x2 --pairs 512" 2>&1 | tail -2
bash tools/ab.sh "base mseq" "exh720mse" 1 "--content noise --pairs 512" 2>&1 | tail -2
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "benched or hostile or redo or gme1080exh or streamed" > $O/pytest_mse.log 2>&1; tail -2 $O/pytest_mse.log
echo "== hostile content: kernel split at 512 pairs (elimination + redo) vs brute force alone"
for v in "" "GME_EXH_BRUTE=1"; do for c in exh720 exh720mse; do
  env $v bash tools/gpu_trace.sh $TAG/noise_${c}_${v:-sea} --config $c --content noise --pairs 512 --no-pcie 2>&1 | grep -E "^k_exh|value" | cut -c1-150
done; done
