TAG=${1:-r03e}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "round3 or hostile or redo or benched" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
echo "== redo kernel with scalar anchors (redo2) vs LDS anchors (mseq), noise content"
bash tools/ab.sh "mseq redo2" "exh720 exh720mse" 2 "--content noise --pairs 512" 2>&1 | tail -8
bash tools/ab.sh "mseq redo2" "exh720 exh720mse" 1 "--content noise" 2>&1 | tail -4
bash tools/ab.sh "mseq redo2" "exh1080 exh1080mse" 1 "--content noise --pairs 256 --steps 5" 2>&1 | tail -4
for c in exh720 exh720mse; do
  bash tools/gpu_trace.sh $TAG/noise_${c} --config $c --content noise --pairs 512 --no-pcie 2>&1 | grep -E "^k_exh|value" | cut -c1-150
done
echo "== streamed GME (host frames) vs resident"
timeout -k 10 300 python3 tools/stream_gme.py 2>&1 | tail -12
