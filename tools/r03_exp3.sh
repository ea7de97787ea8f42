TAG=${1:-r03f}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
echo "== packed table (sq24) and lanes per patch (lpp2, lpp4) vs redo2, synthetic"
bash tools/ab.sh "redo2 sq24 lpp2 lpp4" "exh720mse" 2 2>&1 | tail -8
bash tools/ab.sh "redo2 sq24 lpp2 lpp4" "exh1080mse" 1 "--pairs 512 --steps 5" 2>&1 | tail -4
echo "== redo kernel: opaque thread index (redo3), + min 6 waves (redo3w6), noise"
bash tools/ab.sh "sq24 redo3 redo3w6" "exh720mse exh720" 2 "--content noise --pairs 512" 2>&1 | tail -12
bash tools/ab.sh "sq24 redo3 redo3w6" "exh1080mse" 1 "--content noise --pairs 256 --steps 5" 2>&1 | tail -3
echo "== streamed GME (host frames), shared upload stream"
timeout -k 10 300 python3 tools/stream_gme.py 2>&1 | tail -8
