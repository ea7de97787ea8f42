TAG=${1:-r03h}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
echo "== u32 table with 16-byte reads (cur2) vs redo2 (u32, 4 reads) vs cur (24-bit planes)"
bash tools/ab.sh "redo2 cur cur2" "exh720mse" 2 2>&1 | tail -6
bash tools/ab.sh "redo2 cur2" "exh1080mse" 1 "--pairs 512 --steps 5" 2>&1 | tail -2
bash tools/ab.sh "redo2 cur cur2" "exh720mse exh720" 1 "--content noise --pairs 512" 2>&1 | tail -6
bash tools/ab.sh "cur cur2" "exh720" 2 2>&1 | tail -4
echo "== streamed GME, lanes set up once"
timeout -k 10 300 python3 tools/stream_gme.py 2>&1 | tail -9
