TAG=${1:-r03g}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
echo "== where the packed table loses: per-kernel times, u32 table (redo2) vs 16+8-bit planes (cur)"
bash tools/ab_trace.sh "redo2 cur" "k_" --config exh720mse 2>&1 | grep -E "k_exh|k_sqbox"
echo "== pyramid: LDS-tiled kernel vs two launches per level (GME_PYR_NOLDS=1)"
bash tools/abenv.sh "- GME_PYR_NOLDS=1 GME_BENCH_STREAMS=1 GME_BENCH_STREAMS=1,GME_PYR_NOLDS=1" "gme720" 2 "--no-secondary" 2>&1 | tail -8
GME_BENCH_STREAMS=1 bash tools/gpu_trace.sh $TAG/gme720_1stream --config gme720 2>&1 | grep -E "^k_|value" | cut -c1-150
echo "== previous-tile probe (probe) vs current (cur)"
bash tools/ab.sh "cur probe" "exh720" 2 2>&1 | tail -4
bash tools/ab.sh "cur probe" "exh720" 1 "--content pan240x2 --pairs 512" 2>&1 | tail -2
bash tools/ab.sh "cur probe" "exh720" 1 "--content race --pairs 512" 2>&1 | tail -2
echo "== streamed GME, ready-first scheduling"
timeout -k 10 300 python3 tools/stream_gme.py 2>&1 | tail -8
