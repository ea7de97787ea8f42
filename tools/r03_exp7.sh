TAG=${1:-r03j}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 200 python3 tools/soak.py 90 11 walk > $O/soak_walk.log 2>&1; tail -2 $O/soak_walk.log
run() { n=$1; shift
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-content-sweep --no-secondary --no-pcie "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -3 $O/$n.err; return; }
  python3 -c "
import json
d = json.loads(open('$O/$n.json').read())
print('%-18s %9.0f pairs/s  %.3f ms/step  parity %s' % ('$n', d['value'], d['ms_per_step'], d['parity']['ok']))"
}
run dia720mse --config dia720mse
run dia720 --config dia720
run gme720 --config gme720
GME_BENCH_STREAMS=1 run gme720_1stream --config gme720
run gme1080 --config gme1080 --pairs 512 --steps 5
run dia720mse_race --config dia720mse --content race --pairs 512
run dia720mse_noise --config dia720mse --content noise --pairs 512
GME_BENCH_STREAMS=1 bash tools/gpu_trace.sh $TAG/gme720_1stream --config gme720 --no-pcie 2>&1 | grep -E "^k_" | cut -c1-150
echo "== streamed GME, lanes set up once, larger chunks"
timeout -k 10 300 python3 tools/stream_gme.py 2>&1 | tail -9
