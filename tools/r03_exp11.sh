TAG=${1:-r03y}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python3 tools/stream_gme_timeline.py 512 2 512 | head -10
python3 tools/stream_gme.py | tail -9
run() { n=$1; shift
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-content-sweep --no-secondary "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -3 $O/$n.err; return; }
  python3 -c "
import json
d = json.loads(open('$O/$n.json').read()); p = d.get('pcie_inclusive', {})
print('%-18s %9.0f pairs/s  %.3f ms/step  parity %s  pcie %s' % ('$n', d['value'], d['ms_per_step'], d['parity']['ok'], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items() if k in ('value', 'fraction_of_copy_ceiling', 'equals_resident_result')}))"
}
run gme720 --config gme720
GME_BENCH_STREAMS=1 run gme720_1stream --config gme720 --no-pcie
run exh720 --config exh720
run exh720mse --config exh720mse
