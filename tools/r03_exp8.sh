TAG=${1:-r03m}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { n=$1; shift
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-content-sweep --no-secondary "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -3 $O/$n.err; return; }
  python3 -c "
import json
d = json.loads(open('$O/$n.json').read()); e = d.get('elimination', {}); p = d.get('pcie_inclusive', {})
print('%-18s %9.0f pairs/s  %.3f ms/step  parity %s  surviving %s  pcie %s' % ('$n', d['value'], d['ms_per_step'], d['parity']['ok'], round(e.get('surviving_fraction', 0), 4), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items() if k in ('value', 'fraction_of_copy_ceiling', 'equals_resident_result')}))"
}
run exh720mse --config exh720mse --no-pcie
run exh1080mse --config exh1080mse --pairs 512 --steps 5 --no-pcie
run mse_pan240x2 --config exh720mse --content pan240x2 --pairs 512 --no-pcie
run mse_noise512 --config exh720mse --content noise --pairs 512 --no-pcie
run gme720 --config gme720
echo "== streamed GME, decreasing chunk schedule"
timeout -k 10 300 python3 tools/stream_gme.py 2>&1 | tail -9
