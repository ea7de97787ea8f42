TAG=${1:-r03c}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "walk or parity or smoke or golden or pan240 or cli" > $O/pytest_walk.log 2>&1; tail -3 $O/pytest_walk.log
timeout -k 10 200 python3 tools/soak.py 60 7 walk > $O/soak_walk.log 2>&1; tail -2 $O/soak_walk.log
for c in tss720 tdl720 dia720mse dia720 gme720; do
  timeout -k 10 300 python3 bench.py --config $c --no-cpu-baseline 2>$O/${c}.err > $O/${c}.json
  python3 -c "
import json
d=json.loads(open('$O/${c}.json').read()); print('$c', round(d['value']), round(d['ms_per_step'],3), 'parity', d['parity']['ok'], d['roofline']['kernel'][:50])"
done
