# One gpurun call that regenerates everything DESIGN.md / profiles/ quote: tests, headline profile
# (bench + rocprofv3 stats + PMC), one bench line per config, content sweep, GME kernel split,
# multi-rank rehearsals.  usage (on the GPU box): bash tools/final_run.sh <tag>
TAG=${1:-r02_final}
cd /root/repo
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
bash tools/gpu_profile.sh $TAG/exh720 > $O/exh720_profile.log 2>&1; tail -1 $O/exh720_profile.log
python3 tools/pmc_summary.py $O/exh720 k_exh_sea16p > $O/exh720_pmc_summary.txt 2>/dev/null
for c in exh720mse exh1080 exh1080mse dia720 dia720mse gme720 gme1080 gme1080exh seq1080; do
  timeout -k 10 300 python3 bench.py --config $c 2>$O/${c}_bench.err > $O/${c}_bench.json
  python3 -c "
import json
d=json.loads(open('$O/${c}_bench.json').read()); print('$c', round(d['value']), round(d['ms_per_step'],3), d['config'].get('pairs_per_step_per_gpu'), d['config'].get('streams_per_gpu'), 'parity', d['parity']['ok'], d['parity']['pairs_checked'], 'cpu', round(d.get('cpu_baseline',{}).get('value',0),4))"
done
GME_EXH_BRUTE=1 timeout -k 10 300 python3 bench.py --config exh720 --no-cpu-baseline --no-content-sweep --no-pcie 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('exh720 brute', round(d['value']), d['parity']['ok'])"
for c in noise flat race pan240seq; do
  timeout -k 10 300 python3 bench.py --config exh720 --content $c --no-cpu-baseline --no-pcie 2>/dev/null > $O/exh720_${c}_bench.json
  python3 -c "
import json
d=json.loads(open('$O/exh720_${c}_bench.json').read()); print('exh720 content $c', round(d['value']), d['parity']['ok'], d.get('elimination'))"
done
GME_BENCH_STREAMS=1 PMC=k_walk16 bash tools/gpu_trace.sh $TAG/gme720_1stream --config gme720 --pairs 2048 > $O/gme720_trace.log 2>&1; grep -E "calls=|value" $O/gme720_trace.log | cut -c1-120
echo "== multi-rank rehearsal (torch.distributed/gloo, 2 ranks on one GPU; then the C ABI's RCCL communicator with 1 rank)"
for c in exh720 seq1080; do
GME_BENCH_BACKEND=gloo GME_BENCH_FRAMES=200 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-content-sweep --no-pcie 2>/dev/null | tail -1 | cut -c1-600
done
GME_BENCH_FORCE_DIST=1 GME_BENCH_FRAMES=200 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python3 bench.py --config seq1080 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-700
