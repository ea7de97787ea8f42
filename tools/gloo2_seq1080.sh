# usage (GPU box, one card): bash tools/gloo2_seq1080.sh <outdir>  -- BASELINE configs[4] with TWO ranks (both on device 0, rows
# exchanged over gloo): the sharding of the 2000-frame video with its halo frame, the gather of the parameter + PSNR rows and
# rank 0's oracle check of rows OWNED BY THE OTHER RANK.  GME_BENCH_FRAMES keeps it short.
O=${1:-gpurun_out/gloo2_seq1080}; mkdir -p $O
cd ${GRAFT_REPO_ROOT:-/root/repo}
GME_BENCH_BACKEND=gloo GME_BENCH_FRAMES=${GME_BENCH_FRAMES:-401} timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29588 bench.py --gpus 2 --config seq1080 --steps 3 --warmup 1 2>$O/stderr.log | tail -1 > $O/gloo2_seq1080_bench.json
python3 - <<PY
import json
d = json.loads(open("$O/gloo2_seq1080_bench.json").read())
print("gloo2 seq1080", round(d["value"]), "pairs/s", round(d["ms_per_step"], 2), "ms/step", d["config"]["collective"], "n_gpus", d["n_gpus"],
      "parity", d["parity"]["ok"], {k: v for k, v in d["parity"].items() if k not in ("checker",)})
PY
tail -3 $O/stderr.log | cut -c1-300
