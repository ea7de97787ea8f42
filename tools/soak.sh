# Non-default code paths of the exhaustive kernels under the parity tests (one GPU call):
# tile shapes x schedules x phase-E variants.  usage (GPU box): bash tools/soak.sh
cd /root/repo
K="schedules or elimination or batched_exhaustive or full_size or golden_small or random_geometry"
for tile in 1x16 2x8 4x4 1x8 2x4 4x2 2x6 4x3 1x12 1x5 2x3 1x1 4x1; do
  for persist in 0 2; do
    echo -n "tile=$tile persist=$persist: "
    GME_SEA_TILE=$tile GME_SEA_PERSIST=$persist timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "$K" 2>&1 | tail -1
  done
done
for e4 in 0 1; do
  echo -n "E4=$e4 (one tile per workgroup): "
  GME_SEA_E4=$e4 GME_SEA_PERSIST=0 timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "$K" 2>&1 | tail -1
done
