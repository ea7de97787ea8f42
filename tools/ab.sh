# usage: bash tools/ab.sh "<lib names under tools/microbench without libgme_ prefix>" "<bench configs>" [rounds] [extra bench args]
# Same-box A/B of library builds (device-to-device variance is ~12 %, so only compare inside one call).
set -e
cd /root/repo
L=global-motion-estimation_amd/lib/libgme_hip.so
cp $L /tmp/keep.so
trap 'cp /tmp/keep.so $L' EXIT      # the tree's own library comes back whatever happens to a variant
for r in $(seq 1 ${3:-2}); do
for v in $1; do
  cp tools/microbench/libgme_$v.so $L
  for c in $2; do
    echo -n "$v $c $4 "; timeout -k 10 200 python3 bench.py --config $c --no-cpu-baseline --no-pcie --no-content-sweep $4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), d['parity']['ok'], d.get('elimination',{}).get('surviving_fraction'), d.get('elimination',{}).get('tiles_redone_by_brute_force'))"
  done
done
done
cp /tmp/keep.so $L
