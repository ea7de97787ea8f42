# usage: bash tools/ab.sh "<lib names under tools/microbench without libgme_ prefix>" "<bench configs>" [rounds] [extra bench args]
# Same-box A/B of library builds (device-to-device variance is ~12 %, so only compare inside one call).
# A config may carry a content after a colon: "exh720:pan240x2" = --config exh720 --content pan240x2.
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=global-motion-estimation_amd/lib/libgme_hip.so
cp $L /tmp/keep.so
trap 'cp /tmp/keep.so $L' EXIT      # the tree's own library comes back whatever happens to a variant
for r in $(seq 1 ${3:-2}); do
for v in $1; do
  cp tools/microbench/libgme_$v.so $L
  for cc in $2; do
    c=${cc%%:*}; content=""; [ "$cc" != "$c" ] && content="--content ${cc#*:}"
    echo -n "$v $cc $4 "; timeout -k 10 200 python3 bench.py --config $c $content --no-cpu-baseline --no-pcie --no-content-sweep --no-secondary $4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('elimination',{})
print(round(d['value']), d['parity']['ok'], 'surviving', e.get('surviving_fraction'), 'listed', e.get('listed_fraction_before_ordered_rounds'), 'redo', e.get('tiles_redone_by_brute_force'))"
  done
done
done
cp /tmp/keep.so $L
