# usage: bash tools/abenv.sh "<variants>" "<bench configs>" [rounds] [extra bench args]
# Same-box A/B of runtime switches of one build.  A variant is '-' (no switch) or comma-separated
# assignments, e.g. bash tools/abenv.sh "- GME_SEA_PERSIST=2 GME_SEA_PERSIST=2,GME_SEA_NB=8" "exh720"
# A config may carry a content after a colon: "exh720:pan240x2" = --config exh720 --content pan240x2.
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 ${3:-2}); do
for v in $1; do
  for cc in $2; do
    c=${cc%%:*}; content=""; [ "$cc" != "$c" ] && content="--content ${cc#*:}"
    echo -n "$v $cc $4 "
    if [ "$v" = "-" ]; then timeout -k 10 200 python3 bench.py --config $c $content --no-cpu-baseline --no-secondary --no-pcie --no-content-sweep $4 2>/dev/null > /tmp/o.json; else env ${v//,/ } timeout -k 10 200 python3 bench.py --config $c $content --no-cpu-baseline --no-secondary --no-pcie --no-content-sweep $4 2>/dev/null > /tmp/o.json; fi
    python3 -c "
import json
d=json.loads(open('/tmp/o.json').read()); e=d.get('elimination',{})
print(round(d['value']), 'parity', d['parity']['ok'], 'surviving', round(e.get('surviving_fraction',0),4), 'listed', round(e.get('listed_fraction_before_ordered_rounds',0),4), 'redo', e.get('tiles_redone_by_brute_force'))"
  done
done
done
