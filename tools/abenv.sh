# usage: bash tools/abenv.sh "<variants>" "<bench configs>" [rounds] [extra bench args]
# Same-box A/B of runtime switches of one build.  A variant is '-' (no switch) or comma-separated
# assignments, e.g. bash tools/abenv.sh "- GME_SEA_PERSIST=2 GME_SEA_PERSIST=2,GME_SEA_NB=8" "exh720"
set -e
cd /root/repo
for r in $(seq 1 ${3:-2}); do
for v in $1; do
  for c in $2; do
    echo -n "$v $c $4 "
    if [ "$v" = "-" ]; then timeout -k 10 200 python3 bench.py --config $c --no-cpu-baseline $4 2>/dev/null > /tmp/o.json; else env ${v//,/ } timeout -k 10 200 python3 bench.py --config $c --no-cpu-baseline $4 2>/dev/null > /tmp/o.json; fi
    python3 -c "
import json
d=json.loads(open('/tmp/o.json').read()); print(round(d['value']))"
  done
done
done
