# usage: bash tools/ab_trace.sh "<variants>" <kernel regex> <bench args...>
# Same-box per-kernel timing of library variants (tools/build_variant.sh): rocprofv3 kernel stats of one bench run each.
V=$1; K=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
L=$ROOT/global-motion-estimation_amd/lib/libgme_hip.so
cp $L /tmp/keep.so
cd /tmp && export TMPDIR=/tmp
for v in $V; do
  cp $ROOT/tools/microbench/libgme_$v.so $L
  rm -rf /tmp/abt_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abt_$v -- python3 $ROOT/bench.py --no-cpu-baseline --no-pcie --no-content-sweep "$@" > /tmp/abt_$v.log 2>&1 || echo "$v: run failed"
  python3 - <<PY
import csv,glob,re
for f in glob.glob("/tmp/abt_$v/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if re.search(r"$K", r["Name"]):
            n=re.search(r"(k_\w+(<\d+>)?)",r["Name"]).group(1)
            print("%-6s %-22s calls=%-4s avg_us=%9.1f min_us=%9.1f max_us=%9.1f"%("$v",n,r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
PY
done
cp /tmp/keep.so $L
