# usage (GPU box): bash tools/r03_trace_stream.sh <tag> "<chunk,lanes> ..."  -- link utilisation of StreamEstimator runs from a rocprofv3 trace
TAG=${1:-r03k}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cfg in ${2:-128,2}; do
  C=${cfg%%,*}; L=${cfg##*,}
  O=$ROOT/gpurun_out/$TAG/c${C}_l${L}; mkdir -p $O
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $ROOT/tools/stream_gme_trace.py $C $L > $O/trace.log 2>&1 || { echo trace failed; tail -5 $O/trace.log; }
  python3 $ROOT/tools/stream_trace_report.py $O/trace "chunk $C lanes $L"
done
