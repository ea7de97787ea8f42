# usage: bash tools/collect_profiles.sh <tag>   -- copies the judged summaries of gpurun_out/<tag>/ into profiles/
TAG=$1
for f in gpurun_out/$TAG/${TAG}_*_bench.json gpurun_out/$TAG/${TAG}_*_kernel_stats.csv gpurun_out/$TAG/${TAG}_*_pmc_summary.txt; do
  [ -s "$f" ] && cp "$f" profiles/
done
ls profiles | grep "^$TAG" | wc -l
