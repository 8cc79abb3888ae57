#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel statistics only (no counters) of bench.py in both collision modes -> gpurun_out/<tag>[0]_kernel_stats.csv
#   bash tools/stats_only.sh <tag>
set -u
TAG=${1:-r3n}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  T=$TAG; [ $m = 0 ] && T=${TAG}0
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_stats -o run -- python3 $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline --self-collision $m > $O/prof_${T}_stats.log 2>&1 || exit 1
  f=$(find $O/prof_${T}_stats -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] || { echo "no stats file"; exit 1; }
  cp $f $O/${T}_kernel_stats.csv
  head -5 $O/${T}_kernel_stats.csv | cut -c1-150
done
