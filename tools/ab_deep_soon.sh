#!/bin/bash
# Run ON THE GPU BOX: narrowphase kernel time, prediction and speculation hit rates for several values of a scheduling threshold
#   bash tools/ab_deep_soon.sh EVM_DEEP_SOON "-0.08 -0.06 -0.045"      (distance below which a pair is flagged for the next step's urgent list)
#   bash tools/ab_deep_soon.sh EVM_GAP_SOON "0 -0.01 -0.02 -0.04"       (core-box separation below which a pair without cached points is urgent)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
VAR=${1:-EVM_DEEP_SOON}
cd /tmp && export TMPDIR=/tmp
for v in ${2:--0.06 -0.03}; do
  export $VAR=$v
  echo "== $VAR=$v"
  python3 $R/tools/spec_stats.py || exit 1
  rm -rf $O/prof_ab
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ab -o run -- python3 $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline --self-collision 1 > $O/prof_ab.log 2>&1 || exit 1
  f=$(find $O/prof_ab -name '*kernel_stats.csv' | head -1)
  grep -E "k_split_pairs_rec<7>|k_sweeps_g|k_split_pre_a<7>" $f | cut -d, -f1-4,6,7 | cut -c1-160
  grep -o '"value": [0-9.]*' $O/prof_ab.log | head -1
done
rm -rf $O/prof_ab
