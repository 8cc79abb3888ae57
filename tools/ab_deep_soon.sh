#!/bin/bash
# Run ON THE GPU BOX: narrowphase kernel time and speculation hit rate for several values of EVM_DEEP_SOON (scheduling threshold)
#   bash tools/ab_deep_soon.sh "-0.06 -0.03 0.0"
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in ${1:--0.06 -0.03 0.0}; do
  export EVM_DEEP_SOON=$v
  echo "== EVM_DEEP_SOON=$v"
  python3 $R/tools/spec_stats.py || exit 1
  rm -rf $O/prof_ab
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ab -o run -- python3 $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline --self-collision 1 > $O/prof_ab.log 2>&1 || exit 1
  f=$(find $O/prof_ab -name '*kernel_stats.csv' | head -1)
  grep -E "k_split_pairs_rec<7>|k_sweeps_g|k_split_pre_a" $f | cut -d, -f1-4,6,7 | cut -c1-160
  grep -o '"value": [0-9.]*' $O/prof_ab.log | head -1
done
rm -rf $O/prof_ab
