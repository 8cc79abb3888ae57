#!/bin/bash
# Run ON THE GPU BOX: GPU tests + the bench lines of every mode + kernel statistics of the training benches (tag r4u).
set -u
TAG=${1:-r4u}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/${TAG}_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -1 $O/${TAG}_gputests.log
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench20.json 2> $O/${TAG}_bench20.err; echo "bench20 rc=$?"
timeout -k 10 200 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
timeout -k 10 100 python bench.py --self-collision 0 --no-cpu-baseline > $O/${TAG}_bench_sc0.json 2>/dev/null; echo "bench sc0 rc=$?"
timeout -k 10 200 python bench.py --mode ppo --no-cpu-baseline > $O/${TAG}_bench_ppo_update.json 2>/dev/null; echo "bench ppo rc=$?"
timeout -k 10 200 python bench.py --mode sac --no-cpu-baseline > $O/${TAG}_bench_sac_update.json 2>/dev/null; echo "bench sac rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_sac_stats -o run -- python3 $R/bench.py --mode sac --steps 128 --warmup 32 --no-cpu-baseline > $O/prof_${TAG}_sac_stats.log 2>&1 || echo "sac stats failed"
f=$(find $O/prof_${TAG}_sac_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_sac_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_ppo_stats -o run -- python3 $R/bench.py --mode ppo --steps 128 --warmup 32 --no-cpu-baseline > $O/prof_${TAG}_ppo_stats.log 2>&1 || echo "ppo stats failed"
f=$(find $O/prof_${TAG}_ppo_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_ppo_kernel_stats.csv
for d in $O/prof_${TAG}_sac_stats $O/prof_${TAG}_ppo_stats; do find $d -name '*kernel_trace.csv' -delete 2>/dev/null; done
cd $R
python tools/show_bench.py $O/${TAG}_bench20.json $O/${TAG}_bench.json $O/${TAG}_bench_sc0.json $O/${TAG}_bench_ppo_update.json $O/${TAG}_bench_sac_update.json 2>/dev/null
echo final_lines done
