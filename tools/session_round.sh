#!/bin/bash
# Run ON THE GPU BOX: evidence of a build whose dynamics kernels are unchanged but whose agent-side kernels moved —
# GPU tests, rocprofv3 kernel statistics of the dynamics and PPO benches, PMC pass of the policy forward, the bench lines of every
# mode, the policy tile timings.   bash tools/session_round.sh <tag>
set -u
TAG=${1:-r4s}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/${TAG}_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -1 $O/${TAG}_gputests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats -o run -- python3 $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline > $O/prof_${TAG}_stats.log 2>&1 || echo "stats failed"
f=$(find $O/prof_${TAG}_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats.csv && head -8 $O/${TAG}_kernel_stats.csv | cut -c1-160
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_ppo_stats -o run -- python3 $R/bench.py --mode ppo --steps 128 --warmup 32 --no-cpu-baseline > $O/prof_${TAG}_ppo_stats.log 2>&1 || echo "ppo stats failed"
f=$(find $O/prof_${TAG}_ppo_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_ppo_kernel_stats.csv && head -14 $O/${TAG}_ppo_kernel_stats.csv | cut -c1-160
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_sac_stats -o run -- python3 $R/bench.py --mode sac --steps 128 --warmup 32 --no-cpu-baseline > $O/prof_${TAG}_sac_stats.log 2>&1 || echo "sac stats failed"
f=$(find $O/prof_${TAG}_sac_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_sac_kernel_stats.csv
# counters of the policy forward alone (its own passes, --pmc only)
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
    i=$((i + 1))
    timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $O/prof_${TAG}p_pmc$i -o run -- python3 $R/tools/policy_tiles.py 4096 > $O/prof_${TAG}p_pmc$i.log 2>&1 || echo "policy pmc pass $i failed"
done
for d in $O/prof_${TAG}_stats $O/prof_${TAG}_ppo_stats $O/prof_${TAG}_sac_stats; do find $d -name '*kernel_trace.csv' -delete 2>/dev/null; done
cd $R
python tools/pmc_summary.py ${TAG}p k_policy_forward16 "k_policy_forward(" | tail -3
mkdir -p $O/summary && cp profiles/${TAG}p_* $O/summary/ 2>/dev/null
for d in $O/prof_${TAG}p_pmc*; do find $d -name '*counter_collection.csv' -delete 2>/dev/null; done
timeout -k 10 100 python tools/policy_tiles.py 4096 2048 8192 > $O/${TAG}_policy_tiles.json 2>/dev/null; cat $O/${TAG}_policy_tiles.json
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench20.json 2> $O/${TAG}_bench20.err; echo "bench20 rc=$?"
timeout -k 10 200 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
timeout -k 10 100 python bench.py --self-collision 0 --no-cpu-baseline > $O/${TAG}_bench_sc0.json 2>/dev/null; echo "bench sc0 rc=$?"
timeout -k 10 200 python bench.py --mode ppo --no-cpu-baseline > $O/${TAG}_bench_ppo_update.json 2>/dev/null; echo "bench ppo rc=$?"
timeout -k 10 200 python bench.py --mode sac --no-cpu-baseline > $O/${TAG}_bench_sac_update.json 2>/dev/null; echo "bench sac rc=$?"
python tools/show_bench.py $O/${TAG}_bench20.json $O/${TAG}_bench.json $O/${TAG}_bench_sc0.json $O/${TAG}_bench_ppo_update.json $O/${TAG}_bench_sac_update.json 2>/dev/null
echo session_round done
