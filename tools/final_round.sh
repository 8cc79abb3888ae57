#!/bin/bash
# Run ON THE GPU BOX: the round's evidence in one call — GPU tests, both profile sets (kernel statistics + PMC passes + FETCH_SIZE
# calibration), the bench lines of every mode, the narrowphase work statistics.   bash tools/final_round.sh <tag>
set -u
TAG=${1:-r4z}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/${TAG}_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -1 $O/${TAG}_gputests.log
timeout -k 10 500 bash tools/profile_round.sh $TAG || echo "profile $TAG failed"
BENCH_EXTRA="--self-collision 0" timeout -k 10 500 bash tools/profile_round.sh ${TAG}0 || echo "profile ${TAG}0 failed"
cd $R
# summaries on the box (the raw per-dispatch counter files are too big to travel back): profiles/<tag>* -> gpurun_out/summary/
python tools/pmc_summary.py $TAG k_sweeps_g "k_split_pairs_rec<7>" "k_split_pre_a<7>" > /dev/null && python tools/traffic_json.py $TAG 1 > /dev/null
python tools/pmc_summary.py ${TAG}0 k_sweeps_g "k_split_pre_b<7>" "k_split_pre_a<7>" > /dev/null && python tools/traffic_json.py ${TAG}0 0 > /dev/null
mkdir -p $O/summary && cp profiles/${TAG}_* profiles/${TAG}0_* $O/summary/ 2>/dev/null
for t in $TAG ${TAG}0; do for d in $O/prof_${t}_pmc* $O/prof_${t}_calib $O/prof_${t}_stats; do find $d -name '*counter_collection.csv' -delete 2>/dev/null; find $d -name '*kernel_trace.csv' -delete 2>/dev/null; done; done
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench20.json 2> $O/${TAG}_bench20.err; echo "bench20 rc=$?"
timeout -k 10 200 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
timeout -k 10 100 python bench.py --self-collision 0 --no-cpu-baseline > $O/${TAG}_bench_sc0.json 2>/dev/null; echo "bench sc0 rc=$?"
timeout -k 10 200 python bench.py --mode ppo --no-cpu-baseline > $O/${TAG}_bench_ppo_update.json 2>/dev/null; echo "bench ppo rc=$?"
timeout -k 10 200 python bench.py --mode sac --no-cpu-baseline > $O/${TAG}_bench_sac_update.json 2>/dev/null; echo "bench sac rc=$?"
timeout -k 10 300 python tools/pair_stats.py --steps 64 --trained 250 > $O/${TAG}_pair_stats.txt 2>&1; tail -3 $O/${TAG}_pair_stats.txt
echo final_round done
