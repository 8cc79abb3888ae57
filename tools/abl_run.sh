#!/bin/bash
# Run ON THE GPU BOX: kernel stats of the PPO-mode bench for each ablation build under build/abl/ (scratch experiment).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cp $R/evomotion_amd/libevomotion_hip.so /tmp/lib_keep.so
cd /tmp && export TMPDIR=/tmp
for lib in $R/build/abl/libabl_*.so; do
    v=$(basename $lib .so)
    cp $lib $R/evomotion_amd/libevomotion_hip.so
    timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl_$v -o run -- python3 $R/bench.py --mode ppo --steps 64 --warmup 32 --no-cpu-baseline > $O/abl_$v.log 2>&1 || { echo "$v failed"; tail -3 $O/abl_$v.log; }
    echo "== $v"; grep -E "k_ppo_(forward|backward|wgrad<)" $O/abl_$v/run_kernel_stats.csv | awk -F, '{print $1, $2, $4}' | cut -c1-120
done
cp /tmp/lib_keep.so $R/evomotion_amd/libevomotion_hip.so
