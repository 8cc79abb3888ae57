#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats + PMC passes of the PPO-training bench (update kernels included),
# the MFMA / VALU co-execution and store probes.  Raw output under gpurun_out/; summaries with tools/pmc_summary.py <tag>.
set -u
TAG=${1:-r2c_ppo}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats -o run -- python3 $R/bench.py --mode ppo --steps 256 --warmup 64 --no-cpu-baseline > $O/prof_${TAG}_stats.log 2>&1 || exit 1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
    i=$((i + 1))
    timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/prof_${TAG}_pmc$i -o run -- python3 $R/bench.py --mode ppo --steps 64 --warmup 32 --no-cpu-baseline > $O/prof_${TAG}_pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
cd $R
[ -x tools/coexec_probe ] && timeout -k 10 60 ./tools/coexec_probe > $O/${TAG}_coexec_probe.txt 2>&1
[ -x tools/store_probe ] && timeout -k 10 60 ./tools/store_probe > $O/${TAG}_store_probe.txt 2>&1
echo done
