#!/bin/bash
# Run ON THE GPU BOX: PMC passes over the PPO-mode bench (scratch experiment)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
    i=$((i + 1))
    timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/ablpmc$i -o run -- python3 $R/bench.py --mode ppo --steps 64 --warmup 32 --no-cpu-baseline > $O/ablpmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $O/ablpmc$i.log; }
done
ls $O/ablpmc1 $O/ablpmc2
