#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats + PMC passes of bench.py, raw output under gpurun_out/.
#   bash tools/profile_round.sh <tag> [dynamics|ppo]
# Counters are collected in their own passes (--pmc only, one group per pass); the program follows `--` directly.
# Summaries for profiles/ are made afterwards with tools/pmc_summary.py.
set -u
TAG=${1:-r1}; MODE=${2:-dynamics}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
EXTRA=""; [ "$MODE" = ppo ] && EXTRA="--mode ppo --no-update"
[ -n "${ENVS:-}" ] && EXTRA="$EXTRA --envs $ENVS"   # default: 4096 envs (BASELINE configs[1])
[ -n "${BENCH_EXTRA:-}" ] && EXTRA="$EXTRA $BENCH_EXTRA"   # e.g. BENCH_EXTRA="--self-collision 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats -o run -- python3 $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline $EXTRA > $O/prof_${TAG}_stats.log 2>&1 || exit 1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d $O/prof_${TAG}_pmc$i -o run -- python3 $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline $EXTRA > $O/prof_${TAG}_pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
# FETCH_SIZE calibration on this kernel's access widths (guide: only 16 B/lane streaming reads are calibrated)
if [ -x $R/tools/calib_fetch ]; then
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_${TAG}_calib -o run -- $R/tools/calib_fetch > $O/prof_${TAG}_calib.log 2>&1 || echo "calibration failed"
fi
echo done
