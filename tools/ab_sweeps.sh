#!/bin/bash
# Run ON THE GPU BOX: parity tests + a short bench for the sweeps-kernel variants (A/B), results under gpurun_out/<tag>_*.
#   bash tools/ab_sweeps.sh <tag> "<sweeps waves>" ...      e.g.  bash tools/ab_sweeps.sh r2c "group 1" "group 4" "tile 1"
TAG=${1:-ab}; shift
O=gpurun_out
[ $# -eq 0 ] && set -- "group 1" "group 2" "group 4" "tile 1"
for cfg in "$@"; do
    set -- $cfg
    export EVM_SWEEPS=$1 EVM_G_WAVES=$2
    n=${TAG}_$1$2
    timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -x -q > $O/${n}_parity.log 2>&1
    echo "$n parity rc=$? $(tail -1 $O/${n}_parity.log)"
    timeout -k 10 90 python bench.py --no-cpu-baseline --steps 512 --warmup 32 > $O/${n}_bench.json 2> $O/${n}_bench.err || { echo "$n bench failed"; tail -3 $O/${n}_bench.err; exit 1; }
    python tools/show_bench.py $O/${n}_bench.json | head -1
done
if [ -f build/libevm_gstamps.so ]; then
    cp build/libevm_gstamps.so evomotion_amd/libevomotion_hip.so
    for w in 1 2 4; do echo "== stamps, waves $w"; EVM_SWEEPS=group EVM_G_WAVES=$w timeout -k 10 200 python tools/gstamps.py 2>&1 | grep -v amdgpu; done
fi
