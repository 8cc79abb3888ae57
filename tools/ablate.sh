#!/bin/bash
# usage: tools/ablate.sh <alt .so>   -- bench with an alternative build of the library (timing only)
cp evomotion_amd/libevomotion_hip.so /tmp/lib_orig.so
cp "$1" evomotion_amd/libevomotion_hip.so
python bench.py --steps 256 --warmup 64 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'launch_ms', d['roofline']['launch_ms'], 'env-steps/s', d['value'])"
cp /tmp/lib_orig.so evomotion_amd/libevomotion_hip.so
