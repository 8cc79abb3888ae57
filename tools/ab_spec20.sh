#!/bin/bash
# Run ON THE GPU BOX: the 20-step form the driver records (bench.py --gpus 1 --steps 20 --warmup 5), speculation blocks on / off, three runs each
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2 3; do
for v in 1 0; do
  EVM_SPECULATE=$v python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spec=$v', round(d['value']), d['ms_per_step'], d['roofline']['launch_ms'])"
done; done
