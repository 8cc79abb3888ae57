#!/bin/bash
# AddressSanitizer + UBSan run of the CPU oracle (sanitizers are CPU-only on this pool): rollouts with resets in both collision
# modes, state round trips and a deep-interpenetration scene (the penetration-depth solver).  Clean on 2026-10-04 (round 3).
#   bash tests/diag/oracle_sanitize.sh
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
SO=/tmp/liborc_asan.so
(cd "$ROOT/oracle" && g++ -std=c++17 -O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -shared orc_world.cpp orc_narrow.cpp orc_api.cpp -o $SO)
cat > /tmp/asan_run.py <<PY
import sys
sys.path.insert(0, '$ROOT/tests'); sys.path.insert(0, '$ROOT')
import orc
orc.ORC_LIB = '$SO'
import numpy as np
L = orc.load()
for mode in (0, 1):
    e = orc.OracleEnv(seed=7, lib=L, self_collision=mode)
    obs, r, d = e.reset()
    rng = np.random.default_rng(1)
    for k in range(400):
        if d: obs, r, d = e.reset()
        else: obs, r, d = e.do_step(rng.uniform(-1, 1, 12).astype(np.float32))
    s = e.get_state(); e.set_state(s)
    assert np.isfinite(obs).all()
    print("mode", mode, "ok", e.pair_stats() if mode else "")
import physics_cases as pc, pathlib, tempfile
from conftest import write_skeleton
tmp = pathlib.Path(tempfile.mkdtemp())
sk = write_skeleton(tmp / "two.skel", [dict(name="body", mass=4.0, scale=(0.5, 0.2, 0.5)), dict(name="other", mass=0.5, t=(0.0, 1.0, 0.0), scale=(0.3, 0.25, 0.2), shape="feet")])
ow = pc.OracleWorld(sk, lib=L, self_collision=1)
ow.set_state(pc.clean_state(ow, [[0.0, 3000.0, 0.0], [0.05, 3000.02, 0.01]])); ow.step(3)
print("penetration scene ok", ow.e.pair_stats())
PY
LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python /tmp/asan_run.py
