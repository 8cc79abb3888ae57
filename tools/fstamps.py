"""Diagnostic (-DEVM_FSTAMPS build): phase timeline of k_ppo_forward (131072 rows, both networks), every wave's s_memtime stamps.
    make -C evomotion_amd/csrc fstamps && cp build/libevm_fstamps.so evomotion_amd/libevomotion_hip.so && python tools/fstamps.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from evomotion_amd import VecRobotWalk, VecPpoGaeAgent
from evomotion_amd._lib import lib
env = VecRobotWalk(4096, seed=1, device=0); env.reset()
agent = VecPpoGaeAgent(1234, [env.state_dim], [env.action_dim], hidden_size=256, device=0, horizon=32, epoch=8, learning_rate=1e-3)
for _ in range(2):
    agent.rollout(env); agent.update()
torch.cuda.synchronize()
buf = np.zeros(8192 * 4 * 8, np.uint64)
lib.evm_debug_fstamps.argtypes = [ctypes.c_void_p]
assert lib.evm_debug_fstamps(buf.ctypes.data) == 0
t = buf.reshape(8192, 4, 8)
st = t[:, :, :7].astype(np.int64); hw = t[:, :, 7]
t0 = st[:, :, 0][st[:, :, 0] > 0].min()
d = np.diff(st, axis=2)
names = ["stage", "L1", "E1", "L2", "E2", "heads"]
print("kernel span %.1f us (2.4 GHz assumed)" % ((st[:, :, 6].max() - t0) / 2400.0))
print("phase medians (cycles):", {n: int(np.median(d[:, :, i])) for i, n in enumerate(names)})
print("phase means   (cycles):", {n: int(d[:, :, i].mean()) for i, n in enumerate(names)})
print("workgroup duration median %d, mean %d" % (np.median(st[:, 0, 6] - st[:, 0, 0]), (st[:, 0, 6] - st[:, 0, 0]).mean()))
# one CU's timeline: all waves on (xcc, se, cu) of workgroup 0 wave 0, SIMD 0
key = lambda h: ((h >> 32) & 15, (h >> 13) & 7, (h >> 8) & 15)
k0 = key(int(hw[0, 0]))
rows = []
for wg in range(8192):
    for w in range(4):
        h = int(hw[wg, w])
        if key(h) == k0 and ((h >> 4) & 3) == ((int(hw[0, 0]) >> 4) & 3):
            rows.append((int(st[wg, w, 0] - t0), wg, w, h & 15, [int(x) for x in st[wg, w] - t0]))
rows.sort()
print("timeline of one SIMD (start, wg, wave, slot, stamps relative to kernel start):")
for r in rows[:24]:
    print("  ", r)
