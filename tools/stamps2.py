"""Diagnostic (-DEVM_STAMPS2 build): per-wave time spent waiting on body versions vs solving joint visits in the 10 sweeps."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from evomotion_amd import VecRobotWalk
from evomotion_amd._lib import lib, check
n = 4096
env = VecRobotWalk(n, seed=1234)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(0)
acc = np.zeros((8, 2)); K = 50
for k in range(K + 70):
    env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
    if k >= 70:
        st = (ctypes.c_ulonglong * (n // 64 * 16))()
        check(lib.evm_env_get_stamps(env._h, st))
        acc += np.median(np.array(st, dtype=np.uint64).reshape(-1, 8, 2).astype(np.float64), axis=0)
for w in range(8):
    print("wave %d: wait %8.0f  solve %8.0f cycles per step (joint visits of 10 sweeps)" % (w, acc[w, 0] / K, acc[w, 1] / K))
