"""Diagnostic (-DEVM_STAMPS3 build): mean cycles of one sweep entry by kind (hinge, fixed, slider, p2p, active contact, idle contact)."""
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
acc = np.zeros((6, 2)); K = 50
for k in range(K + 70):
    env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
    if k >= 70:
        st = (ctypes.c_ulonglong * (n // 64 * 16))()
        check(lib.evm_env_get_stamps(env._h, st))
        acc += np.array(st, dtype=np.uint64).reshape(-1, 8, 2)[:, :6].astype(np.float64).sum(axis=0)
for q, name in enumerate(["hinge", "fixed", "muscle (work)", "muscle (waits)", "contact(active)", "contact(idle)"]):
    print("%-16s %8.0f cycles/entry   %6.1f entries per tile-step" % (name, acc[q, 0] / max(acc[q, 1], 1), acc[q, 1] / K / (n // 64)))
