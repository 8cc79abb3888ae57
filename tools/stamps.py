"""Diagnostic: per-phase share of the step kernel from s_memtime stamps (needs a -DEVM_STAMPS build copied over
evomotion_amd/libevomotion_hip.so).  Shares only; the stamped build's own run time is not a measurement."""
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
acc = np.zeros(9)
names = ["bodies", "collide", "joint setup", "contact setup", "10 sweeps", "readback", "integrate", "observe"]
K = 100
for k in range(K + 70):
    a = torch.rand(n, 12, device="cuda", generator=g) * 2 - 1
    env.step_autoreset(a)
    if k >= 70:
        st = (ctypes.c_ulonglong * (n // 64 * 16))()
        check(lib.evm_env_get_stamps(env._h, st))
        s = np.array(st, dtype=np.uint64).reshape(-1, 16).astype(np.float64)
        d = np.diff(s[:, :9], axis=1)
        d[d < 0] = 0
        acc[:8] += np.median(d, axis=0)
tot = acc[:8].sum()
for nm, v in zip(names, acc[:8]):
    print("%-14s %9.0f cycles %5.1f %%" % (nm, v / K, 100 * v / tot))
print("total cycles/step", tot / K, "(s_memtime counts shader-clock cycles, ~2.0 GHz under this load)")
# fine stamps inside "collide" (wave 0 of each tile): own scans, all scans, own manifolds
st = (ctypes.c_ulonglong * (n // 64 * 16))()
check(lib.evm_env_get_stamps(env._h, st))
s = np.array(st, dtype=np.uint64).reshape(-1, 16).astype(np.float64)
print("collide detail (median over tiles, last step): wave0 scans %.0f, wait for all scans %.0f, wave0 manifolds %.0f, wait for all %.0f cycles"
      % (np.median(s[:, 9] - s[:, 1]), np.median(s[:, 10] - s[:, 9]), np.median(s[:, 11] - s[:, 10]), np.median(s[:, 2] - s[:, 11])))
