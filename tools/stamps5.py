"""Diagnostic (-DEVM_STAMPS5 build): longest item of each kind in k_split_pre_b, per tile (cycles)."""
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
acc = []
for k in range(120):
    if k >= 70:
        torch.cuda.synchronize()
        zero = (ctypes.c_ulonglong * (n // 64 * 16))()
        import ctypes as C
        # stamps are max-accumulated: clear them before the step
        torch.cuda.synchronize()
    env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
    if k >= 70:
        st = (ctypes.c_ulonglong * (n // 64 * 16))()
        check(lib.evm_env_get_stamps(env._h, st))
        acc.append(np.array(st, dtype=np.uint64).reshape(-1, 16)[:, :8].astype(np.float64))
a = np.stack(acc)[-1]  # max since creation (never cleared): upper envelope
for q, name in enumerate(["hinge", "fixed", "slider", "p2p", "member (cube hull)", "member (foot hull)", "  manifold update", "  contact-row setup"]):
    print("%-20s longest item: median over tiles %8.0f cycles, max %8.0f" % (name, np.median(a[:, q]), a[:, q].max()))
