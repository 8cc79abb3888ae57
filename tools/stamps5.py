"""Diagnostic (-DEVM_STAMPS5 build): longest item of each kind in k_split_pre_b and k_split_post, cycles (max over the run, median over tiles).
    hipcc ... -DEVM_STAMPS5 -> build/libevm_stamps5.so; cp over evomotion_amd/libevomotion_hip.so; python tools/stamps5.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from evomotion_amd import VecRobotWalk
from evomotion_amd._lib import lib, check
n = 4096
env = VecRobotWalk(n, seed=1234)
env.reset()
env.stagger_episodes()
g = torch.Generator(device="cuda"); g.manual_seed(0)
for k in range(150):
    env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
st = (ctypes.c_ulonglong * (n // 64 * 16))()
check(lib.evm_env_get_stamps(env._h, st))
s = np.array(st, dtype=np.uint64).reshape(-1, 16).astype(np.float64)
for q, name in enumerate(["hinge item", "fixed item", "slider item", "p2p item", "member item (cube)", "member item (foot)", "  manifold update", "  contact-row setup", "post: attach sphere", "post: member", "post: root member"]):
    print("%-22s median of tiles' longest %8.0f   max %8.0f cycles" % (name, np.median(s[:, q]), s[:, q].max()))
