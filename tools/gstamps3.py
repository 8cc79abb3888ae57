"""Diagnostic (-DEVM_GSTAMPS3 build): where the cycles of the root body's hinge chain entry go (k_sweeps_g, g_hinge_chain): record +
body loads, the rows of the phases (one hinge visit each, the other lane groups masked), the hand-over of the shared body's six
deltas between phases (ds_bpermute), the stores.  Every stamp drains the LDS queue first, so the parts add up (and the entry is
a little slower than in the product build).  Build + run on the GPU box:
    make -C evomotion_amd/csrc stamps3 && cp build/libevm_gstamps3.so evomotion_amd/libevomotion_hip.so && python tools/gstamps3.py [self_collision]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from evomotion_amd import VecRobotWalk
from evomotion_amd._lib import lib, check
n = 4096
sc = int(sys.argv[1]) if len(sys.argv) > 1 else 0
env = VecRobotWalk(n, seed=1234, parameters={"self_collision": sc})
env.reset()
env.stagger_episodes()
g = torch.Generator(device="cuda"); g.manual_seed(0)
for k in range(200):
    env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
st = (ctypes.c_ulonglong * (n // 64 * 16))()
check(lib.evm_env_get_stamps(env._h, st))
a = np.array(st, dtype=np.uint64)[:12].astype(np.float64)
e, ph = a[4], a[5]
print("self_collision=%d: hinge chain entries %.0f, phases per entry %.2f" % (sc, e, ph / e))
print("per entry: loads %.0f  rows %.0f (%.0f per phase = one hinge visit)  hand-over %.0f (%.0f per phase)  stores %.0f  sum %.0f cycles"
      % (a[0] / e, a[1] / e, a[1] / ph, a[2] / e, a[2] / ph, a[3] / e, a[:4].sum() / e))
print("around it, per hinge chain entry: decode + wait for versions %.0f, the chain function %.0f, publish %.0f; from the end of the previous chain entry of the wave to this one's start %.0f (p2p chain entries are stamped too: they set the 'previous end')"
      % (a[8] / e, a[9] / e, a[10] / e, a[11] / e))
