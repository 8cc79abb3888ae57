"""Diagnostic (-DEVM_GSTAMPS build of the library): cycles of the lane-group sweeps kernel by phase and by entry type
(wave 0 of the first quarter of every 64-env tile).  Build + run on the GPU box:
    make -C evomotion_amd/csrc stamps   (-> build/libevm_gstamps.so)
    cp build/libevm_gstamps.so evomotion_amd/libevomotion_hip.so && python tools/gstamps.py"""
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
K = 40
acc = np.zeros(16)
for k in range(K + 200):
    env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
    if k >= 200:
        st = (ctypes.c_ulonglong * (n // 64 * 16))()
        check(lib.evm_env_get_stamps(env._h, st))
        acc += np.array(st, dtype=np.uint64).reshape(-1, 16).astype(np.float64).mean(axis=0)
acc /= K
if "--waves" in sys.argv:  # -DEVM_GSTAMPS2 build: per wave
    for w in range(4):
        print("wave %d: sweeps phase %.0f cycles, waiting for versions %.0f (%.0f %%), %d entries, prologue %.0f" % (w, acc[4 * w], acc[4 * w + 1], 100 * acc[4 * w + 1] / max(acc[4 * w], 1), acc[4 * w + 2], acc[4 * w + 3]))
    sys.exit(0)
for q, name in enumerate(["hinge", "fixed", "slider", "p2p", "contact"]):
    if q == 2 and env.n_pairs:
        print("contact rounds alone (no barriers) %8.0f cycles per step" % acc[4])
        continue
    print("%-8s %7.0f cycles/entry  %6.1f entries per step (wave 0)  %8.0f cycles per step" % (name, acc[2 * q] / max(acc[2 * q + 1], 1), acc[2 * q + 1], acc[2 * q]))
if env.n_pairs:   # member-vs-member mode: slots 13..15 hold the contact set-up phases, the "contact" line is the contact rounds of a sweep
    print("prologue %.0f (of it: record image + bodies + barrier %.0f, contact program %.0f, owners' records %.0f, split-impulse recovery + warm start %.0f)  sweeps %.0f  epilogue %.0f cycles (wave 0); contact line = cycles per round-phase"
          % (acc[10], acc[5], acc[13], acc[14], acc[15], acc[11], acc[12]))
else:
    print("prologue %.0f (table %.0f, records %.0f, bodies %.0f, manifold counts + barrier %.0f)  sweeps %.0f  epilogue %.0f cycles (wave 0)"
          % (acc[10], acc[13], acc[14], acc[15], acc[10] - acc[13] - acc[14] - acc[15], acc[11], acc[12]))
