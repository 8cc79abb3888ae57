"""Diagnostic (-DEVM_KSTAMPS build of the library): the working wavefronts of the narrowphase kernel k_split_pairs — how many, how
long each runs (s_memtime cycles: mean and longest), and the extent of the kernel from the first working wave's start to the last
one's end (s_memrealtime, 100 MHz).  Build + run on the GPU box:
    make -C evomotion_amd/csrc kstamps && cp build/libevm_kstamps.so evomotion_amd/libevomotion_hip.so && python tools/kstamps.py"""
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
rows = []
for k in range(K + 200):
    if k == 199:
        check(lib.evm_env_get_stamps(env._h, (ctypes.c_ulonglong * (n // 64 * 16))()))  # (a read resets the accumulators)
    env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
    if k >= 200:
        st = (ctypes.c_ulonglong * (n // 64 * 16))()
        check(lib.evm_env_get_stamps(env._h, st))
        rows.append(np.array(st, dtype=np.uint64)[:64].astype(np.float64))
r = np.array(rows)
for kind, name in ((0, "big-hull waves (4 queries of 16 lanes)"), (1, "small-hull waves (one env per lane)")):
    print("%-40s %7.1f per step, mean %8.0f cycles, longest %8.0f cycles" % (name, r[:, 3 * kind + 1].mean(), (r[:, 3 * kind] / np.maximum(r[:, 3 * kind + 1], 1)).mean(), r[:, 3 * kind + 2].mean()))
print("extent first start -> last end: %.1f us" % ((r[:, 7] - r[:, 6]).mean() / 100.0))
for kind, name in ((0, "big-hull"), (1, "small-hull")):
    w = np.maximum(r[:, 3 * kind + 1], 1)
    ph = [(r[:, 8 + 4 * kind + k] / w).mean() for k in range(4)]
    print("%-10s per working wave: set-up + manifold loads %.0f, closest points (GJK) %.0f, refresh + manifold stores %.0f, contact record %.0f cycles%s"
          % (name, ph[0], ph[1], ph[2], ph[3], (", hull staging into LDS %.0f" % (r[:, 16] / w).mean()) if kind == 0 else ""))
w = r[:, 1] + r[:, 4]
print("queries that took the penetration branch: %.2f per step in %.2f wavefronts; GJK iterations of a wave's slowest lane: %.1f on average"
      % (r[:, 17].mean(), r[:, 18].mean(), (r[:, 19] / np.maximum(w, 1)).mean()))
for kind, name in ((0, "big-hull"), (1, "small-hull")):
    it = np.maximum(r[:, 22 + 4 * kind], 1)
    print("%-10s GJK loop trips %.1f per step (%.1f per working wave): supports (Minkowski difference) %.0f, rest of the iteration (simplex, exits) %.0f cycles per trip"
          % (name, it.mean(), (it / np.maximum(r[:, 3 * kind + 1], 1)).mean(), (r[:, 20 + 4 * kind] / it).mean(), (r[:, 21 + 4 * kind] / it).mean()))

c = np.maximum(r[:, 32].sum(), 1)
print("penetration queries (EPA): %.2f per step, %.2f guess vectors each; per query: its GJK %.0f cycles, EPA %.0f cycles in %.1f rounds" % (r[:, 32].mean(), r[:, 42].sum() / c, r[:, 34].sum() / c, r[:, 35].sum() / c, r[:, 36].sum() / c))
it = np.maximum(r[:, 36].sum(), 1)
print("  per EPA round: support point %.0f, visibility of the faces %.0f, horizon walk %.0f, new faces %.0f, findbest %.0f cycles" % tuple(r[:, k].sum() / it for k in (37, 38, 39, 40, 41)))

u = np.maximum(r[:, 47].sum(), 1)
names = ["block start -> entry (hull staging)", "set-up (transforms, boxes)", "Voronoi GJK", "EPA's own GJK", "EPA rounds", "witnesses, normal check", "manifold refresh + stores", "contact record"]
print("urgent-list queries that took the penetration solver: %.2f per step; cycles of one such query by way-point:" % r[:, 47].mean())
for k, nm in enumerate(names):
    print("   %-40s %8.0f" % (nm, r[:, 48 + k].sum() / u))
print("   %-40s %8.0f" % ("total", r[:, 48:56].sum() / u))
print("   EPA rounds by phase (sums over the query's rounds): support point %.0f, visibility %.0f, horizon walk %.0f, new faces %.0f, findbest + register reload %.0f"
      % tuple(r[:, 56 + k].sum() / u for k in range(5)))
print("   hull scans with margins (EPA's GJK + rounds + witnesses): %.1f per query; %.0f cycles per scan of A's hull, %.0f of B's"
      % (r[:, 63].sum() / u, 2 * r[:, 61].sum() / max(r[:, 63].sum(), 1), 2 * r[:, 62].sum() / max(r[:, 63].sum(), 1)))
