"""Where the penetration-solver queries of a rollout come from (GPU box, a library built with `make -C evomotion_amd/csrc EXTRA=-DEVM_DIAG_PEN`):
by origin (reset starting, pending, flagged by the previous step, other) and, for the unflagged ones, by the settle step of the episode's start
they fall in.  python tools/diag_pen.py"""
import sys, ctypes, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evomotion_amd import VecRobotWalk
from evomotion_amd._lib import lib, check
n = 4096
env = VecRobotWalk(n, seed=1234); env.reset(); env.stagger_episodes()
g = torch.Generator(device="cuda"); g.manual_seed(0)
for k in range(200): env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
out = (ctypes.c_int * 30)()
check(lib.evm_env_get_speculation_counters(env._h, out, 1, None)); env.penetration_queries(); env.clear_stats()
K = 400
for k in range(K): env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
check(lib.evm_env_get_speculation_counters(env._h, out, 1, None))
q = env.penetration_queries()
print("solver queries/step %.2f; spec runs %.2f used %.2f unusable %.2f; by origin: reset starting %.2f, pending %.2f, flagged %.2f, other with points %.2f, other without %.2f; in settle steps %.2f" % ((q / K,) + tuple(out[i] / K for i in range(9))))
print("unflagged solver queries by settle-step index (settle_steps - settle_left):", [round(out[9 + i] / K, 3) for i in range(10)])
print("big-hull queries per step by GJK iterations [0-3,4-7,...,28+]:", [round(out[19 + i] / K, 1) for i in range(8)], "; >= 12 iterations: %.1f per step, of them predicted by the previous step (>= 10 there): %.1f; predicted in all: %.1f" % (out[27] / K, out[28] / K, out[29] / K))
print(env.stats(), "per step resets %.1f" % (env.stats()["resets"] / K))
