"""Per step over 400 random-action rollout steps of 4096 envs: penetration-solver queries, how many the urgent list predicted, and what the
speculation blocks did (runs, answers used, waits that ran out).  GPU box: python tools/spec_stats.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evomotion_amd import VecRobotWalk
n = 4096
env = VecRobotWalk(n, seed=1234); env.reset(); env.stagger_episodes()
g = torch.Generator(device="cuda"); g.manual_seed(0)
for k in range(200): env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
env.penetration_queries(); env.speculation_counters()
K = 400
for k in range(K): env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
q = env.penetration_queries(); s = env.speculation_counters()
print("per step: solver queries %.2f, predicted %.2f, urgent entries %.1f; speculation runs %.2f, used %.2f, expired %.3f" % (q / K, env.predicted_penetration_queries / K, env.urgent_entries / K, s[0] / K, s[1] / K, s[2] / K))
