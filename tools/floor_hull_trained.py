"""tests/diag/floor_hull_diff.py in the regime a trained policy visits (GPU box): PPO for --updates updates on 4096 envs, then
--steps steps of --envs environments with the policy's own actions; every visited state is stepped once by the CPU oracle with the
plane shortcut and once with the floor as a hull pair (three variants: a well-conditioned 16 m box, the reference's 2000 m box, the
same with the libccd-derived pre-test), and the one-step differences are reported.  The GPU only supplies states and actions.
   python tools/floor_hull_trained.py [--updates 250] [--envs 12] [--steps 150]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "diag"))
import floor_hull_diff as fh  # noqa: E402
import orc  # noqa: E402
from evomotion_amd import VecPpoGaeAgent, VecRobotWalk  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--updates", type=int, default=250)
ap.add_argument("--envs", type=int, default=12)
ap.add_argument("--steps", type=int, default=150)
a = ap.parse_args()

env = VecRobotWalk(4096, seed=77, device=0)
env.reset(); env.stagger_episodes()
agent = VecPpoGaeAgent(5, [env.state_dim], [env.action_dim], hidden_size=256, device=0, horizon=32, epoch=8, learning_rate=3e-4)
lengths = []
for u in range(a.updates):
    b = agent.rollout(env)
    agent.update()
    if u >= a.updates - 20:
        m = b["valid_u8"] == 1
        lengths.append(float(m.sum()) / max(float((b["done_u8"][m] != 0).sum()), 1.0))
print("trained %d updates: mean episode length %.0f steps" % (a.updates, np.mean(lengths)))

# the states and actions of the first --envs environments over --steps policy steps (rollout form: resets included, skipped below)
samples = []
obs = env.obs.clone()
for k in range(a.steps):
    S = env.get_state()[: a.envs].copy()
    act, _, _ = agent.fused.forward(obs, seed=9000 + k)
    st = env.step_autoreset(act)
    valid = st.valid[: a.envs].cpu().numpy()
    ah = act[: a.envs].cpu().numpy()
    for i in range(a.envs):
        if valid[i] == 1:                      # a do_step transition (not a settle call)
            samples.append((S[i], ah[i].copy()))
    obs = st.state.clone()
print("%d (state, action) samples" % len(samples))

L = orc.load()
o = orc.OracleEnv(seed=1, lib=L, self_collision=1)
o.reset()
for half, pre, what in ((8.0, 0, "a 16 m box kept under the member (the algorithmic difference alone)"),
                        (1000.0, 0, "the reference's 2000 m box"),
                        (1000.0, 1, "the reference's 2000 m box + the libccd-derived pre-test of bullet3 >= 2.88")):
    acc = fh.Acc()
    for S, act in samples:
        fh.one_step(o, S, act, acc, L, half, pre)
    fh.report(acc, "floor as a hull pair [%s] vs the plane shortcut, one step from identical state, TRAINED regime" % what)
