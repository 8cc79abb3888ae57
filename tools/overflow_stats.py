import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, blob
from evomotion_amd import VecRobotWalk, RandomAgent
env = VecRobotWalk(4096, seed=1234, device=0)
env.reset(); env.stagger_episodes()
ag = RandomAgent([env.action_dim], env.device, seed=1)
for i in range(192): env.step_autoreset(ag.act(env.obs))
npairs = env.n_pairs
f = blob.fields(env.n_bodies, env.n_members, env.n_muscles, npairs)
cnt = []; mx = []
for k in range(60):
    for j in range(5): env.step_autoreset(ag.act(env.obs))
    s = env.get_state()
    fl = s[:, f["manifold"]].reshape(4096, env.n_members, 37)[:, :, 0]; pm = s[:, f["pairs"]].reshape(4096, npairs, 49)[:, :, 0]
    act = (fl > 0).sum(1) + (pm > 0).sum(1)
    cnt.append(int((act > 16).sum())); mx.append(int(act.max()))
print("snapshots 60: envs with > 16 live manifolds per snapshot: mean %.2f, snapshots with at least one: %d of 60; max live %d; hist of per-snapshot max: %s" % (np.mean(cnt), sum(c > 0 for c in cnt), max(mx), np.bincount(mx)[10:]))
