"""Narrowphase work of the member-vs-member mode on a 4096-env rollout (diagnostic, GPU box): per step the broadphase's
work-list sizes per pair class, the wavefront count of the narrowphase kernel, the live manifolds / contact rounds, and how many
queries went through the penetration-depth solver (EPA) — under random actions and, with --trained N, after N PPO updates with
the policy's own actions (the regime a trained robot visits).
   python tools/pair_stats.py [--steps 40] [--trained 250]"""
import argparse, ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from evomotion_amd import VecRobotWalk, RandomAgent
from evomotion_amd._lib import lib, check
import blob

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=40); ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--trained", type=int, default=0)
a = ap.parse_args()
env = VecRobotWalk(a.envs, seed=1234, device=0)
env.reset(); env.stagger_episodes()
ag = RandomAgent([env.action_dim], env.device, seed=1)
for i in range(192):
    env.step_autoreset(ag.act(env.obs))
npairs = env.n_pairs
pairs = np.zeros((npairs, 2), np.int32); n = ctypes.c_int()
check(lib.evm_env_pairs(env._h, ctypes.byref(n), pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
hull = np.array([8] * env.n_members); 
# members with the big hull: read the skeleton fixture
lines = open(os.path.join(ROOT, "evomotion_amd", "data", "robot_walk_spider.skel")).read().split("\n")
shp = [l.split()[2] for l in lines if l.startswith("member ")]
big = np.array([s == "feet" for s in shp])
nbig = big[pairs[:, 0]].astype(int) + big[pairs[:, 1]].astype(int)
cnt = np.zeros(npairs + 1, np.int32)
tot = []
env.penetration_queries()
st0 = env.stats(); env.clear_stats()
for k in range(a.steps):
    env.step_autoreset(ag.act(env.obs))
    check(lib.evm_env_debug_pair_counts(env._h, cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
    small, bigc = cnt[:npairs], int(cnt[npairs])
    waves = np.ceil(small / 64).astype(int)
    tot.append([small.sum() + bigc, small.sum(), bigc, waves.sum(), (bigc + 3) // 4])
tot = np.array(tot, float)
m_ = tot.mean(0)
print("per step, mean over %d steps of %d envs: narrowphase queries %.0f = %.2f per env (small-hull pairs %.0f in %.0f wavefronts of one env per lane; "
      "big-hull pairs %.0f in %.0f wavefronts of four queries)" % (a.steps, a.envs, m_[0], m_[0] / a.envs, m_[1], m_[3], m_[2], m_[4]))
s = env.get_state(); f = blob.fields(env.n_bodies, env.n_members, env.n_muscles, npairs)
fl = s[:, f["manifold"]].reshape(a.envs, env.n_members, 37)[:, :, 0]; pm = s[:, f["pairs"]].reshape(a.envs, npairs, 49)[:, :, 0]
act = (fl > 0).sum(1) + (pm > 0).sum(1)
print("live manifolds per env: mean %.2f max %d; live pair manifolds per env %.2f; envs with > 16: %d" % (act.mean(), act.max(), (pm > 0).sum(1).mean(), (act > 16).sum()))
R = []
for e in range(a.envs):
    nf = np.zeros(env.n_members, int); r_ = 1 if (fl[e] > 0).any() else 0
    nf[fl[e] > 0] = 1
    for p in np.nonzero(pm[e] > 0)[0]:
        x, y = pairs[p]; r = max(nf[x], nf[y]); nf[x] = nf[y] = r + 1; r_ = max(r_, r + 1)
    R.append(r_)
R = np.array(R); Rw = R.reshape(-1, 16).max(1)
print("contact rounds per env: mean %.2f max %d; per 16-env workgroup (max over its envs): mean %.2f max %d" % (R.mean(), R.max(), Rw.mean(), Rw.max()))
pen = env.penetration_queries(); st = env.stats()
print("penetration-depth solver (EPA): %d of %.0f queries in %d steps of %d envs = %.2f per step (%.2e of the queries), %d of them predicted (urgent list: %d entries in all); do_step share of the calls %.2f"
      % (pen, tot[:, 0].sum(), a.steps, a.envs, pen / a.steps, pen / tot[:, 0].sum(), env.predicted_penetration_queries, env.urgent_entries, st["env_steps"] / (a.steps * a.envs)))
print("errors", env.errors())
if a.trained:
    from evomotion_amd import VecPpoGaeAgent
    agent = VecPpoGaeAgent(5, [env.state_dim], [env.action_dim], hidden_size=256, device=0, horizon=32, epoch=8, learning_rate=3e-4)
    for u in range(a.trained):
        agent.rollout(env); agent.update()
    env.penetration_queries(); env.clear_stats()
    q = 0
    nsteps = 8 * 32
    for u in range(8):
        for k in range(32):
            act, _, _ = agent.fused.forward(env.obs, seed=5000 + 32 * u + k)
            env.step_autoreset(act)
            check(lib.evm_env_debug_pair_counts(env._h, cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
            q += int(cnt.sum())
    pen = env.penetration_queries(); st = env.stats()
    print("trained regime (%d PPO updates, policy actions): EPA %d of %d queries in %d steps = %.2f per step (%.2e of the queries); do_step share of the calls %.2f, resets started %d"
          % (a.trained, pen, q, nsteps, pen / nsteps, pen / max(q, 1), st["env_steps"] / (nsteps * a.envs), st["resets"]))
