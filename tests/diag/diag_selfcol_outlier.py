"""Diagnostic (GPU box): the teacher-forced member-vs-member comparison of tests/test_gpu_selfcol.py over a long run, reporting
the env-steps whose single-step velocity error stands out, with what those envs hold (live manifolds, rounds, split-impulse
depth, resets) — to tell an ill-conditioned step from a rare path that is wrong.
    python tests/diag/diag_selfcol_outlier.py [envs] [steps] [threshold] [self_collision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import blob, orc
from evomotion_amd import VecRobotWalk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 3e-4
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
L = orc.load()
env = VecRobotWalk(n, seed=1234, parameters={"self_collision": mode})
orcs = [orc.OracleEnv(seed=1234 + i, lib=L, self_collision=mode) for i in range(n)]
npairs = env.n_pairs
pairs = orcs[0].pairs() if npairs else np.zeros((0, 2), int)
for o in orcs:
    o.reset()
rng = np.random.default_rng(0)
f = blob.fields(41, 17, 12, npairs)
hits = 0
for k in range(steps):
    so = np.stack([o.get_state() for o in orcs])
    env.set_state(so)
    a = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
    env.do_step(torch.from_numpy(a))
    outs = [o.do_step(a[i]) for i, o in enumerate(orcs)]
    sa = np.stack([o.get_state() for o in orcs]); sb = env.get_state()
    va, vb = blob.body_view(sa, 41), blob.body_view(sb, 41)
    el = np.abs(va["lin"] - vb["lin"]).reshape(n, -1).max(1); ea = np.abs(va["ang"] - vb["ang"]).reshape(n, -1).max(1)
    for i in np.nonzero((el > thr) | (ea > 4 * thr))[0]:
        hits += 1
        fl1 = sa[i, f["manifold"]].reshape(17, 37); pm1 = sa[i, f["pairs"]].reshape(npairs, 49) if npairs else np.zeros((0, 49), np.float32)
        live_f, live_p = int((fl1[:, 0] > 0).sum()), int((pm1[:, 0] > 0).sum())
        nf = np.zeros(17, int); rounds = 1 if live_f else 0
        nf[fl1[:, 0] > 0] = 1
        for p in np.nonzero(pm1[:, 0] > 0)[0]:
            x, y = pairs[p]; r = max(nf[x], nf[y]); nf[x] = nf[y] = r + 1; rounds = max(rounds, r + 1)
        dist_f = fl1[:, 1:].reshape(17, 4, 9)[:, :, 6]; dist_p = pm1[:, 1:].reshape(npairs, 4, 12)[:, :, 9]
        deep = min(float(dist_f[fl1[:, 0] > 0].min()) if live_f else 0.0, float(dist_p[pm1[:, 0] > 0].min()) if live_p else 0.0)
        body = int(np.abs(va["lin"][i] - vb["lin"][i]).max(1).argmax())
        imp_p = pm1[:, 1:].reshape(npairs, 4, 12)[:, :, 10].max() if live_p else 0.0
        print("step %d env %d: lin err %.2e ang err %.2e (body %d)  live manifolds %d floor + %d pairs, rounds %d, deepest point %.4f, pending %d, done %d, max pair impulse %.3f, |v|max %.2f"
              % (k, i, el[i], ea[i], body, live_f, live_p, rounds, deep, int(so[i, f["pending"]][0]), int(outs[i][2]), imp_p, float(np.abs(va["lin"][i]).max())))
    for i, o in enumerate(orcs):
        if outs[i][2]:
            o.reset()
print("self_collision=%d: " % mode, end="")
print("env-steps %d, outliers %d (threshold lin %.1e / ang %.1e); errors %s" % (n * steps, hits, thr, 4 * thr, env.errors()))
