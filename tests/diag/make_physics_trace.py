"""Writes tests/golden/physics_trace.txt (floor contacts only) and tests/golden/physics_trace_selfcol.txt (member-vs-member
contacts, the reference's behaviour): a SELF-PIN of the CPU oracle (oracle/liborc.so) — 4 environments x 256 do_step
calls with seeded uniform actions (resets included), every 4th call recorded.  It is not a reference output (Bullet3 cannot
be built here): its purpose is that a refactor of the oracle cannot drift silently
(tests/test_oracle_constants.py::test_oracle_matches_its_committed_trace).   python tests/diag/make_physics_trace.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

N_ENV, N_STEP, EVERY = 4, 256, 4


def hash_action(env, call, k):
    """oracle/bullet_harness.cpp::hash_action — an action stream both programs can produce"""
    h = ((env * 1000003 + call) * 2654435761 + k * 40503 + 12345) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0x5bd1e995) & 0xFFFFFFFF
    h ^= h >> 15
    return np.float32(np.float32(h & 0xFFFFFF) / np.float32(16777216.0)) * np.float32(2.0) - np.float32(1.0)


def trace(lib=None, self_collision=0, hash_actions=False):
    rows = []
    for i in range(N_ENV):
        e = orc.OracleEnv(seed=1234 + i, lib=lib, self_collision=self_collision)
        obs, rew, done = e.reset()
        rng = np.random.default_rng(100 + i)
        for k in range(N_STEP):
            a = rng.uniform(-1, 1, 12).astype(np.float32)
            if hash_actions:
                a = np.array([hash_action(i, k, j) for j in range(12)], np.float32)
            if done:
                obs, rew, done = e.reset()
            else:
                obs, rew, done = e.do_step(a)
            if k % EVERY == EVERY - 1:
                p = e.poses()
                rows.append([i, k, int(done), rew, float(p[0, 0]), float(p[0, 1]), float(p[0, 2]), float(np.abs(p[:17, :3]).sum()),
                             float(obs[:19].sum()), float(np.abs(obs).sum())])
    return np.array(rows, np.float64)


if __name__ == "__main__":
  if "--hash-actions" in sys.argv:
    # the restatement's side of a comparison with `oracle/_ref/bullet_harness <fixture>` on a machine that has Bullet3: stdout, same format
    mode = 0 if "--self-collision=0" in sys.argv else 1
    print("# oracle/liborc.so, hash actions, self_collision=%d: env, call, done, reward, root xyz, sum |member positions|, sum(root block of the observation), sum |observation|" % mode)
    for r in trace(self_collision=mode, hash_actions=True):
        print("%d %d %d %s" % (r[0], r[1], r[2], " ".join("%.9g" % v for v in r[3:])))
    sys.exit(0)
  for mode, name in ((0, "physics_trace.txt"), (1, "physics_trace_selfcol.txt")):
    t = trace(self_collision=mode)
    out = os.path.join(ROOT, "tests", "golden", name)
    with open(out, "w") as f:
        f.write("# SELF-PIN of oracle/liborc.so (tests/diag/make_physics_trace.py), not a Bullet3 output: env, call, done, reward, root xyz, "
                "sum |member positions|, sum(root block of the observation), sum |observation|\n")
        for r in t:
            f.write("%d %d %d %s\n" % (r[0], r[1], r[2], " ".join("%.9g" % v for v in r[3:])))
    print("wrote", out, t.shape)
