"""How much does the plane shortcut for the floor differ from what the reference's world does?

The reference's floor is a btConvexHullShape — cube.obj scaled (1000, 1, 1000) at (0, -2, 2) (evo_motion_model/src/env/robot_walk.cpp:22-25,
src/item.cpp:17-41) — so every member-vs-floor pair goes through the same GJK + persistent-manifold path as a member pair
(environment.cpp:20-31).  The oracle and the HIP path use deepest-hull-vertex-vs-plane instead (north_star: "plane contact").  This
diagnostic (ORACLE side only, test infrastructure) measures the difference ONE STEP AT A TIME from identical states: the same state
and action are stepped once with the plane and once with the floor as a hull pair (orc_set_floor_as_hull: the cube through
orc_narrow.cpp's btGjkPairDetector restatement, AABB cull, btManifoldResult::addContactPoint), and poses, velocities, observation,
reward, the floor manifolds (count, new contact point, distance) and the accumulated impulses are compared.

   python tests/diag/floor_hull_diff.py [--envs 8] [--steps 300]            random actions (robots that flail, fall and reset)
   tools/floor_hull_trained.py (GPU box)                                    the states a trained policy visits"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import blob  # noqa: E402
import orc  # noqa: E402


class Acc:
    def __init__(self):
        self.v = {}
        self.n = 0

    def add(self, k, x):
        self.v.setdefault(k, []).append(float(x))


def one_step(o, S, action, acc, L, half=1000.0, pretest=0):
    """step the oracle env `o` once from state S with the plane and once with the hull floor; accumulate the differences.
    half: xz half width of the floor box (1000 = the reference's; smaller = a well-conditioned box kept under each member);
    pretest: 1 = with the libccd-derived intersection pre-test of btGjkPairDetector in front of every query"""
    nb, nm, nmus = o.nb, o.nm, o.act_dim
    npairs = len(o.pairs()) if o.self_collision else 0
    f = blob.fields(nb, nm, nmus, npairs)
    orc.set_floor_as_hull(0, lib=L)
    o.set_state(S)
    obs_p, rew_p, done_p = o.do_step(action)
    Sp = o.get_state()
    orc.set_floor_as_hull(1, lib=L)
    L.orc_set_floor_hull_half(float(half))
    orc.set_ccd_pretest(pretest, lib=L)
    o.set_state(S)
    obs_h, rew_h, done_h = o.do_step(action)
    Sh = o.get_state()
    fs = o.floor_stats()
    orc.set_floor_as_hull(0, lib=L)
    L.orc_set_floor_hull_half(1000.0)
    orc.set_ccd_pretest(0, lib=L)
    for k_, v_ in fs.items():
        acc.add("floor_" + k_, v_)
    bp, bh = blob.body_view(Sp, nb), blob.body_view(Sh, nb)
    acc.add("pos", np.abs(bp["pos"] - bh["pos"]).max())
    acc.add("pose_l2", np.sqrt(((bp["pos"] - bh["pos"]) ** 2).sum()))
    acc.add("quat", np.minimum(np.abs(bp["quat"] - bh["quat"]).max(-1), np.abs(bp["quat"] + bh["quat"]).max(-1)).max())
    acc.add("lin", np.abs(bp["lin"] - bh["lin"]).max())
    acc.add("ang", np.abs(bp["ang"] - bh["ang"]).max())
    acc.add("obs", np.abs(obs_p - obs_h).max())
    acc.add("reward", abs(rew_p - rew_h))
    acc.add("done_mismatch", float(bool(done_p) != bool(done_h)))
    mp, mh = Sp[f["manifold"]].reshape(nm, 37), Sh[f["manifold"]].reshape(nm, 37)
    m0 = S[f["manifold"]].reshape(nm, 37)
    acc.add("live_points", (mp[:, 0]).sum())
    acc.add("count_mismatch", (mp[:, 0] != mh[:, 0]).sum())
    same = (mp[:, 0] == mh[:, 0]) & (mp[:, 0] > 0)
    for m in np.nonzero(same)[0]:
        n = int(mp[m, 0])
        pp, ph = mp[m, 1:1 + 9 * n].reshape(n, 9), mh[m, 1:1 + 9 * n].reshape(n, 9)
        acc.add("point_on_member", np.abs(pp[:, 3:6] - ph[:, 3:6]).max())      # local point on the member, every cached point
        acc.add("point_on_floor_xz", np.abs(pp[:, [0, 2]] - ph[:, [0, 2]]).max())
        acc.add("distance", np.abs(pp[:, 6] - ph[:, 6]).max())
        acc.add("impulse", np.abs(pp[:, 7:9] - ph[:, 7:9]).max())
        # did the member stand flat?  (two or more cached points at nearly the same height = a face or an edge on the floor)
        if n >= 2 and np.ptp(pp[:, 6]) < 2e-3:
            acc.add("flat_point_on_member", np.abs(pp[:, 3:6] - ph[:, 3:6]).max())
            acc.add("flat_impulse", np.abs(pp[:, 7:9] - ph[:, 7:9]).max())
    del m0
    acc.n += 1
    return (obs_p, rew_p, done_p), Sp


def report(acc, title, out=sys.stdout):
    out.write("%s: %d env-steps, %.2f live floor points per env-step\n" % (title, acc.n, np.mean(acc.v.get("live_points", [0]))))
    for k in ("pose_l2", "pos", "quat", "lin", "ang", "obs", "reward", "point_on_member", "point_on_floor_xz", "distance", "impulse",
              "flat_point_on_member", "flat_impulse"):
        if k in acc.v:
            x = np.array(acc.v[k])
            out.write("  %-22s median %.3g  p99 %.3g  max %.3g   (%d samples)\n" % (k, np.median(x), np.percentile(x, 99), x.max(), len(x)))
    if "floor_queries" in acc.v:
        q = max(np.sum(acc.v["floor_queries"]), 1)
        out.write("  floor queries per env-step %.2f, GJK iterations per query %.2f, through the penetration solver %.3f of the queries, pre-test says intersect %.3f\n"
                  % (q / max(acc.n, 1), np.sum(acc.v["floor_gjk_iterations"]) / q, np.sum(acc.v["floor_penetration_calls"]) / q, np.sum(acc.v["floor_ccd_intersect"]) / q))
    out.write("  manifold count mismatches %d, done mismatches %d\n" % (int(np.sum(acc.v.get("count_mismatch", [0]))), int(np.sum(acc.v.get("done_mismatch", [0])))))


def random_regime(n_envs, steps, self_collision=1, L=None, half=1000.0, pretest=0):
    L = L or orc.load()
    acc = Acc()
    for i in range(n_envs):
        o = orc.OracleEnv(seed=4321 + i, lib=L, self_collision=self_collision)
        o.reset()
        rng = np.random.default_rng(50 + i)
        for k in range(steps):
            S = o.get_state()
            a = rng.uniform(-1, 1, o.act_dim).astype(np.float32)
            (obs, rew, done), Sp = one_step(o, S, a, acc, L, half, pretest)
            o.set_state(Sp)          # the plane run is the trajectory
            if done:
                o.reset()
    return acc


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--self-collision", type=int, default=1)
    a = ap.parse_args()
    for half, pre, what in ((8.0, 0, "a 16 m box kept under the member (the algorithmic difference alone)"),
                            (1000.0, 0, "the reference's 2000 m box"),
                            (1000.0, 1, "the reference's 2000 m box + the libccd-derived pre-test of bullet3 >= 2.88")):
        report(random_regime(a.envs, a.steps, a.self_collision, half=half, pretest=pre),
               "floor as a hull pair [%s] vs the plane shortcut, one step from identical state, random actions" % what)
