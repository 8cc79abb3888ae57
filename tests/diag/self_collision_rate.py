"""How often would the reference (Bullet3) have produced member<->member contacts that this path cannot?

DESIGN.md §2 states the deviation: only member-vs-floor contacts are modelled (north_star: "plane contact"); Bullet also
runs its narrowphase on every pair of members whose AABBs overlap, except pairs joined by a constraint
(setIgnoreCollisionCheck, constraint.cpp:65,147).  This diagnostic (ORACLE side, test infrastructure, never imported by
the product) rolls the CPU oracle with uniform random actions and counts, per env-step, the non-adjacent member pairs
  (a) whose margin-inflated world AABBs overlap             = pairs Bullet's broadphase would hand to the narrowphase
  (b) whose hulls are closer than the two 0.04 margins      = pairs that would really get a contact point; tested with a
      separating-axis search over the face normals of both (oriented) boxes and the 9 edge cross products on the hulls'
      vertex sets — exact for the cube members, and for the feet applied to their vertex hull (conservative: "no
      separating axis found among those" counts as touching)
With --self-collision 1 the oracle runs WITH member-vs-member contacts (EvmEnvParams::self_collision, the reference's
behaviour) and the same geometric test, independent of the oracle's own GJK, reports what is left: pairs still closer than the
margins are the contacts being held (expected), pairs whose un-margined cores overlap would be real inter-penetration.  The
oracle's narrowphase statistics ride along (pairs tested, GJK iterations, calls of the penetration solver).
   python tests/diag/self_collision_rate.py [--envs 8] [--steps 600] [--self-collision 0|1]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

MARGIN = 0.04


def load_members(skel=orc.SKEL):
    """names, scaled hull vertices [n,3] per member, and the set of member pairs joined by a constraint"""
    lines = open(skel).read().split("\n")
    shapes, i = {}, next(k for k, l in enumerate(lines) if l.startswith("shapes ")) + 1
    while i < len(lines) and lines[i].startswith("shape "):
        _, name, n, _ = lines[i].split()
        shapes[name] = np.array([[float.fromhex(t) for t in lines[i + 1 + k].split()] for k in range(int(n))])
        i += 1 + int(n)
    names, hulls = [], []
    for l in lines:
        if l.startswith("member "):
            t = l.split()
            vals = [float.fromhex(v) for v in t[3:15]]
            names.append(t[1])
            hulls.append(shapes[t[2]] * np.array(vals[9:12]))
    adjacent = set()
    for l in lines:
        if l.startswith(("hinge ", "fixed ")):
            t = l.split()
            adjacent.add(frozenset((names.index(t[2]), names.index(t[3]))))
    return names, hulls, adjacent


def rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def separated(va, vb, Ra, Rb, gap):
    """True when some axis (face normals of both local frames + their 9 cross products) separates the vertex sets by > gap"""
    axes = [Ra[:, k] for k in range(3)] + [Rb[:, k] for k in range(3)]
    for i in range(3):
        for j in range(3):
            c = np.cross(Ra[:, i], Rb[:, j])
            n = np.linalg.norm(c)
            if n > 1e-6:
                axes.append(c / n)
    for ax in axes:
        pa, pb = va @ ax, vb @ ax
        if pa.min() - pb.max() > gap or pb.min() - pa.max() > gap:
            return True
    return False


def count(poses, hulls, adjacent, core_gap=0.0):
    nm = len(hulls)
    world = [poses[m, :3] + hulls[m] @ rot(poses[m, 3:]).T for m in range(nm)]
    lo = np.array([w.min(0) - MARGIN for w in world])
    hi = np.array([w.max(0) + MARGIN for w in world])
    aabb = touch = core = 0
    for a in range(nm):
        for b in range(a + 1, nm):
            if frozenset((a, b)) in adjacent:
                continue
            if np.all(lo[a] <= hi[b]) and np.all(lo[b] <= hi[a]):
                aabb += 1
                Ra, Rb = rot(poses[a, 3:]), rot(poses[b, 3:])
                if not separated(world[a], world[b], Ra, Rb, 2 * MARGIN):
                    touch += 1
                    if not separated(world[a], world[b], Ra, Rb, core_gap):
                        core += 1
    return aabb, touch, core


def run(n_envs=8, steps=600, seed=1234, lib=None, self_collision=0, core_gap=0.07):
    """core_gap: a pair counts as inter-penetrating when no tested axis separates the un-margined hulls by more than this
    (0.07 = the margins have given way by more than 1 cm)"""
    names, hulls, adjacent = load_members()
    out = dict(env_steps=0, steps_with_aabb_pair=0, steps_with_touching_pair=0, aabb_pairs=0, touching_pairs=0, core_pairs=0,
               pair_contacts=0, live_pairs=0, pair_tests=0, gjk_iterations=0, penetration_calls=0, deepest=0.0, steps_with_pair_contact=0)
    for i in range(n_envs):
        e = orc.OracleEnv(seed=seed + i, lib=lib, self_collision=self_collision)
        e.reset()
        rng = np.random.default_rng(seed + 1000 + i)
        done = False
        for k in range(steps):
            if done:
                e.reset()
                done = False
            _, _, done = e.do_step(rng.uniform(-1, 1, 12).astype(np.float32))
            a, t, c = count(e.poses()[:17].astype(np.float64), hulls, adjacent, core_gap)
            out["env_steps"] += 1
            out["aabb_pairs"] += a
            out["touching_pairs"] += t
            out["core_pairs"] += c
            if self_collision:
                st = e.pair_stats()
                for k_ in ("pair_contacts", "live_pairs", "pair_tests", "gjk_iterations", "penetration_calls"):
                    out[k_] += st[k_]
                out["deepest"] = min(out["deepest"], st["deepest"])
                out["steps_with_pair_contact"] += st["pair_contacts"] > 0
            out["steps_with_aabb_pair"] += a > 0
            out["steps_with_touching_pair"] += t > 0
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--self-collision", type=int, default=0)
    a = ap.parse_args()
    r = run(a.envs, a.steps, self_collision=a.self_collision)
    n = r["env_steps"]
    print("env-steps %d (uniform random actions, resets included)" % n)
    print("non-adjacent member pairs with overlapping margin-inflated AABBs: %.2f per env-step; env-steps with at least one: %.1f %%"
          % (r["aabb_pairs"] / n, 100.0 * r["steps_with_aabb_pair"] / n))
    print("non-adjacent member pairs closer than the two margins (would get a Bullet contact point): %.3f per env-step; "
          "env-steps with at least one: %.1f %%" % (r["touching_pairs"] / n, 100.0 * r["steps_with_touching_pair"] / n))
    print("of those, pairs whose cores are within 0.07 m (margins given way by more than 1 cm): %.4f per env-step" % (r["core_pairs"] / n))
    if a.self_collision:
        print("oracle narrowphase: %.2f pairs tested per env-step (boxes overlap), %.2f GJK iterations per test, penetration solver "
              "called in %d of %d tests; %.2f live pair manifolds and %.2f pair contact points per env-step, env-steps with a pair "
              "contact %.1f %%; deepest pair contact distance %.4f m (margins are 0.08 m together)"
              % (r["pair_tests"] / n, r["gjk_iterations"] / max(r["pair_tests"], 1), r["penetration_calls"], r["pair_tests"],
                 r["live_pairs"] / n, r["pair_contacts"] / n, 100.0 * r["steps_with_pair_contact"] / n, r["deepest"]))
