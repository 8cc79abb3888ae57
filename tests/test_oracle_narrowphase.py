"""The narrowphase restatement (oracle/orc_narrow.cpp: GJK closest points after btGjkPairDetector / btVoronoiSimplexSolver) against
an INDEPENDENT known answer that shares no code and no algorithm with it: convex duality.  For two convex polytopes and ANY unit
direction n, the support gap  sep(n) = min_i n.a_i - max_j n.b_j  is a LOWER bound of their distance, and the length of ANY
segment between a point of conv(A) and a point of conv(B) is an UPPER bound; the two meet only at the true distance and the true
(unique) closest direction.  So a query result is exact if and only if
    * its witness points lie in the two hulls (checked as linear-programme feasibility, scipy.optimize.linprog, float64), and
    * the support gap along its normal equals the length of the witness segment.
Reported quantities follow Bullet's convention: distance = core distance - the two 0.04 margins, the witness on B pushed out by B's
margin along the normal (from B towards A).  Bullet is not installed here, so this is the analytic pin of the round's new collision
code; the HIP narrowphase is held to this oracle point for point (tests/test_gpu_selfcol.py).  Touching / penetrating CORES go through
the penetration solver (EPA, oracle/orc_epa.cpp) and have their own known answers below: the separating-axis depth for boxes, the
overlap along sampled directions for general hulls."""
import numpy as np
import pytest
from scipy.optimize import linprog

import orc

MARGIN = 0.04
CUBE = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _world(pts, scale, R, o):
    return (np.asarray(pts, np.float64) * np.asarray(scale, np.float64)) @ R.T + o


def _outside_hull(P, x):
    """how far x is from conv(P) in the max norm: min t  s.t.  |P^T mu - x| <= t, mu >= 0, sum mu = 1"""
    n = len(P)
    c = np.zeros(n + 1); c[-1] = 1.0
    A_ub = np.block([[P.T, -np.ones((3, 1))], [-P.T, -np.ones((3, 1))]])
    b_ub = np.concatenate([x, -x])
    A_eq = np.concatenate([np.ones(n), [0.0]])[None, :]
    res = linprog(c, A_ub=A_ub, b_ub=b_ub, A_eq=A_eq, b_eq=[1.0], bounds=[(0, None)] * (n + 1), method="highs")
    assert res.status == 0
    return float(res.x[-1])


def _random_hull(rng, n):
    p = rng.normal(size=(n, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    return (p * rng.uniform(0.6, 1.0, (n, 1))).astype(np.float32)


@pytest.mark.parametrize("kind", ["box_box", "box_hull", "hull_hull", "box_big_hull"])
def test_gjk_results_are_certified_by_convex_duality(orc_lib, kind):
    rng = np.random.default_rng({"box_box": 1, "box_hull": 2, "hull_hull": 3, "box_big_hull": 4}[kind])
    checked = in_margin = 0
    worst_gap = worst_in = 0.0
    for trial in range(120):
        ptsA = CUBE if kind.startswith("box") else _random_hull(rng, 24)
        ptsB = CUBE if kind == "box_box" else _random_hull(rng, 451 if kind == "box_big_hull" else 32)
        sA, sB = rng.uniform(0.08, 0.4, 3).astype(np.float32), rng.uniform(0.08, 0.4, 3).astype(np.float32)
        RA, RB = _rot(rng).astype(np.float32), _rot(rng).astype(np.float32)
        oA = rng.uniform(-0.2, 0.2, 3).astype(np.float32)
        d = rng.normal(size=3)
        oB = (oA + d / np.linalg.norm(d) * rng.uniform(0.15, 0.9)).astype(np.float32)
        r = orc.gjk_query(ptsA, sA, (RA, oA), ptsB, sB, (RB, oB), lib=orc_lib)
        assert r["has"], (kind, trial)
        if r["used_penetration"] or r["distance"] + 2 * MARGIN < 2e-3:   # touching / penetrating cores
            continue
        A, B = _world(ptsA, sA, RA.astype(np.float64), oA.astype(np.float64)), _world(ptsB, sB, RB.astype(np.float64), oB.astype(np.float64))
        n = r["normal"].astype(np.float64)
        assert abs(np.linalg.norm(n) - 1.0) < 1e-5
        core = r["distance"] + 2 * MARGIN                              # length of the witness segment between the cores
        pb = r["point_b"].astype(np.float64) - MARGIN * n            # witness on B's core
        pa = pb + core * n                                             # ... and on A's
        lower = float((A @ n).min() - (B @ n).max())                   # support gap along the reported normal <= true distance
        worst_in = max(worst_in, _outside_hull(A, pa), _outside_hull(B, pb))
        worst_gap = max(worst_gap, abs(core - lower))
        assert lower <= core + 2e-5, (kind, trial, lower, core)
        checked += 1
        in_margin += core < 2 * MARGIN
    print("%s: %d queries certified (%d inside the margins): duality gap <= %.2e, witnesses within %.2e of the hulls" % (kind, checked, in_margin, worst_gap, worst_in))
    assert checked >= 40 and in_margin >= 5
    assert worst_in < 5e-5
    # Boxes: exact to fp32 rounding (observed 6e-6).  Many-vertex hulls: Bullet's simplex solver treats a new support point within
    # 1 cm of a simplex vertex as already in the simplex (BT_USE_EQUAL_VERTEX_THRESHOLD, distance^2 <= 1e-4) and stops there, so the
    # answer may be short of the optimum by a fraction of that centimetre (observed up to 5.7e-3): restated on purpose, bounded here.
    assert worst_gap < (2e-5 if kind == "box_box" else 1e-2)


def _sat_depth(RA, hA, oA, RB, hB, oB):
    """exact penetration depth of two overlapping boxes (the minimal translation that separates them): the smallest overlap over the
    15 separating-axis candidates — what an exact solver such as Bullet's EPA converges to"""
    axes = [RA[:, i] for i in range(3)] + [RB[:, i] for i in range(3)]
    for i in range(3):
        for j in range(3):
            c = np.cross(RA[:, i], RB[:, j])
            if np.linalg.norm(c) > 1e-6:
                axes.append(c / np.linalg.norm(c))
    t = oB - oA
    return min(sum(hA[i] * abs(n @ RA[:, i]) for i in range(3)) + sum(hB[i] * abs(n @ RB[:, i]) for i in range(3)) - abs(n @ t) for n in axes)


def _sat_axis(RA, hA, oA, RB, hB, oB):
    """(smallest overlap, its axis, second smallest overlap) over the 15 separating-axis candidates"""
    axes = [RA[:, i] for i in range(3)] + [RB[:, i] for i in range(3)]
    for i in range(3):
        for j in range(3):
            c = np.cross(RA[:, i], RB[:, j])
            if np.linalg.norm(c) > 1e-6:
                axes.append(c / np.linalg.norm(c))
    t = oB - oA
    ov = np.array([sum(hA[i] * abs(n @ RA[:, i]) for i in range(3)) + sum(hB[i] * abs(n @ RB[:, i]) for i in range(3)) - abs(n @ t) for n in axes])
    k = np.argsort(ov)
    return ov[k[0]], axes[k[0]], ov[k[1]]


def _overlapping_boxes(rng):
    sA, sB = rng.uniform(0.08, 0.4, 3).astype(np.float32), rng.uniform(0.08, 0.4, 3).astype(np.float32)
    RA, RB = _rot(rng).astype(np.float32), _rot(rng).astype(np.float32)
    oA = rng.uniform(-0.2, 0.2, 3).astype(np.float32)
    d = rng.normal(size=3)
    oB = (oA + d / np.linalg.norm(d) * rng.uniform(0.02, 0.3)).astype(np.float32)
    return sA, sB, RA, RB, oA, oB


def test_epa_depth_equals_the_exact_separating_axis_depth(orc_lib):
    """Cores that overlap go to the penetration solver: btGjkEpaPenetrationDepthSolver, as in the reference's world
    (btDefaultCollisionConfiguration; oracle/orc_epa.cpp restates btGjkEpa2.cpp).  Boxes have an exact answer that shares nothing
    with EPA — the separating-axis theorem: the penetration depth is the smallest overlap over the 15 candidate axes and the
    contact normal is that axis.  EPA runs on the margin-inflated shapes, so its depth is the core depth + the two 0.04 margins,
    to its own termination tolerance EPA_ACCURACY = 1e-4."""
    rng = np.random.default_rng(5)
    err, ang, n_unique = [], [], 0
    for trial in range(400):
        sA, sB, RA, RB, oA, oB = _overlapping_boxes(rng)
        core, axis, second = _sat_axis(RA.astype(np.float64), sA.astype(np.float64), oA.astype(np.float64), RB.astype(np.float64),
                                       sB.astype(np.float64), oB.astype(np.float64))
        if core <= 1e-3:
            continue
        r = orc.gjk_query(CUBE, sA, (RA, oA), CUBE, sB, (RB, oB), lib=orc_lib)
        assert r["has"] and r["used_penetration"] and r["distance"] < 0, (trial, core, r)
        err.append(-r["distance"] - (core + 2 * MARGIN))
        if second - core > 5e-3:      # the minimal axis is unique: the reported normal must be it (up to sign)
            n_unique += 1
            ang.append(1.0 - abs(float(r["normal"].astype(np.float64) @ axis)))
    err = np.array(err)
    print("EPA depth - exact depth over %d overlapping box pairs: min %.2e max %.2e; 1 - |n . axis| <= %.2e over %d pairs with a unique axis"
          % (len(err), err.min(), err.max(), max(ang), n_unique))
    assert len(err) > 300 and n_unique > 200
    assert np.abs(err).max() < 1.2e-4          # observed 8.8e-5 (never above the exact depth: the polytope is inscribed)
    assert err.max() < 1e-6
    assert max(ang) < 1e-4


@pytest.mark.parametrize("kind", ["box_hull", "hull_hull", "box_big_hull"])
def test_epa_results_on_hulls_are_certified_by_the_overlap_along_directions(orc_lib, kind):
    """For general hulls the penetration depth is min over unit directions n of the overlap  max_i n.a_i - min_j n.b_j  (+ the
    margins).  So: the overlap along EPA's own normal must equal its reported depth, and no other direction may give a smaller one."""
    rng = np.random.default_rng({"box_hull": 12, "hull_hull": 13, "box_big_hull": 14}[kind])
    dirs = rng.normal(size=(4000, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    checked, worst_self, worst_lower = 0, 0.0, 0.0
    for trial in range(60):
        ptsA = CUBE if kind.startswith("box") else _random_hull(rng, 24)
        ptsB = _random_hull(rng, 451 if kind == "box_big_hull" else 32)
        sA, sB = rng.uniform(0.08, 0.4, 3).astype(np.float32), rng.uniform(0.08, 0.4, 3).astype(np.float32)
        RA, RB = _rot(rng).astype(np.float32), _rot(rng).astype(np.float32)
        oA = rng.uniform(-0.2, 0.2, 3).astype(np.float32)
        d = rng.normal(size=3)
        oB = (oA + d / np.linalg.norm(d) * rng.uniform(0.0, 0.12)).astype(np.float32)
        r = orc.gjk_query(ptsA, sA, (RA, oA), ptsB, sB, (RB, oB), lib=orc_lib)
        if not (r["has"] and r["used_penetration"] and r["distance"] < -2 * MARGIN - 1e-3):
            continue
        A, B = _world(ptsA, sA, RA.astype(np.float64), oA.astype(np.float64)), _world(ptsB, sB, RB.astype(np.float64), oB.astype(np.float64))
        n = r["normal"].astype(np.float64)
        depth = -r["distance"] - 2 * MARGIN                               # of the cores
        # normal on B points from B towards A: A must move along +n by `depth` to separate
        along = float((B @ n).max() - (A @ n).min())
        other = ((B @ dirs.T).max(axis=0) - (A @ dirs.T).min(axis=0)).min()
        worst_self = max(worst_self, abs(along - depth))
        worst_lower = max(worst_lower, depth - other)
        checked += 1
    print("%s: %d penetrating queries: |overlap along the reported normal - reported depth| <= %.2e, depth - best sampled overlap <= %.2e"
          % (kind, checked, worst_self, worst_lower))
    assert checked >= 25
    # EPA works on the margin-rounded shapes: where the deepest feature is an edge or a vertex its face normal is within
    # sqrt(2 EPA_ACCURACY / margin) of the true direction, a second-order effect on the overlap (observed 6e-4); its depth
    # (inscribed polytope) was never above any sampled direction's overlap
    assert worst_self < 1.5e-3 and worst_lower < 1e-5


def test_the_former_deviation_sampled_penetration_depth_against_the_exact_one(orc_lib):
    """Rounds 2-3 resolved overlapping cores with btMinkowskiPenetrationDepthSolver's 42 sampled directions instead of EPA.  The
    oracle keeps that solver behind a switch so that the size of the removed deviation stays a measured fact: never below the exact
    depth, 5 % above it in the median, 28 % at worst."""
    rng = np.random.default_rng(5)
    ratios = []
    orc.set_penetration_solver(1, lib=orc_lib)
    try:
        for trial in range(400):
            sA, sB, RA, RB, oA, oB = _overlapping_boxes(rng)
            core = _sat_depth(RA.astype(np.float64), sA.astype(np.float64), oA.astype(np.float64), RB.astype(np.float64), sB.astype(np.float64), oB.astype(np.float64))
            if core <= 1e-3:
                continue
            r = orc.gjk_query(CUBE, sA, (RA, oA), CUBE, sB, (RB, oB), lib=orc_lib)
            assert r["has"] and r["used_penetration"] and r["distance"] < 0, (trial, core, r)
            ratios.append(-r["distance"] / (core + 2 * MARGIN))     # both include the two margins
    finally:
        orc.set_penetration_solver(0, lib=orc_lib)
    ratios = np.array(ratios)
    print("sampled / exact penetration depth over %d overlapping box pairs: min %.3f median %.3f p90 %.3f max %.3f"
          % (len(ratios), ratios.min(), np.median(ratios), np.percentile(ratios, 90), ratios.max()))
    assert len(ratios) > 300
    assert ratios.min() > 1 - 1e-3 and np.median(ratios) < 1.1 and ratios.max() < 1.5
