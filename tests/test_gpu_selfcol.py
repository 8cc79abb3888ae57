"""Member-vs-member contacts (EvmEnvParams::self_collision = 1, the reference's behaviour: every pair of members may collide
except constraint parent / child, evo_motion_model/src/robot/constraint.cpp:65,147) — the HIP path through the C ABI against
the CPU oracle, teacher-forced (both sides start every step from the oracle's state, so the numbers are single-step errors),
and the known answers of tests/physics_cases.py on the HIP backend."""
import numpy as np
import pytest

import blob
import orc
import physics_cases as pc
from conftest import write_skeleton

pytestmark = pytest.mark.gpu

SLIDER_IMPULSE_COLS = np.arange(324, 371, 4)


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def make(n, seed=1234, **kw):
    from evomotion_amd import VecRobotWalk
    prm = dict(kw.pop("parameters", {}))
    prm["self_collision"] = 1
    return VecRobotWalk(n, seed=seed, parameters=prm, **kw)


def test_pair_table_matches_the_oracle(torch_mod, orc_lib):
    import ctypes
    from evomotion_amd._lib import check, lib
    env = make(2)
    o = orc.OracleEnv(lib=orc_lib, self_collision=1)
    assert env.n_pairs == len(o.pairs()) == 120            # 17 * 16 / 2 - 16 constraint pairs
    got = np.zeros((env.n_pairs, 2), np.int32)
    n = ctypes.c_int()
    check(lib.evm_env_pairs(env._h, ctypes.byref(n), got.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
    assert np.array_equal(got, o.pairs())
    assert env.state_size() == o.state_size()


def test_teacher_forced_steps_with_member_contacts(torch_mod, orc_lib):
    import os
    torch = torch_mod
    # (EVM_TF_ENVS / EVM_TF_STEPS: a longer run of the same comparison, e.g. 64 x 600 for profiles/r3_parity_long.txt)
    n, steps = int(os.environ.get("EVM_TF_ENVS", 32)), int(os.environ.get("EVM_TF_STEPS", 150))
    env = make(n)
    orcs = [orc.OracleEnv(seed=1234 + i, lib=orc_lib, self_collision=1) for i in range(n)]
    npairs = env.n_pairs
    for o in orcs:
        o.reset()
    rng = np.random.default_rng(0)
    worst = dict(pos=0.0, quat=0.0, lin=0.0, ang=0.0, obs=0.0, slider_imp_flips=0, rew=0.0, done=0, mf=0, pm_mismatch=0, pm_live=0,
                 pm_geom=0.0, pm_impulse=0.0)
    pair_points = 0
    for k in range(steps):
        so = np.stack([o.get_state() for o in orcs])
        env.set_state(so)
        a = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
        st = env.do_step(torch.from_numpy(a))
        og, rg, dg = st.state.cpu().numpy(), st.reward.cpu().numpy(), st.done.cpu().numpy()
        outs = [o.do_step(a[i]) for i, o in enumerate(orcs)]
        pair_points += sum(o.pair_stats()["pair_contacts"] for o in orcs)
        oo = np.stack([x[0] for x in outs])
        d = blob.compare(np.stack([o.get_state() for o in orcs]), env.get_state(), 41, 17, 12, npairs)
        for key in ("pos", "quat", "lin", "ang", "pm_geom", "pm_impulse"):
            worst[key] = max(worst[key], d[key])
        worst["mf"] = max(worst["mf"], d["mf_count"])
        worst["pm_mismatch"] += d["pm_count_mismatches"]
        worst["pm_live"] += d["pm_live"]
        e = np.abs(og - oo)
        flips = e[:, SLIDER_IMPULSE_COLS] > 1e-3
        worst["slider_imp_flips"] += int(flips.sum())
        e[:, SLIDER_IMPULSE_COLS] = np.where(flips, 0, e[:, SLIDER_IMPULSE_COLS])
        worst["obs"] = max(worst["obs"], float(e.max()))
        worst["rew"] = max(worst["rew"], float(np.abs(rg - np.array([x[1] for x in outs])).max()))
        worst["done"] += int((dg.astype(bool) != np.array([x[2] for x in outs])).sum())
        for i, o in enumerate(orcs):
            if outs[i][2]:
                o.reset()
    print("teacher-forced with member-vs-member contacts, worst:", worst, "pair contact points per env-step: %.2f" % (pair_points / (n * steps)))
    assert pair_points > 0.5 * n * steps            # the regime is exercised: member pairs really touch
    # the same fp32 tolerances as the floor-only test, for ONE 1/60 s step from identical state
    assert worst["pos"] < 5e-6 and worst["quat"] < 5e-6
    assert worst["lin"] < 5e-4 and worst["ang"] < 2e-3
    assert worst["obs"] < 2e-3 and worst["rew"] < 1e-4
    assert worst["done"] == 0 and worst["mf"] == 0
    # the narrowphase is compiled without contraction on both sides: the same transforms give the same contact point
    assert worst["pm_mismatch"] == 0, worst
    assert worst["pm_geom"] < 5e-6 and worst["pm_impulse"] < 2e-4, worst
    assert env.residual(clear=True) < 1e30           # no version wait timed out, no manifold was left out


def test_reset_and_rollout_with_member_contacts(torch_mod, orc_lib):
    """free-running: reset() (60 settle steps) and a short rollout stay close to the oracle; in-band autoreset keeps going"""
    torch = torch_mod
    n = 16
    env = make(n)
    orcs = [orc.OracleEnv(seed=1234 + i, lib=orc_lib, self_collision=1) for i in range(n)]
    st = env.reset()
    for o in orcs:
        o.reset()
    assert np.isfinite(st.state.cpu().numpy()).all()
    errs = [np.abs(env.body_poses().cpu().numpy()[i, :17, :3] - orcs[i].poses()[:17, :3]).max() for i in range(n)]
    print("member position error after reset(): max %.3g median %.3g" % (max(errs), np.median(errs)))
    assert np.median(errs) < 2e-2 and max(errs) < 0.3
    rng = np.random.default_rng(5)
    for k in range(300):
        a = torch.from_numpy(rng.uniform(-1, 1, (n, 12)).astype(np.float32))
        r = env.step_autoreset(a)
    assert np.isfinite(r.state.cpu().numpy()).all()
    p = env.body_poses().cpu().numpy()
    assert np.isfinite(p).all() and np.abs(p[..., :3]).max() < 50.0
    assert env.residual(clear=True) < 1e30
    assert env.errors() == (0, 0)   # no version wait timed out, no manifold left out of a step


# ---- known answers on the HIP backend (the same functions run on the oracle in tests/test_oracle_physics.py) ----
def test_known_answer_box_rests_on_box(tmp_path):
    gap, imp = pc.check_box_rests_on_box(pc.HipWorld(pc.skel_two_free_boxes(write_skeleton, tmp_path), self_collision=1))
    print("box on box: core gap %.4f m (2 margins = 0.08), normal impulse sum %.5f" % (gap, imp))


def test_known_answer_free_boxes_collide_inelastically(tmp_path):
    sk = pc.skel_two_free_boxes(write_skeleton, tmp_path)
    min_gap, rel = pc.check_free_boxes_collide_inelastically(pc.HipWorld(sk, self_collision=1))
    print("head-on boxes: smallest core gap %.4f m, separation speed afterwards %.4f m/s" % (min_gap, rel))


def test_known_answer_folded_arm_stops_at_the_body(tmp_path):
    sk = pc.skel_folding_arm(write_skeleton, tmp_path)
    deep1, touched = pc.check_folded_arm_stops_at_the_body(pc.HipWorld(sk, self_collision=1))
    deep0, _ = pc.check_folded_arm_stops_at_the_body(pc.HipWorld(sk, self_collision=0))
    print("forearm vs body: closest core distance %.4f m with member contacts, %.4f m without" % (deep1, deep0))
    assert touched
    assert deep1 > 2 * pc.MARGIN - 2.0 * 1.0 * pc.DT - 0.01, deep1
    assert deep0 < -0.02, deep0


@pytest.mark.parametrize("shape_b", ["cube", "feet"])
def test_deep_interpenetration_takes_the_penetration_branch(tmp_path, orc_lib, shape_b):
    """Cores that interpenetrate (GJK ends degenerate with the cores touching: btGjkPairDetector's catchDegeneracies / the invalid
    result) go through the penetration-depth solver — 42 sphere directions + a nested GJK run on the displaced pair.
    Random rollouts reach it a few times per 18 000 queries, so it gets a scene of its own: 70 pairs of boxes (more than one
    wavefront's worth: a full wave and a ragged one) set inside each other at random offsets and rotations, one physics step from
    identical state on the oracle and on the HIP path — in one-query-per-lane form the kernel deals the 42 directions of such a
    query to the lanes of its wavefront.  shape_b = "feet": the 451-vertex hull, i.e. the grouped form of the narrowphase (one
    query per 16-lane row, hulls in LDS), where the directions go to the lanes of all four rows."""
    from evomotion_amd import VecRobotWalk
    sk = write_skeleton(tmp_path / "two_bodies.skel", [dict(name="body", mass=4.0, scale=(0.5, 0.2, 0.5)),
                                                       dict(name="other", mass=0.5, t=(0.0, 1.0, 0.0), scale=(0.3, 0.25, 0.2), shape=shape_b)])
    n = 70
    env = VecRobotWalk(n, seed=1, device=0, parameters={"skeleton_json_path": sk, "self_collision": 1})
    env.debug_reset_begin()
    ow = pc.OracleWorld(sk, lib=orc_lib, self_collision=1)
    assert env.n_pairs == ow.npairs == 1
    tmpl = pc.clean_state(ow, [[0.0, 3000.0, 0.0], [0.0, 3000.0, 0.0]])
    f = pc.fields(ow)
    rng = np.random.default_rng(5)
    S = np.repeat(tmpl[None], n, 0).copy()
    for i in range(n):
        b = S[i, f["bodies"]].reshape(ow.nb, 13)
        depth = rng.uniform(0.0, 1.0)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        b[1, 0:3] = b[0, 0:3] + (d * np.array([0.8, 0.45, 0.7]) * depth * (1.0 if i % 7 else 0.0)).astype(np.float32)  # every 7th: centres coincide
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        b[1, 3:7] = q.astype(np.float32)
        S[i, f["ms"]] = b[: ow.nm, 0:3].ravel()
    env.set_state(S)
    env.debug_physics_steps(1)
    got = env.get_state()
    pen_calls = 0
    worst = dict(pos=0.0, lin=0.0, ang=0.0, geom=0.0, imp=0.0)
    for i in range(n):
        ow.set_state(S[i])
        ow.step(1)
        pen_calls += ow.e.pair_stats()["penetration_calls"]
        want = ow.state()
        pg, pw = got[i, f["pairs"]].reshape(1, 49), want[f["pairs"]].reshape(1, 49)
        assert pg[0, 0] == pw[0, 0], (i, pg[0, 0], pw[0, 0])
        if pw[0, 0] > 0:
            worst["geom"] = max(worst["geom"], float(np.abs(pg[0, 1:11] - pw[0, 1:11]).max()))   # local points, normal, distance
            worst["imp"] = max(worst["imp"], float(np.abs(pg[0, 11:13] - pw[0, 11:13]).max()))
        bg, bw = got[i, f["bodies"]].reshape(ow.nb, 13), want[f["bodies"]].reshape(ow.nb, 13)
        worst["pos"] = max(worst["pos"], float(np.abs(bg[:, 0:3] - bw[:, 0:3]).max()))
        worst["lin"] = max(worst["lin"], float(np.abs(bg[:, 7:10] - bw[:, 7:10]).max()))
        worst["ang"] = max(worst["ang"], float(np.abs(bg[:, 10:13] - bw[:, 10:13]).max()))
    print("penetration-branch queries: %d of %d; worst" % (pen_calls, n), worst)
    assert pen_calls >= n // 2, pen_calls
    assert worst["geom"] < 2e-6 and worst["pos"] < 1e-4 and worst["lin"] < 2e-3 and worst["ang"] < 2e-2 and worst["imp"] < 2e-3, worst
    assert env.errors() == (0, 0)


@pytest.mark.parametrize("shape_b", ["cube", "feet"])
def test_deep_pairs_stay_in_step_with_the_oracle_through_the_urgent_list(tmp_path, orc_lib, shape_b):
    """A pair whose query went through the penetration solver is flagged (bit 8 of its manifold count word) and its next query goes
    to the narrowphase kernel's urgent list: first blocks of the launch, one query carried by all 64 lanes of a wavefront, EPA's
    polytope cached in the lanes' registers.  Same results by construction — checked here: the scene of the test above stepped THREE
    times without touching the state in between (steps 2 and 3 take the urgent list), against the oracle doing the same."""
    from evomotion_amd import VecRobotWalk
    sk = write_skeleton(tmp_path / "two_bodies.skel", [dict(name="body", mass=4.0, scale=(0.5, 0.2, 0.5)),
                                                       dict(name="other", mass=0.5, t=(0.0, 1.0, 0.0), scale=(0.3, 0.25, 0.2), shape=shape_b)])
    n = 70
    env = VecRobotWalk(n, seed=1, device=0, parameters={"skeleton_json_path": sk, "self_collision": 1})
    env.debug_reset_begin()
    ow = pc.OracleWorld(sk, lib=orc_lib, self_collision=1)
    tmpl = pc.clean_state(ow, [[0.0, 3000.0, 0.0], [0.0, 3000.0, 0.0]])
    f = pc.fields(ow)
    rng = np.random.default_rng(11)
    S = np.repeat(tmpl[None], n, 0).copy()
    for i in range(n):
        b = S[i, f["bodies"]].reshape(ow.nb, 13)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        b[1, 0:3] = b[0, 0:3] + (d * np.array([0.5, 0.25, 0.4]) * rng.uniform(0.0, 0.8)).astype(np.float32)   # deep: stays deep for steps
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        b[1, 3:7] = q.astype(np.float32)
        S[i, f["ms"]] = b[: ow.nm, 0:3].ravel()
    env.set_state(S)
    env.penetration_queries()
    env.debug_physics_steps(3)
    got = env.get_state()
    pen_dev = env.penetration_queries()
    pen_orc = 0
    worst = dict(pos=0.0, geom=0.0, imp=0.0, count=0)
    for i in range(n):
        ow.set_state(S[i])
        for k in range(3):
            ow.step(1)
            pen_orc += ow.e.pair_stats()["penetration_calls"]
        want = ow.state()
        pg, pw = got[i, f["pairs"]].reshape(1, 49), want[f["pairs"]].reshape(1, 49)
        worst["count"] += int(pg[0, 0] != pw[0, 0])
        if pw[0, 0] > 0 and pg[0, 0] == pw[0, 0]:
            worst["geom"] = max(worst["geom"], float(np.abs(pg[0, 1:11] - pw[0, 1:11]).max()))
            worst["imp"] = max(worst["imp"], float(np.abs(pg[0, 11:13] - pw[0, 11:13]).max()))
        bg, bw = got[i, f["bodies"]].reshape(ow.nb, 13), want[f["bodies"]].reshape(ow.nb, 13)
        worst["pos"] = max(worst["pos"], float(np.abs(bg[:, 0:3] - bw[:, 0:3]).max()))
    print("penetration queries over three steps: device %d, oracle %d; worst" % (pen_dev, pen_orc), worst)
    assert pen_dev == pen_orc and pen_orc >= 2 * n            # the same queries took the branch, most of them in every step
    assert worst["count"] == 0 and worst["geom"] < 5e-6 and worst["pos"] < 1e-4 and worst["imp"] < 2e-3, worst
    assert env.errors() == (0, 0)


def test_masked_reset_leaves_the_other_envs_untouched(torch_mod):
    """evm_env_reset with a mask steps only the selected envs through reset().  The narrowphase works on lists compacted over the
    whole batch, whose counters are zeroed by one wavefront of the first setup kernel: when the mask excluded that wavefront's
    whole tile, the lists of the previous step survived and the narrowphase ran again on envs that were not part of the call.
    (Counters are now double-buffered and zeroed before the mask guard.)  Here the mask leaves out tile 0 entirely: every env of
    it must come through bit for bit — bodies, manifolds, pair manifolds, counters."""
    torch = torch_mod
    n = 128
    env = make(n)
    env.reset()
    rng = np.random.default_rng(11)
    for k in range(45):   # into the contact-rich part of the episodes
        env.step_autoreset(torch.from_numpy(rng.uniform(-1, 1, (n, 12)).astype(np.float32)))
    before = env.get_state()
    mask = torch.zeros(n, dtype=torch.uint8)
    mask[64:] = 1
    env.reset(mask=mask)
    after = env.get_state()
    assert np.array_equal(before[:64], after[:64]), int((before[:64] != after[:64]).sum())
    assert not np.array_equal(before[64:], after[64:])
    assert np.isfinite(after).all() and env.errors() == (0, 0)


def test_queries_started_ahead_of_time_change_nothing_but_the_schedule(torch_mod, monkeypatch):
    """The urgent list's speculation blocks (narrow_dev.h: every urgent entry's penetration query is also run by a block of its own from the
    first cycle of the launch; the entry's owner takes the answer or calls the run off) are scheduling only: with EVM_SPECULATE=0 the
    same rollout — resets, settle steps, flagged pairs and all — must come out bit for bit, observations every step and the full state
    blob at the end.  And the path must have been exercised: every predicted solver query took its answer from a speculation block,
    and no wait ran out."""
    torch = torch_mod
    n, steps = 512, 260

    def run(speculate):
        if speculate:
            monkeypatch.delenv("EVM_SPECULATE", raising=False)
        else:
            monkeypatch.setenv("EVM_SPECULATE", "0")
        env = make(n, seed=77)
        env.reset()
        env.stagger_episodes()
        env.penetration_queries(); env.speculation_counters()
        rng = np.random.default_rng(3)
        obs = []
        for k in range(steps):
            r = env.step_autoreset(torch.from_numpy(rng.uniform(-1, 1, (n, 12)).astype(np.float32)))
            obs.append(r.state.cpu().numpy().copy())
        return np.stack(obs), env.get_state(), env.penetration_queries(), env.speculation_counters(), env.errors()

    o1, s1, q1, c1, e1 = run(True)
    o0, s0, q0, c0, e0 = run(False)
    print("solver queries %d; speculation runs %d, answers used %d, waits run out %d" % ((q1,) + c1))
    assert c0 == (0, 0, 0) and q0 == q1 > 20
    assert c1[1] >= (9 * q1) // 10, "solver queries are predicted (urgent list) all but once in a few hundred: each of those has its speculation block"
    assert c1[2] == 0
    assert e1 == e0 == (0, 0)
    assert np.array_equal(o1, o0), int((o1 != o0).sum())
    assert np.array_equal(s1, s0), int((s1 != s0).sum())
