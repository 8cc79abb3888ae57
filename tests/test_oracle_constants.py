"""Pins of the CPU oracle that do NOT need Bullet: dimensions, RNG stream, counters.
The physics half of the oracle has no golden vectors in the reference (parity unpinned, see oracle/orc_world.h)."""
import ctypes

import numpy as np

import orc


def test_dimensions(orc_lib):
    e = orc.OracleEnv(lib=orc_lib)
    # 371 / 12 are pinned independently by the reference's checked-in checkpoint
    # resources/robot_walk_crossq_save_34/actor.th: head.0.weight (256, 371), mu.0.weight (12, 256)
    assert e.obs_dim == 371 and e.act_dim == 12
    assert e.nb == 41 and e.nm == 17
    c = e.counters()
    assert c["max_steps"] == 1799  # int(30.f / (1.f/60.f)) in fp32, SURVEY App. D.3
    assert c["remaining_steps"] == 59
    k = e.body_constants()
    assert abs(float(k[:, 0].sum()) - 9.875) < 1e-6  # total mass, SURVEY App. A


def test_rng_matches_libstdcxx(orc_lib):
    # the restated MT19937 + uniform_real_distribution<float> against the container's libstdc++ itself
    assert orc_lib.orc_selftest_rng(1234, 20000) == 0
    assert orc_lib.orc_selftest_rng(0, 2000) == 0
    assert orc_lib.orc_selftest_rng(2**31 - 1, 2000) == 0
    u = np.zeros(6, np.float32)
    orc_lib.orc_rng_draws(1234, 6, u.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    # SURVEY App. D.5 (produced with g++/libstdc++ from the reference's expressions)
    np.testing.assert_allclose(u[:3], [0.191519454, 0.497663677, 0.622108757], rtol=0, atol=1e-9)
    lim = np.float32(np.float32(np.pi) * np.float32(2) / np.float32(3))
    ang = u * lim - lim / np.float32(2)
    np.testing.assert_allclose(ang[:3], [-0.646080136, -0.00489318371, 0.25574398], rtol=0, atol=2e-7)
    np.testing.assert_allclose(ang[3:], [0.665679216, -0.130422711, 0.234806538], rtol=0, atol=2e-7)


def test_reset_contract(orc_lib):
    e = orc.OracleEnv(seed=1234, lib=orc_lib)
    obs, reward, done = e.reset()
    assert obs.shape == (371,) and np.isfinite(obs).all()
    c = e.counters()
    assert c["curr_step"] == 1  # reset()'s own compute_step already counts (environment.cpp:45-48)
    assert c["remaining_steps"] in (58, 59, 60)
    assert 212 <= c["joint_rows"] <= 240  # 12*(5|6) + 4*6 + 12*(4..6) + 24*3
    assert (obs[15::19][:17] == 0).all()  # "touched" is always 0
    assert reward == obs[5]  # reward = root linear velocity z, which is also obs[5]
    # first observation's "acceleration" block is (0 - v): history starts at zero (App. D.2)
    np.testing.assert_allclose(obs[9:12], -obs[3:6], atol=0)


def test_history_survives_reset(orc_lib):
    e = orc.OracleEnv(seed=7, lib=orc_lib)
    obs, _, _ = e.reset()
    a = np.zeros(12, np.float32)
    for _ in range(3):
        obs, _, _ = e.do_step(a)
    v_last = obs[3:6].copy()
    obs2, _, _ = e.reset()
    np.testing.assert_allclose(obs2[9:12], v_last - obs2[3:6], atol=1e-7)


def test_episode_terminates_and_is_deterministic(orc_lib):
    outs = []
    for _ in range(2):
        e = orc.OracleEnv(seed=99, lib=orc_lib)
        e.reset()
        rng = np.random.default_rng(5)
        n, done = 0, False
        while not done and n < 2000:
            o, r, done = e.do_step(rng.uniform(-1, 1, 12).astype(np.float32))
            n += 1
        outs.append((n, o.copy()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][0] < 1799  # random actions do not walk: fails on remaining_steps


def test_state_roundtrip(orc_lib):
    a, b = orc.OracleEnv(seed=3, lib=orc_lib), orc.OracleEnv(seed=4, lib=orc_lib)
    a.reset()
    rng = np.random.default_rng(1)
    for _ in range(20):
        a.do_step(rng.uniform(-1, 1, 12).astype(np.float32))
    b.set_state(a.get_state())
    act = rng.uniform(-1, 1, 12).astype(np.float32)
    oa, ra, da = a.do_step(act)
    ob, rb, db = b.do_step(act)
    np.testing.assert_allclose(oa, ob, atol=2e-5)
    assert da == db


def test_oracle_matches_its_committed_trace(orc_lib):
    """tests/golden/physics_trace.txt is a SELF-PIN (written by tests/diag/make_physics_trace.py from this same oracle): it does
    not add evidence about Bullet3, it makes a silent drift of the restatement impossible.  Tolerance: libm's sinf / atan2f /
    asinf may differ in the last place between glibc builds; contact-rich rollouts amplify that, so the early calls are
    compared tightly and the whole trace loosely."""
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "tests", "diag"))
    import make_physics_trace as mpt
    for mode, name in ((0, "physics_trace.txt"), (1, "physics_trace_selfcol.txt")):   # floor only / member-vs-member contacts
        ref = np.loadtxt(os.path.join(ROOT, "tests", "golden", name))
        got = mpt.trace(lib=orc_lib, self_collision=mode)
        assert got.shape == ref.shape == (mpt.N_ENV * mpt.N_STEP // mpt.EVERY, 10)
        assert np.array_equal(got[:, :3], ref[:, :3])                      # env, call, done: the episode structure is exact
        early = ref[:, 1] < 32
        np.testing.assert_allclose(got[early, 3:], ref[early, 3:], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(got[:, 3:], ref[:, 3:], rtol=5e-2, atol=5e-2)
