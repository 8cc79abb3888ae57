"""The agent-side oracle (oracle/agent_oracle.py) against golden vectors produced by the REFERENCE's own compiled
evo_motion_networks code (oracle/ref_build.sh + ref_golden.cpp -> tests/golden/agent_golden.txt)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import agent_oracle as ao  # noqa: E402
import golden_io  # noqa: E402


@pytest.fixture(scope="module")
def gold():
    return golden_io.load()


def test_parameter_inventory(gold):
    got = [(n, s) for who, n, s in gold["_params"] if who == "actor"]
    assert got == ao.ACTOR_SHAPES
    got = [(n, s) for who, n, s in gold["_params"] if who == "critic"]
    assert got == ao.CRITIC_SHAPES
    n_actor = sum(int(np.prod(s)) for _, s in ao.ACTOR_SHAPES)
    n_critic = sum(int(np.prod(s)) for _, s in ao.CRITIC_SHAPES)
    assert (n_actor, n_critic, n_actor + n_critic) == (168216, 162305, 330521)


def test_pattern_inputs_reproduce(gold):
    x = ao.pat(7, np.arange(8 * 371), 2.0).reshape(8, 371)
    assert np.array_equal(x, gold["X"])


def test_actor_critic_forward(gold):
    pa, pc = ao.pattern_params(ao.ACTOR_SHAPES, 100), ao.pattern_params(ao.CRITIC_SHAPES, 200)
    mu, sigma = ao.actor_forward(gold["X"], pa)
    v = ao.critic_forward(gold["X"], pc)
    np.testing.assert_allclose(mu, gold["mu"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(sigma, gold["sigma"], atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(v, gold["value"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(ao.actor_forward(gold["X"][0], pa)[0], gold["mu_1d"], atol=2e-5)
    assert (np.abs(gold["mu"]) <= 1).all() and (gold["sigma"] > 0).all()  # the reference's own test_linear_modules asserts


def test_truncated_normal(gold):
    m, s, x, u = gold["tn_mu"], gold["tn_sigma"], gold["tn_x"], gold["tn_u"]
    lp, en, sm = ao.tn_log_pdf(x, m, s), ao.tn_entropy(m, s), ao.tn_sample(m, s, u)
    ok = np.isfinite(gold["tn_log_pdf"])
    assert np.array_equal(ok, np.isfinite(lp))
    np.testing.assert_allclose(lp[ok], gold["tn_log_pdf"][ok], rtol=2e-5, atol=2e-4)
    ok = np.isfinite(gold["tn_entropy"])
    assert np.array_equal(ok, np.isfinite(en))
    np.testing.assert_allclose(en[ok], gold["tn_entropy"][ok], rtol=2e-5, atol=2e-4)
    # sample: inverse-CDF through erfinv; wide tolerance only where cdf saturates near 0/1
    assert (np.abs(gold["tn_sample"]) <= 1).all()
    np.testing.assert_allclose(sm, gold["tn_sample"], atol=2e-4)
