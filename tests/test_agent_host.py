"""Host-side agent logic on CPU torch against the reference's golden vectors: module mirrors, truncated normal,
GAE and one full PpoGaeAgent::train() call (loss, clip-grad-norm, Adam)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import agent_oracle as ao  # noqa: E402
import golden_io  # noqa: E402


@pytest.fixture(scope="module")
def gold():
    return golden_io.load()


@pytest.fixture(scope="module")
def agent_mod(hip_lib):
    from evomotion_amd import agent
    return agent


def load_pattern(module, shapes, base):
    p = ao.pattern_params(shapes, base)
    with torch.no_grad():
        for name, t in module.named_parameters():
            t.copy_(torch.from_numpy(p[name]))
    return p


def test_module_mirrors_match_reference(gold, agent_mod):
    torch.set_num_threads(1)
    actor = agent_mod.ActorModule([371], [12], 256)
    critic = agent_mod.CriticModule([371], 256)
    assert [n for n, _ in actor.named_parameters()] == [n for n, _ in ao.ACTOR_SHAPES]
    assert [n for n, _ in critic.named_parameters()] == [n for n, _ in ao.CRITIC_SHAPES]
    assert agent_mod.count_parameters(actor, critic) == 330521
    load_pattern(actor, ao.ACTOR_SHAPES, 100)
    load_pattern(critic, ao.CRITIC_SHAPES, 200)
    actor.eval(); critic.eval()
    x = torch.from_numpy(gold["X"])
    with torch.no_grad():
        mu, sigma = actor(x)
        v = critic(x)
    np.testing.assert_allclose(mu.numpy(), gold["mu"], atol=1e-6)
    np.testing.assert_allclose(sigma.numpy(), gold["sigma"], atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(v.numpy(), gold["value"], atol=2e-6)


def test_init_weights_statistics(agent_mod):
    torch.manual_seed(0)
    a = agent_mod.ActorModule([371], [12], 256)
    w = a.head[0].weight
    assert abs(float(w.std()) - 0.1 * (2.0 / (371 + 256)) ** 0.5) < 2e-4  # xavier_normal gain 0.1 (init.cpp:11)
    assert abs(float(a.head[0].bias.std()) - 0.1) < 0.02
    assert torch.equal(a.head[2].weight, torch.ones(256)) and torch.equal(a.head[2].bias, torch.zeros(256))


def test_truncated_normal_torch(gold, agent_mod):
    m, s, x, u = [torch.from_numpy(gold[k]) for k in ("tn_mu", "tn_sigma", "tn_x", "tn_u")]
    lp = agent_mod.truncated_normal_log_pdf(x, m, s).numpy()
    en = agent_mod.truncated_normal_entropy(m, s).numpy()
    sm = agent_mod.truncated_normal_sample(m, s, u=u).numpy()
    ok = np.isfinite(gold["tn_log_pdf"])
    np.testing.assert_allclose(lp[ok], gold["tn_log_pdf"][ok], rtol=1e-6, atol=1e-6)
    ok = np.isfinite(gold["tn_entropy"])
    np.testing.assert_allclose(en[ok], gold["tn_entropy"][ok], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sm, gold["tn_sample"], atol=1e-6)
    # the reference's own test_functions.cpp assertions: samples inside the bounds and finite
    assert np.isfinite(sm).all() and (sm >= -1).all() and (sm <= 1).all()


def test_gae_matches_numpy_oracle(gold, agent_mod):
    r, d, cv, nv = [torch.from_numpy(gold[k]) for k in ("ppo_rewards", "ppo_done", "ppo_curr_values", "ppo_next_values")]
    import torch_ref
    mask, adv, ret = torch_ref.gae_advantages(r, d, cv, nv, 0.99, 0.95)
    m2, a2, r2 = ao.gae(gold["ppo_rewards"], gold["ppo_done"], gold["ppo_curr_values"], gold["ppo_next_values"], 0.99, 0.95)
    assert np.array_equal(mask.numpy(), m2)
    np.testing.assert_allclose(adv.numpy(), a2, atol=2e-6)
    np.testing.assert_allclose(ret.numpy(), r2, atol=2e-6)
    # explicit transition mask == the reference's shifted-done mask on trailing padding
    mask3, adv3, _ = torch_ref.gae_advantages(r, d, cv, nv, 0.99, 0.95, mask=mask)
    np.testing.assert_allclose(adv3.numpy(), adv.numpy(), atol=1e-7)


def test_one_train_call_matches_reference(gold, agent_mod):
    """PpoGaeAgent::train on a padded [4,7] batch, epoch = 2: GAE + clipped surrogate + entropy + clip-grad + Adam."""
    torch.set_num_threads(1)
    actor = agent_mod.ActorModule([371], [12], 256)
    critic = agent_mod.CriticModule([371], 256)
    load_pattern(actor, ao.ACTOR_SHAPES, 100)
    load_pattern(critic, ao.CRITIC_SHAPES, 200)
    oa = torch.optim.Adam(actor.parameters(), lr=1e-3)
    oc = torch.optim.Adam(critic.parameters(), lr=1e-3)
    t = lambda k: torch.from_numpy(gold[k])
    import torch_ref
    torch_ref.ppo_train(actor, critic, oa, oc, t("ppo_states"), t("ppo_actions"), t("ppo_rewards"), t("ppo_done"),
                        t("ppo_log_prob"), t("ppo_curr_values"), t("ppo_next_values"), gamma=0.99, lam=0.95, epsilon=0.2,
                        entropy_factor=0.01, critic_loss_factor=0.5, epoch=2, clip_grad_norm=0.5)
    actor.eval(); critic.eval()
    x = torch.from_numpy(gold["X"])
    with torch.no_grad():
        mu, sigma = actor(x)
        v = critic(x)
    np.testing.assert_allclose(mu.numpy(), gold["ppo_after_mu"], atol=2e-5)
    np.testing.assert_allclose(sigma.numpy(), gold["ppo_after_sigma"], atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(v.numpy(), gold["ppo_after_value"], atol=5e-5)
    np.testing.assert_allclose(actor.head[0].weight[0].detach().numpy(), gold["ppo_after_actor_w0_row0"], atol=2e-6)
    # and it really moved (lr 1e-3, 2 Adam steps)
    assert np.abs(mu.numpy() - gold["mu"]).max() > 1e-4


def test_random_agent_matches_reference_golden(gold, agent_mod):
    """RandomAgent::act (debug_agents.cpp:28-30): first three actions after manual_seed(1234), recorded from the compiled
    reference by oracle/ref_golden.cpp (-0.9420, -0.1962, -0.4803, ...).  Same generator, same draws: exact."""
    torch.manual_seed(1234)
    ra = agent_mod.RandomAgent([12], "cpu")
    state = torch.zeros(371)
    acts = torch.stack([ra.act(state, 0.0) for _ in range(3)]).numpy()
    assert acts.shape == (3, 12)
    np.testing.assert_array_equal(acts, gold["random_agent_actions"])
    assert abs(float(acts[0, 0]) + 0.9420) < 1e-4 and abs(float(acts[0, 2]) + 0.4803) < 1e-4
    # interface of DebugAgent (debug_agents.cpp:7-23): no parameters, no metrics, no-op done/save/load
    assert ra.count_parameters() == 0 and ra.get_metrics() == []
    ra.done(state, 0.0); ra.save("/nonexistent"); ra.load("/nonexistent"); ra.set_eval(True)
    # batched form: [N, A] in [-1, 1), private generator reproducible
    a = agent_mod.RandomAgent([12], "cpu", seed=7).act(torch.zeros(64, 371))
    b = agent_mod.RandomAgent([12], "cpu", seed=7).act(torch.zeros(64, 371))
    assert a.shape == (64, 12) and torch.equal(a, b) and float(a.min()) >= -1.0 and float(a.max()) < 1.0
