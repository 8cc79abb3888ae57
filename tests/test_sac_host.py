"""SAC host mirrors (evomotion_amd/sac.py) and the numpy Q oracle against the reference's golden vectors
(tests/golden/sac_golden.txt, produced by oracle/ref_sac.cpp from the reference's compiled library)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import agent_oracle as ao  # noqa: E402
import golden_io  # noqa: E402

SAC_GOLDEN = os.path.join(ROOT, "tests", "golden", "sac_golden.txt")


@pytest.fixture(scope="module")
def gold():
    g = golden_io.load(SAC_GOLDEN)
    g["_scalars"] = {l.split()[1]: float(l.split()[2]) for l in open(SAC_GOLDEN) if l.startswith("scalar ")}
    g["_replay"] = [l.rstrip("\n") for l in open(SAC_GOLDEN) if l.startswith("replay ")]
    return g


def load_pattern(module, shapes, base):
    p = ao.pattern_params(shapes, base)
    with torch.no_grad():
        for n, t in module.named_parameters():
            t.copy_(torch.from_numpy(p[n]))
    return p


def build():
    from evomotion_amd.agent import ActorModule
    from evomotion_amd.sac import EntropyParameter, QNetworkModule
    torch.manual_seed(0)
    actor = ActorModule([371], [12], 256)
    qs = [QNetworkModule([371], [12], 256) for _ in range(4)]
    load_pattern(actor, ao.ACTOR_SHAPES, 100)
    for q, base in zip(qs, (300, 400, 500, 600)):
        load_pattern(q, ao.Q_SHAPES, base)
    return actor, qs, EntropyParameter(1.0, 1)


def test_q_network_names_counts_and_forward(gold):
    from evomotion_amd.agent import count_parameters
    actor, qs, ent = build()
    ref_order = [(n, s) for who, n, s in gold["_params"] if who == "q"]
    assert [(n, tuple(p.shape)) for n, p in qs[0].named_parameters()] == ref_order == [(n, s) for n, s in ao.Q_SHAPES]
    assert [(n, tuple(p.shape)) for n, p in ent.named_parameters()] == [("log_alpha", (1,))]
    assert count_parameters(actor, *qs, ent) == int(gold["_scalars"]["count_parameters"]) == 1094941
    x, a = torch.from_numpy(gold["sac_states"]), torch.from_numpy(gold["sac_actions"])
    with torch.no_grad():
        q1, tq2 = qs[0](x, a), qs[3](x, a)
        q1_1d = qs[0](x[0], a[0])
    np.testing.assert_allclose(q1.numpy(), gold["q1_before"], atol=3e-5)
    np.testing.assert_allclose(tq2.numpy(), gold["tq2_before"], atol=3e-5)
    np.testing.assert_allclose(q1_1d.numpy(), gold["q1_before_1d"], atol=3e-5)
    # numpy oracle of the Q network (used by the GPU parity tests)
    np.testing.assert_allclose(ao.q_forward(gold["sac_states"], gold["sac_actions"], ao.pattern_params(ao.Q_SHAPES, 300)),
                               gold["q1_before"], atol=3e-5)


def test_one_train_call_reproduces_the_reference(gold):
    from torch_ref import sac_train
    actor, (c1, c2, t1, t2), ent = build()
    opts = [torch.optim.Adam(m.parameters(), lr=1e-3) for m in (actor, c1, c2, ent)]
    t = lambda k: torch.from_numpy(gold[k])
    for m in (actor, c1, c2, t1, t2):
        m.train()
    out = sac_train(actor, c1, c2, t1, t2, ent, opts[0], opts[1], opts[2], opts[3], t("sac_states"), t("sac_actions"),
                    t("sac_rewards"), t("sac_done"), t("sac_next_states"), gamma=0.99, tau=0.005,
                    target_entropy=gold["_scalars"]["target_entropy"], u_next=t("sac_u_next"), u_curr=t("sac_u_curr"))
    sc = gold["_scalars"]
    assert abs(float(out["critic_1"]) - sc["loss_critic_1"]) < 2e-4 * abs(sc["loss_critic_1"])
    assert abs(float(out["critic_2"]) - sc["loss_critic_2"]) < 2e-4 * abs(sc["loss_critic_2"])
    assert abs(float(out["actor"]) - sc["loss_actor"]) < 2e-4 * abs(sc["loss_actor"])
    assert abs(float(out["entropy"]) - sc["loss_entropy"]) < 1e-6
    x, a = t("sac_states"), t("sac_actions")
    for m in (actor, c1, c2, t1, t2):
        m.eval()
    with torch.no_grad():
        mu, sigma = actor(x)
        np.testing.assert_allclose(mu.numpy(), gold["after_mu"], atol=1e-4)
        np.testing.assert_allclose(sigma.numpy(), gold["after_sigma"], atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(c1(x, a).numpy(), gold["after_q1"], atol=2e-4)
        np.testing.assert_allclose(c2(x, a).numpy(), gold["after_q2"], atol=2e-4)
        np.testing.assert_allclose(t1(x, a).numpy(), gold["after_tq1"], atol=1e-4)   # soft update only
        np.testing.assert_allclose(t2(x, a).numpy(), gold["after_tq2"], atol=1e-4)
        np.testing.assert_allclose(ent.log_alpha.numpy(), gold["after_log_alpha"], atol=1e-6)
    # the update moved things: the "after" vectors differ from the "before" ones by far more than the tolerance
    assert np.abs(gold["after_q1"] - gold["q1_before"]).max() > 1e-2


def test_generator_equivalence_of_the_recorded_draws(gold):
    """The reference draws with at::rand after at::manual_seed(777); torch.rand reproduces those draws, so a seeded
    update needs no recorded uniforms."""
    torch.manual_seed(777)
    assert np.array_equal(torch.rand(8, 12).numpy(), gold["sac_u_next"])
    assert np.array_equal(torch.rand(8, 12).numpy(), gold["sac_u_curr"])
