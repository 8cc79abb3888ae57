"""The reference's Agent surface for SAC — act(state, reward), done(state, reward), the flat ReplayBuffer, train() `epoch` times
every `train_every` act() calls — pinned to the COMPILED reference (tests/golden/sac_loop_golden.txt, written by
oracle/ref_sac_loop.cpp driving the reference's own SoftActorCriticAgent over six scripted episodes:
evo_motion_networks/src/agents/soft_actor_critic.cpp:47-91,172-180, src/replay_buffer.cpp:16-52,146-153).

CPU: the bookkeeping restatement (oracle/agent_oracle.py) around the torch mirrors of the networks and the autograd restatement of
train() reproduces every action, the buffer after every call (including the reference's rewrite of a terminal transition by the
next episode's first act()), when it trains, which transitions it draws and the networks after the ten train() calls."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import agent_oracle as ao  # noqa: E402
import golden_io  # noqa: E402

SAC_LOOP_GOLDEN = os.path.join(ROOT, "tests", "golden", "sac_loop_golden.txt")


def load_sac_loop_golden():
    g = golden_io.load(SAC_LOOP_GOLDEN)
    lines = open(SAC_LOOP_GOLDEN).read().split("\n")
    g["config"] = [int(v) for v in next(l for l in lines if l.startswith("config ")).split()[1:]]
    g["lengths"] = [int(v) for v in next(l for l in lines if l.startswith("lengths ")).split()[2:]]
    g["act"] = [[int(v) for v in l.split()[1:3]] + [int(v) for v in l.split()[4::2]] for l in lines if l.startswith("act ")]  # k, t, global_curr_step, size_after, train
    g["sample"] = [[int(v) for v in l.split()[3:]] for l in lines if l.startswith("sample ")]
    g["buffer"] = {}
    for l in lines:
        if l.startswith("buffer "):
            w = l.split()
            items = [tuple(float(x) for x in m) for m in re.findall(r"\((-?\d+),(-?[0-9.e+-]+),(\d),(-?\d+)\)", l)]
            g["buffer"][(w[1], int(w[2]), int(w[3]))] = items
    g["trains"] = int(next(l for l in lines if l.startswith("trains ")).split()[1])
    return g


def state_of(S, episode, t):
    x = ao.pat(31 + episode, t * S + np.arange(S), 2.0)
    x[0] = np.float32(100 * episode + t) / np.float32(1000.0)
    return x


def reward_of(episode, t):
    return float(ao.pat(77, np.array([100 * episode + t]), 1.0)[0])


def shuffled_for(gold_sample, n):
    """the recorded draw (memory positions) completed to a permutation of range(n)"""
    return list(gold_sample) + [i for i in range(n) if i not in gold_sample]


def buffer_dump(memory):
    tag = lambda s: float(int(round(float(np.asarray(s).reshape(-1)[0]) * 1000.0)))
    return [(tag(i["state"]), float(np.float32(i["reward"])), 1.0 if i["done"] else 0.0, tag(i["next_state"])) for i in memory]


def assert_buffer(memory, want, where):
    got = buffer_dump(memory)
    assert len(got) == len(want), where
    for g_, w_ in zip(got, want):
        assert g_[0] == w_[0] and g_[2] == w_[2] and g_[3] == w_[3] and abs(g_[1] - w_[1]) < 1e-6, (where, g_, w_)


@pytest.fixture(scope="module")
def gold():
    return load_sac_loop_golden()


def test_the_reference_rewrites_a_terminal_transition_at_the_next_act(gold):
    """what the fixture says about the reference itself: after done() the newest element carries done = 1, and the first act() of
    the next episode turns it back into done = 0 with the new episode's first state as next_state"""
    after_done = gold["buffer"][("done", 0, 4)]
    after_act = gold["buffer"][("act", 1, 0)]
    assert after_done[-1][2] == 1.0 and after_done[-1][3] == 4.0
    assert after_act[-2][0] == after_done[-1][0] and after_act[-2][2] == 0.0 and after_act[-2][3] == 100.0
    for key, items in gold["buffer"].items():
        assert all(i[2] == 0.0 for i in items[:-1]), key   # a done flag only ever sits on the newest (never sampled) element


def test_act_done_check_train_match_the_compiled_reference(gold, hip_lib):
    from evomotion_amd import agent as agent_mod
    from test_sac_host import build
    from torch_ref import sac_train
    torch.set_num_threads(1)
    S, A, H, batch_size, epoch, replay, train_every = gold["config"]
    actor, (c1, c2, t1, t2), ent = build()
    opts = [torch.optim.Adam(m.parameters(), lr=1e-3) for m in (actor, c1, c2, ent)]

    def forward(state, u):
        actor.eval()
        with torch.no_grad():
            mu, sigma = actor(torch.from_numpy(np.asarray(state, np.float32)))
            return agent_mod.truncated_normal_sample(mu, sigma, u=torch.from_numpy(u)).numpy()

    def train(b, u_next, u_curr):
        t = lambda k: torch.from_numpy(b[k])
        for m in (actor, c1, c2, t1, t2):
            m.train()
        sac_train(actor, c1, c2, t1, t2, ent, opts[0], opts[1], opts[2], opts[3], t("states"), t("actions"), t("rewards"), t("done"),
                  t("next_states"), gamma=0.99, tau=0.005, target_entropy=-float(A), u_next=torch.from_numpy(u_next), u_curr=torch.from_numpy(u_curr))

    loop = ao.SacLoopOracle(forward, train, batch_size, epoch, train_every, replay)
    k_act = n_train = 0
    for k, L in enumerate(gold["lengths"]):
        for t in range(L):
            gk, gt, gstep, gsize, gtrain = gold["act"][k_act]
            assert (gk, gt, gstep) == (k, t, loop.global_curr_step)
            shuffles = [shuffled_for(gold["sample"][n_train + e], gsize - 1) for e in range(epoch)] if gtrain else ()
            us = [(gold["train_u_next"][n_train + e], gold["train_u_curr"][n_train + e]) for e in range(epoch)] if gtrain else ()
            a, trained = loop.act(state_of(S, k, t), reward_of(k, t), gold["uniform"][k_act], shuffles, us)
            assert trained == (epoch if gtrain else 0) and len(loop.buffer.memory) == gsize
            n_train += trained
            np.testing.assert_allclose(a, gold["actions"][k_act], atol=5e-5, err_msg="episode %d step %d" % (k, t))
            assert_buffer(loop.buffer.memory, gold["buffer"][("act", k, t)], ("act", k, t))
            k_act += 1
        loop.done(state_of(S, k, L), reward_of(k, L))
        assert_buffer(loop.buffer.memory, gold["buffer"][("done", k, L)], ("done", k, L))
    assert n_train == gold["trains"] == loop.curr_train_step == 10 and loop.global_curr_step == 19
    g0 = golden_io.load(os.path.join(ROOT, "tests", "golden", "sac_golden.txt"))
    x, a = torch.from_numpy(g0["sac_states"]), torch.from_numpy(g0["sac_actions"])
    for m in (actor, c1, c2, t1, t2):
        m.eval()
    with torch.no_grad():
        mu, sigma = actor(x)
        np.testing.assert_allclose(mu.numpy(), gold["after_mu"], atol=3e-4)
        np.testing.assert_allclose(sigma.numpy(), gold["after_sigma"], atol=3e-4, rtol=3e-4)
        np.testing.assert_allclose(c1(x, a).numpy(), gold["after_q1"], atol=2e-3)
        np.testing.assert_allclose(c2(x, a).numpy(), gold["after_q2"], atol=2e-3)
        np.testing.assert_allclose(t1(x, a).numpy(), gold["after_tq1"], atol=5e-4)
        np.testing.assert_allclose(t2(x, a).numpy(), gold["after_tq2"], atol=5e-4)
        np.testing.assert_allclose(ent.log_alpha.numpy(), gold["after_log_alpha"], atol=1e-5)
    assert abs(float(gold["after_log_alpha"][0])) > 5e-3   # it did train
