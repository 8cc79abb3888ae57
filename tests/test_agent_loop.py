"""The reference's Agent surface for PPO — act(state, reward), done(state, reward), whole episodes in a TrajectoryReplayBuffer,
train() every `train_every` episodes on padded trajectories — pinned to the COMPILED reference
(tests/golden/agent_loop_golden.txt, written by oracle/ref_loop.cpp driving the reference's own PpoGaeAgent over ten scripted
episodes: evo_motion_networks/src/agents/ppo_gae.cpp:29-115, src/replay_buffer.cpp:73-138,176-189).

CPU: the bookkeeping restatement (oracle/agent_oracle.py) around the torch mirrors of the networks and the autograd
restatement of train() reproduces every action, the buffer's shape after every done(), when it trains, what it stored and
the networks after the four train() calls."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import agent_oracle as ao  # noqa: E402
import golden_io  # noqa: E402

LOOP_GOLDEN = os.path.join(ROOT, "tests", "golden", "agent_loop_golden.txt")


def load_loop_golden():
    g = golden_io.load(LOOP_GOLDEN)
    lines = open(LOOP_GOLDEN).read().split("\n")
    g["config"] = [int(v) for v in next(l for l in lines if l.startswith("config ")).split()[1:]]
    g["lengths"] = [int(v) for v in next(l for l in lines if l.startswith("lengths ")).split()[2:]]
    g["pat_check"] = [float(v) for v in next(l for l in lines if l.startswith("pat_check ")).split()[1:]]
    g["done"] = [[int(v) for v in l.split()[1::2]] for l in lines if l.startswith("done ")]          # k, global_curr_step, memory, filtered, train
    g["sample"] = [[int(v) for v in l.split()[3:]] for l in lines if l.startswith("sample ")]         # memory positions, in draw order
    g["buffer"] = [[int(v) for v in l.split()[2:]] for l in lines if l.startswith("buffer ")]         # trajectory lengths after done()
    g["trains"] = int(next(l for l in lines if l.startswith("trains ")).split()[1])
    return g


def state_of(S, episode, t):
    return ao.pat(31 + episode, t * S + np.arange(S), 2.0)


def reward_of(episode, t):
    return float(ao.pat(77, np.array([100 * episode + t]), 1.0)[0])


def load_pattern(module, shapes, base):
    p = ao.pattern_params(shapes, base)
    with torch.no_grad():
        for name, prm in module.named_parameters():
            prm.copy_(torch.from_numpy(p[name]))


def shuffled_for(gold_sample, filtered_positions):
    """the recorded draw (memory positions) as the shuffled iota the buffer expects, completed by the positions not drawn"""
    idx = [filtered_positions.index(p) for p in gold_sample]
    rest = [i for i in range(len(filtered_positions) - 1) if i not in idx]
    return idx + rest


@pytest.fixture(scope="module")
def gold():
    return load_loop_golden()


def test_scripted_inputs_are_reproducible(gold):
    np.testing.assert_allclose([ao.pat(31, np.array([0]), 2.0)[0], ao.pat(40, np.array([1234]), 2.0)[0], ao.pat(77, np.array([99]), 1.0)[0]],
                               gold["pat_check"], rtol=0, atol=1e-7)
    assert gold["config"] == [371, 12, 256, 2, 3, 2, 5] and sum(gold["lengths"]) == gold["uniform"].shape[0] == 40


def test_act_done_check_train_match_the_compiled_reference(gold, hip_lib):
    from evomotion_amd import agent as agent_mod
    import torch_ref
    torch.set_num_threads(1)
    S, A, H, epoch, batch_size, train_every, replay = gold["config"]
    actor, critic = agent_mod.ActorModule([S], [A], H), agent_mod.CriticModule([S], H)
    load_pattern(actor, ao.ACTOR_SHAPES, 100)
    load_pattern(critic, ao.CRITIC_SHAPES, 200)
    oa, oc = torch.optim.Adam(actor.parameters(), lr=1e-3), torch.optim.Adam(critic.parameters(), lr=1e-3)

    def forward(state, u):
        actor.eval(); critic.eval()
        with torch.no_grad():
            x = torch.from_numpy(np.asarray(state, np.float32))
            mu, sigma = actor(x)
            uu = torch.from_numpy(u) if u is not None else torch.full((A,), 0.5)
            action = agent_mod.truncated_normal_sample(mu, sigma, u=uu)
            return action.numpy(), agent_mod.truncated_normal_log_pdf(action, mu, sigma).numpy(), critic(x).numpy()

    def train(b):
        t = lambda k: torch.from_numpy(b[k])
        torch_ref.ppo_train(actor, critic, oa, oc, t("states"), t("actions"), t("rewards"), t("done"), t("log_prob"), t("curr_values"),
                            t("next_values"), gamma=0.99, lam=0.95, epsilon=0.2, entropy_factor=0.01, critic_loss_factor=0.5,
                            epoch=epoch, clip_grad_norm=0.5)

    loop = ao.PpoLoopOracle(forward, train, batch_size, train_every, replay)
    k_act = n_train = 0
    for k, L in enumerate(gold["lengths"]):
        for t in range(L):
            a = loop.act(state_of(S, k, t), reward_of(k, t), gold["uniform"][k_act])
            np.testing.assert_allclose(a, gold["actions"][k_act], atol=3e-5, err_msg="episode %d step %d" % (k, t))
            k_act += 1
        gk, gstep, gmem, gfilt, gtrain = gold["done"][k]
        assert (gk, gstep, gmem, gfilt) == (k, loop.global_curr_step, len(loop.buffer.memory), len(loop.buffer.filtered_positions()))
        assert bool(gtrain) == loop.will_train()
        shuffled = shuffled_for(gold["sample"][n_train], loop.buffer.filtered_positions()) if gtrain else None
        trained = loop.done(state_of(S, k, L), reward_of(k, L), shuffled)
        assert trained == bool(gtrain)
        n_train += int(trained)
        assert [len(t) for t in loop.buffer.memory] == gold["buffer"][k]       # FIFO of `replay` trajectories, the open one last
    assert n_train == gold["trains"] == loop.curr_train_step == 4
    # what update_last left in the newest complete trajectory
    last = loop.buffer.memory[-2]
    np.testing.assert_allclose([s["reward"] for s in last], gold["last_rewards"], atol=1e-7)
    np.testing.assert_array_equal([1.0 if s["done"] else 0.0 for s in last], gold["last_done"])
    np.testing.assert_allclose(np.concatenate([np.ravel(s["curr_value"]) for s in last]), gold["last_values"], atol=2e-4)
    np.testing.assert_allclose(np.concatenate([np.ravel(s["next_value"]) for s in last]), gold["last_next_values"], atol=2e-4)
    np.testing.assert_allclose(np.stack([s["log_prob"] for s in last]), gold["last_log_prob"], atol=2e-3)
    # the networks after four train() calls of two epochs each (the value head is the ill-conditioned one: DESIGN.md §6)
    g0 = golden_io.load()
    x = torch.from_numpy(g0["X"])
    actor.eval(); critic.eval()
    with torch.no_grad():
        mu, sigma = actor(x)
        v = critic(x)
    np.testing.assert_allclose(mu.numpy(), gold["after_mu"], atol=2e-4)
    np.testing.assert_allclose(sigma.numpy(), gold["after_sigma"], atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(v.numpy(), gold["after_value"], atol=5e-3)
    np.testing.assert_allclose(actor.head[0].weight[0].detach().numpy(), gold["after_actor_w0_row0"], atol=2e-5)
    assert np.abs(mu.numpy() - g0["mu"]).max() > 1e-3      # it did train


def test_product_trajectory_buffer_follows_the_same_rules(hip_lib):
    """the product's own TrajectoryReplayBuffer (evomotion_amd/agent.py) against the restated one on random add / done traffic"""
    from evomotion_amd.agent import TrajectoryReplayBuffer
    rng = np.random.default_rng(3)
    prod, orc = TrajectoryReplayBuffer(6, seed=1), ao.TrajectoryBufferOracle(6)
    prod.shuffle = lambda index: list(reversed(index))
    prod.new_trajectory(); orc.new_trajectory()
    for step in range(300):
        r = rng.random()
        if r < 0.25:
            prod.new_trajectory(); orc.new_trajectory()
        else:
            st = dict(state=step, action=step, reward=0.0, done=False, log_prob=0.0, curr_value=0.0, next_value=0.0)
            prod.add(dict(st)); orc.add(dict(st))
            if rng.random() < 0.5:
                prod.update_last(float(step), bool(step % 3 == 0), 1.5); orc.update_last(float(step), bool(step % 3 == 0), 1.5)
        assert [[s["state"] for s in t] for t in prod.memory] == [[s["state"] for s in t] for t in orc.memory]
        assert [[(s["reward"], s["done"], s["next_value"]) for s in t] for t in prod.memory] == [[(s["reward"], s["done"], s["next_value"]) for s in t] for t in orc.memory]
        for bs in (1, 3, 5):
            assert prod.enough_trajectory(bs) == orc.enough_trajectory(bs)
            nf = len(orc.filtered_positions())
            if nf >= 1:
                got = [[s["state"] for s in t] for t in prod.sample(bs)]
                want = [[s["state"] for s in t] for t in orc.sample(bs, list(reversed(range(nf - 1))))]
                assert got == want
