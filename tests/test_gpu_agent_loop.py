"""The product's PpoGaeAgent (evomotion_amd/agent.py: the reference's Agent surface — act / done / check_train over whole episodes
— around the fused HIP forward and the HIP trainer behind evm_ppo_*) against the COMPILED reference's own run of the same ten
scripted episodes (tests/golden/agent_loop_golden.txt, oracle/ref_loop.cpp; evo_motion_networks/src/agents/ppo_gae.cpp:29-115,
src/replay_buffer.cpp:73-138,176-189).  The uniform draws of truncated_normal_sample and the buffer's shuffles are the ones
the reference's generators produced (recorded in the fixture)."""
import numpy as np
import pytest

import test_agent_loop as tl

pytestmark = pytest.mark.gpu


def test_product_agent_reproduces_the_reference_episode_loop():
    import torch
    from evomotion_amd.agent import PpoGaeAgent
    gold = tl.load_loop_golden()
    S, A, H, epoch, batch_size, train_every, replay = gold["config"]
    agent = PpoGaeAgent(1234, [S], [A], hidden_size=H, gamma=0.99, lam=0.95, epsilon=0.2, entropy_factor=0.01, critic_loss_factor=0.5,
                        epoch=epoch, batch_size=batch_size, train_every=train_every, replay_buffer_size=replay, learning_rate=1e-3,
                        clip_grad_norm=0.5, device=0)
    tl.load_pattern(agent.actor, tl.ao.ACTOR_SHAPES, 100)
    tl.load_pattern(agent.critic, tl.ao.CRITIC_SHAPES, 200)
    agent.fused.load_modules(agent.actor, agent.critic)
    draws = []

    default_shuffle = agent.replay_buffer.shuffle     # std::mt19937(seed) + std::shuffle restated (evomotion_amd/stdrandom.py): nothing plugged in

    def spy(index):
        filtered = [i for i, t in enumerate(agent.replay_buffer.memory) if len(t) > 1]
        assert len(index) == len(filtered) - 1
        order = default_shuffle(index)
        assert [filtered[i] for i in order[:batch_size]] == gold["sample"][len(draws)]      # the trajectories the reference drew
        draws.append(order)
        return order
    agent.replay_buffer.shuffle = spy
    k_act = 0
    worst_action = 0.0
    for k, L in enumerate(gold["lengths"]):
        for t in range(L):
            a = agent.act(torch.from_numpy(tl.state_of(S, k, t)), tl.reward_of(k, t), uniform=torch.from_numpy(gold["uniform"][k_act]))
            assert a.shape == (A,)
            worst_action = max(worst_action, float(np.abs(a.cpu().numpy() - gold["actions"][k_act]).max()))
            k_act += 1
        gk, gstep, gmem, gfilt, gtrain = gold["done"][k]
        mem = agent.replay_buffer.memory
        assert (gstep, gmem, gfilt) == (agent.global_curr_step, len(mem), sum(len(t) > 1 for t in mem))
        before = agent.curr_train_step
        agent.done(torch.from_numpy(tl.state_of(S, k, L)), tl.reward_of(k, L))
        assert agent.curr_train_step - before == gtrain
        assert [len(t) for t in agent.replay_buffer.memory] == gold["buffer"][k]
    assert agent.curr_train_step == gold["trains"] == len(draws) == 4
    # actions: every train() moves the weights, so the later episodes also check the trained networks
    assert worst_action < 2e-4, worst_action
    last = agent.replay_buffer.memory[-2]
    np.testing.assert_allclose([s["reward"] for s in last], gold["last_rewards"], atol=1e-7)
    np.testing.assert_array_equal([1.0 if s["done"] else 0.0 for s in last], gold["last_done"])
    np.testing.assert_allclose(torch.stack([s["curr_value"] for s in last]).cpu().numpy(), gold["last_values"], atol=5e-3)
    np.testing.assert_allclose(torch.stack([s["next_value"] for s in last]).cpu().numpy(), gold["last_next_values"], atol=5e-3)
    np.testing.assert_allclose(torch.stack([s["log_prob"] for s in last]).cpu().numpy(), gold["last_log_prob"], atol=5e-3)
    g0 = tl.golden_io.load()
    x = torch.from_numpy(g0["X"]).cuda()
    _, _, value, mu, sigma = agent.fused.forward(x, uniform=torch.full((8, A), 0.5, device="cuda"), want_dist=True)
    np.testing.assert_allclose(mu.cpu().numpy(), gold["after_mu"], atol=3e-4)
    np.testing.assert_allclose(sigma.cpu().numpy(), gold["after_sigma"], atol=3e-4, rtol=3e-4)
    np.testing.assert_allclose(value.cpu().numpy(), gold["after_value"].ravel(), atol=1e-2)   # (ill-conditioned head: DESIGN.md §6)
    agent.sync_modules()
    np.testing.assert_allclose(agent.actor.head[0].weight[0].detach().cpu().numpy(), gold["after_actor_w0_row0"], atol=3e-5)
    meters = agent.get_metrics()
    assert [m.name for m in meters] == ["actor_loss", "critic_loss", "steps"]                   # ppo_gae.cpp:205-207
    assert meters[2].values == [float(v) for v in gold["lengths"]]
    assert len(meters[0].values) == len(meters[1].values) == gold["trains"] * epoch               # one add per epoch (:185-186)


def test_vector_form_keeps_one_trajectory_per_env():
    """n_envs > 1: every env fills its own open trajectory of the one buffer; an env's done() closes only its own"""
    import torch
    from evomotion_amd.agent import PpoGaeAgent
    n = 3
    agent = PpoGaeAgent(7, [371], [12], epoch=1, batch_size=2, train_every=2, replay_buffer_size=16, n_envs=n)
    g = torch.Generator().manual_seed(0)
    lens = [0] * n
    for t in range(9):
        a = agent.act(torch.rand(n, 371, generator=g), torch.rand(n, generator=g))
        assert a.shape == (n, 12) and bool(torch.isfinite(a).all())
        for e in range(n):
            lens[e] += 1
        if t in (3, 6):
            e = t % n
            agent.done(torch.rand(371, generator=g), 0.5, env=e)
            assert agent._open[e] is None and len([tr for tr in agent.replay_buffer.memory if tr and tr[-1]["done"]]) == (1 if t == 3 else 2)
            lens[e] = 0
    open_lens = sorted(len(agent._open[e]) for e in range(n))
    assert open_lens == sorted(lens)
    assert agent.global_curr_step == 2 and agent.curr_train_step in (0, 1)


def test_vector_form_trains_on_finished_episodes_only():
    """n_envs > 1: open trajectories live outside the buffer, so whatever sample() picks has ended (its last step
    carries done = True and the real reward, not the placeholders of a running episode), the buffer never evicts a trajectory an
    environment still appends to, and the episode that has just ended is the one sample() leaves out, as in the reference."""
    import torch
    from evomotion_amd.agent import PpoGaeAgent
    n = 5
    agent = PpoGaeAgent(11, [371], [12], epoch=1, batch_size=3, train_every=1, replay_buffer_size=4, n_envs=n)
    sampled = []
    inner = agent.replay_buffer.sample

    def spy(batch_size):
        out = inner(batch_size)
        sampled.append(out)
        return out
    agent.replay_buffer.sample = spy
    g = torch.Generator().manual_seed(3)
    finished = 0
    for t in range(40):
        agent.act(torch.rand(n, 371, generator=g), torch.rand(n, generator=g))
        if t % 3 == 2:
            e = (t // 3) % n
            open_before = [id(agent._open[k]) for k in range(n) if k != e and agent._open[k] is not None]
            just_ended = agent._open[e]
            agent.done(torch.rand(371, generator=g), 1.25, env=e)
            finished += 1
            assert len(agent.replay_buffer.memory) == min(finished, 4)                      # FIFO of finished episodes
            assert all(tr[-1]["done"] is True and tr[-1]["reward"] == 1.25 for tr in agent.replay_buffer.memory)
            assert [id(agent._open[k]) for k in range(n) if k != e and agent._open[k] is not None] == open_before   # untouched, not evicted
            if sampled and sampled[-1] is not None:
                assert all(tr is not just_ended for tr in sampled[-1])
    assert agent.curr_train_step >= 5 and len(sampled) == agent.curr_train_step
    for batch in sampled:
        assert 1 <= len(batch) <= 3
        for tr in batch:
            assert len(tr) > 1 and tr[-1]["done"] is True and all(not s["done"] for s in tr[:-1])
