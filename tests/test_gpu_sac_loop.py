"""The product's SoftActorCriticAgent (evomotion_amd/sac.py: the reference's Agent surface — act / done / check_train over a flat
ReplayBuffer — around the fused HIP actor forward and the HIP SAC update) against the COMPILED reference's own run of the same six
scripted episodes (tests/golden/sac_loop_golden.txt, oracle/ref_sac_loop.cpp; evo_motion_networks/src/agents/
soft_actor_critic.cpp:47-91,172-180, src/replay_buffer.cpp:16-52,146-153).  The uniform draws of truncated_normal_sample (act and
both draws of every train()) and the buffer's shuffles are the ones the reference's generators produced (recorded in the fixture)."""
import numpy as np
import pytest

import test_sac_loop as tl

pytestmark = pytest.mark.gpu


def test_product_sac_agent_reproduces_the_reference_episode_loop():
    import torch
    from evomotion_amd.sac import SoftActorCriticAgent
    from test_sac_host import load_pattern
    gold = tl.load_sac_loop_golden()
    S, A, H, batch_size, epoch, replay, train_every = gold["config"]
    agent = SoftActorCriticAgent(1234, [S], [A], actor_hidden_size=H, critic_hidden_size=H, batch_size=batch_size, epoch=epoch,
                                 learning_rate=1e-3, gamma=0.99, tau=0.005, replay_buffer_size=replay, train_every=train_every, device=0)
    load_pattern(agent.actor, tl.ao.ACTOR_SHAPES, 100)
    for q, base in zip((agent.critic_1, agent.critic_2, agent.target_critic_1, agent.target_critic_2), (300, 400, 500, 600)):
        load_pattern(q, tl.ao.Q_SHAPES, base)
    agent._push_actor()
    agent._push_critics()
    draws = []

    default_shuffle = agent.replay_buffer.shuffle     # std::mt19937(seed) + std::shuffle restated (evomotion_amd/stdrandom.py): nothing plugged in

    def spy(index):
        order = default_shuffle(index)
        assert order[:batch_size] == gold["sample"][len(draws)]                               # the transitions the reference drew
        draws.append(order)
        return order
    agent.replay_buffer.shuffle = spy
    mem_np = lambda: [dict(state=i["state"].cpu().numpy(), reward=i["reward"], done=i["done"], next_state=i["next_state"].cpu().numpy())
                      for i in agent.replay_buffer.memory]
    k_act = n_train = 0
    worst = 0.0
    for k, L in enumerate(gold["lengths"]):
        for t in range(L):
            gk, gt, gstep, gsize, gtrain = gold["act"][k_act]
            assert gstep == agent.global_curr_step
            us = [(torch.from_numpy(gold["train_u_next"][n_train + e]), torch.from_numpy(gold["train_u_curr"][n_train + e])) for e in range(epoch)] if gtrain else None
            before = agent.curr_train_step
            a = agent.act(torch.from_numpy(tl.state_of(S, k, t)), tl.reward_of(k, t), uniform=torch.from_numpy(gold["uniform"][k_act]), train_uniforms=us)
            assert a.shape == (A,) and agent.curr_train_step - before == (epoch if gtrain else 0)
            n_train += agent.curr_train_step - before
            worst = max(worst, float(np.abs(a.cpu().numpy() - gold["actions"][k_act]).max()))
            tl.assert_buffer(mem_np(), gold["buffer"][("act", k, t)], ("act", k, t))
            k_act += 1
        agent.done(torch.from_numpy(tl.state_of(S, k, L)), tl.reward_of(k, L))
        tl.assert_buffer(mem_np(), gold["buffer"][("done", k, L)], ("done", k, L))
    assert n_train == gold["trains"] == len(draws) == 10 and agent.global_curr_step == 19
    # every train() moves the weights, so the later actions also check the trained actor
    assert worst < 3e-4, worst
    g0 = tl.golden_io.load(tl.os.path.join(tl.ROOT, "tests", "golden", "sac_golden.txt"))
    x, ac = torch.from_numpy(g0["sac_states"]).cuda(), torch.from_numpy(g0["sac_actions"]).cuda()
    _, _, _, mu, sigma = agent.fused.forward(x, uniform=torch.full((8, A), 0.5, device="cuda"), want_dist=True, actor_only=True)
    np.testing.assert_allclose(mu.cpu().numpy(), gold["after_mu"], atol=5e-4)
    np.testing.assert_allclose(sigma.cpu().numpy(), gold["after_sigma"], atol=5e-4, rtol=5e-4)
    agent.sync_modules()   # the trainers' weights -> the torch modules (the twin-Q trainer is sized for batch_size rows)
    with torch.no_grad():
        for m in (agent.critic_1, agent.critic_2, agent.target_critic_1, agent.target_critic_2):
            m.eval()
        np.testing.assert_allclose(agent.critic_1(x, ac).cpu().numpy(), gold["after_q1"], atol=3e-3)
        np.testing.assert_allclose(agent.critic_2(x, ac).cpu().numpy(), gold["after_q2"], atol=3e-3)
        np.testing.assert_allclose(agent.target_critic_1(x, ac).cpu().numpy(), gold["after_tq1"], atol=1e-3)
        np.testing.assert_allclose(agent.target_critic_2(x, ac).cpu().numpy(), gold["after_tq2"], atol=1e-3)
    np.testing.assert_allclose(agent.entropy.log_alpha.detach().cpu().numpy(), gold["after_log_alpha"], atol=2e-5)
    names = [m.name for m in agent.get_metrics()]
    assert names == ["actor", "critic_1", "critic_2", "entropy", "steps", "rewards"]          # soft_actor_critic.cpp:223-226
    assert agent.get_metrics()[4].values == [float(v) for v in gold["lengths"]]
