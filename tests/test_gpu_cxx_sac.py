"""examples/sac_agent_hip.hpp — the reference's SoftActorCriticAgent as compiled, torch-free C++ over the C ABI (act with check_train
inside, done, a flat ReplayBuffer with the reference's own std::mt19937 + std::shuffle, the six meters, the factory with the reference's
parameter keys) — driven by examples/sac_loop_main.cpp over the six scripted episodes the COMPILED reference ran
(tests/golden/sac_loop_golden.txt, oracle/ref_sac_loop.cpp; evo_motion_networks/src/agents/soft_actor_critic.cpp:47-91,172-180,
src/replay_buffer.cpp:10-58,146-153):

  * against the golden: every action, the buffer after every act() and done() (including the terminal transition the next act()
    rewrites), when it trains, WHICH transitions every batch draws (the adapter's generator is the reference's, nothing is plugged in);
  * against the Python SoftActorCriticAgent (same C ABI underneath): actions, the actor, the four Q networks and log_alpha bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest

import test_sac_loop as tl

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")


def _flat(module):
    import torch
    return torch.cat([p.detach().reshape(-1) for p in module.parameters()]).cpu().numpy().astype(np.float32)


def test_cxx_sac_agent_reproduces_the_reference_episode_loop_and_the_python_agent(tmp_path):
    import torch
    from evomotion_amd.sac import SoftActorCriticAgent
    from evomotion_amd.qnet import PARAMS as QPARAMS
    from evomotion_amd.ppo import ACTOR, PARAMS
    from test_sac_host import load_pattern
    if not os.path.exists(os.path.join(BUILD, "sac_loop_main")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    gold = tl.load_sac_loop_golden()
    S, A, H, batch_size, epoch, replay, train_every = gold["config"]
    lengths = gold["lengths"]

    def make_agent():
        agent = SoftActorCriticAgent(1234, [S], [A], actor_hidden_size=H, critic_hidden_size=H, batch_size=batch_size, epoch=epoch,
                                     learning_rate=1e-3, gamma=0.99, tau=0.005, replay_buffer_size=replay, train_every=train_every, device=0)
        load_pattern(agent.actor, tl.ao.ACTOR_SHAPES, 100)
        for q, base in zip((agent.critic_1, agent.critic_2, agent.target_critic_1, agent.target_critic_2), (300, 400, 500, 600)):
            load_pattern(q, tl.ao.Q_SHAPES, base)
        agent._push_actor()
        agent._push_critics()
        return agent
    agent = make_agent()
    pa = _flat(agent.actor)
    pq = [_flat(q) for q in (agent.critic_1, agent.critic_2, agent.target_critic_1, agent.target_critic_2)]

    # ---- the script: what oracle/ref_sac_loop.cpp fed the compiled reference ----
    script, dump = str(tmp_path / "script.bin"), str(tmp_path / "out.bin")
    with open(script, "wb") as f:
        np.array([S, A, H, batch_size, epoch, replay, train_every, len(lengths), 1234], np.int32).tofile(f)
        np.array(lengths, np.int32).tofile(f)
        np.array([pa.size, pq[0].size], np.int64).tofile(f)
        pa.tofile(f)
        for q in pq:
            q.tofile(f)
        k_act = n_train = 0
        for k, L in enumerate(lengths):
            for t in range(L):
                gtrain = gold["act"][k_act][4]
                tl.state_of(S, k, t).astype(np.float32).tofile(f)
                np.array([tl.reward_of(k, t)], np.float32).tofile(f)
                gold["uniform"][k_act].astype(np.float32).tofile(f)
                np.array([1 if gtrain else 0], np.int32).tofile(f)
                if gtrain:
                    for e in range(epoch):
                        gold["train_u_next"][n_train + e].astype(np.float32).tofile(f)
                        gold["train_u_curr"][n_train + e].astype(np.float32).tofile(f)
                    n_train += epoch
                k_act += 1
            tl.state_of(S, k, L).astype(np.float32).tofile(f)
            np.array([tl.reward_of(k, L)], np.float32).tofile(f)
    p = subprocess.run([os.path.join(BUILD, "sac_loop_main"), "--input", script, "--dump", dump], capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    raw = np.fromfile(dump, np.float32)
    n_act = sum(lengths)
    actions, params = raw[:n_act * A].reshape(n_act, A), raw[n_act * A:]
    assert params.size == pa.size + 4 * pq[0].size + 1 == line["count_parameters"]

    # ---- against the compiled reference's run ----
    assert [[a[2], a[3], epoch if a[4] else 0] for a in gold["act"]] == line["act"]        # global_curr_step, buffer size, train() calls
    assert line["sample"] == gold["sample"]                                                   # the reference's own std::shuffle draws
    assert line["trains"] == gold["trains"] == 10 and line["global_curr_step"] == 19
    assert np.abs(actions - gold["actions"]).max() < 3e-4
    k_act = 0
    for k, L in enumerate(lengths):
        for t in range(L):
            got, want = line["buffer_act"][k_act], gold["buffer"][("act", k, t)]
            assert len(got) == len(want) and all(g[0] == w[0] and g[2] == w[2] and g[3] == w[3] and abs(g[1] - w[1]) < 1e-6 for g, w in zip(got, want)), ("act", k, t)
            k_act += 1
        got, want = line["buffer_done"][k], gold["buffer"][("done", k, L)]
        assert len(got) == len(want) and all(g[0] == w[0] and g[2] == w[2] and g[3] == w[3] and abs(g[1] - w[1]) < 1e-6 for g, w in zip(got, want)), ("done", k)
    np.testing.assert_allclose(params[-1], gold["after_log_alpha"].ravel()[0], atol=2e-5)
    assert line["metric_names"] == ["actor", "critic_1", "critic_2", "entropy", "steps", "rewards"]      # soft_actor_critic.cpp:223-226
    assert abs(line["steps_meter"] - np.mean(lengths)) < 1e-5 and line["loss_meter_adds"] == gold["trains"]
    assert line["missing_key"] == "tau"                                                                    # agent_factory.cpp:27

    # ---- against the Python agent (same kernels through ctypes), with the C++ buffer's draws: bit for bit ----
    draws = []

    def cxx_shuffle(index):
        order = tl.shuffled_for(line["sample"][len(draws)], len(index))
        draws.append(order)
        return order
    agent.replay_buffer.shuffle = cxx_shuffle
    k_act = n_train = 0
    py_actions = []
    for k, L in enumerate(lengths):
        for t in range(L):
            gtrain = gold["act"][k_act][4]
            us = [(torch.from_numpy(gold["train_u_next"][n_train + e]), torch.from_numpy(gold["train_u_curr"][n_train + e])) for e in range(epoch)] if gtrain else None
            before = agent.curr_train_step
            a = agent.act(torch.from_numpy(tl.state_of(S, k, t)), tl.reward_of(k, t), uniform=torch.from_numpy(gold["uniform"][k_act]), train_uniforms=us)
            n_train += agent.curr_train_step - before
            py_actions.append(a.cpu().numpy())
            k_act += 1
        agent.done(torch.from_numpy(tl.state_of(S, k, L)), tl.reward_of(k, L))
    want = torch.cat([agent._actor_tr.vector(PARAMS, ACTOR)] + [agent.twinq.vector(QPARAMS, i) for i in range(4)] + [agent.entropy.log_alpha.detach().reshape(1)]).cpu().numpy()
    assert np.array_equal(np.stack(py_actions), actions), float(np.abs(np.stack(py_actions) - actions).max())
    assert np.array_equal(params, want), float(np.abs(params - want).max())
    assert np.abs(want[:pa.size] - pa).max() > 1e-5   # it did train
