"""examples/ppo_gae_agent_hip.hpp — the reference's `Agent` interface for PPO as compiled, torch-free C++ over the C ABI (act / done /
check_train over whole episodes, a TrajectoryReplayBuffer with the reference's own std::mt19937 + std::shuffle, save / load,
get_metrics, the factory with the reference's parameter keys) — driven by examples/agent_loop_main.cpp over the ten scripted
episodes the COMPILED reference ran (tests/golden/agent_loop_golden.txt, oracle/ref_loop.cpp; evo_motion_networks/src/agents/
ppo_gae.cpp:29-115, src/replay_buffer.cpp:64-146,176-189):

  * against the golden: every action, the buffer's shape after every done(), when it trains, WHICH trajectories it draws (the
    adapter's generator is the reference's, nothing is plugged in), what update_last left behind, the weights after four train() calls;
  * against the Python PpoGaeAgent (same C ABI underneath): actions and final weights bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest

import test_agent_loop as tl

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")


def _flat(module):
    import torch
    return torch.cat([p.detach().reshape(-1) for p in module.parameters()]).numpy().astype(np.float32)


def test_cxx_agent_reproduces_the_reference_episode_loop_and_the_python_agent(tmp_path):
    import torch
    from evomotion_amd.agent import ActorModule, CriticModule, PpoGaeAgent
    from evomotion_amd.ppo import ACTOR, CRITIC, PARAMS
    if not os.path.exists(os.path.join(BUILD, "agent_loop_main")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    gold = tl.load_loop_golden()
    S, A, H, epoch, batch_size, train_every, replay = gold["config"]
    lengths = gold["lengths"]
    actor, critic = ActorModule([S], [A], H), CriticModule([S], H)
    tl.load_pattern(actor, tl.ao.ACTOR_SHAPES, 100)
    tl.load_pattern(critic, tl.ao.CRITIC_SHAPES, 200)
    pa, pc = _flat(actor), _flat(critic)

    # ---- the script: what oracle/ref_loop.cpp fed the compiled reference ----
    script, dump, ckpt = str(tmp_path / "script.bin"), str(tmp_path / "out.bin"), str(tmp_path / "ckpt")
    os.makedirs(ckpt)
    with open(script, "wb") as f:
        np.array([S, A, H, epoch, batch_size, train_every, replay, len(lengths), 1234], np.int32).tofile(f)
        np.array(lengths, np.int32).tofile(f)
        np.array([pa.size, pc.size], np.int64).tofile(f)
        pa.tofile(f)
        pc.tofile(f)
        k_act = 0
        for k, L in enumerate(lengths):
            for t in range(L):
                tl.state_of(S, k, t).astype(np.float32).tofile(f)
                np.array([tl.reward_of(k, t)], np.float32).tofile(f)
                gold["uniform"][k_act].astype(np.float32).tofile(f)
                k_act += 1
            tl.state_of(S, k, L).astype(np.float32).tofile(f)
            np.array([tl.reward_of(k, L)], np.float32).tofile(f)
    p = subprocess.run([os.path.join(BUILD, "agent_loop_main"), "--input", script, "--dump", dump, "--ckpt", ckpt], capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    raw = np.fromfile(dump, np.float32)
    n_act = sum(lengths)
    actions = raw[:n_act * A].reshape(n_act, A)
    params = raw[n_act * A:n_act * A + pa.size + pc.size]
    tail = raw[n_act * A + pa.size + pc.size:]
    L2 = int(tail[0])
    last_rewards, last_done, last_values, last_next = (tail[1 + i * L2:1 + (i + 1) * L2] for i in range(4))
    last_logp = tail[1 + 4 * L2:].reshape(L2, A)

    # ---- against the compiled reference's run ----
    assert line["done"] == gold["done"]                  # k, global_curr_step, memory, filtered, trained
    assert line["buffer"] == gold["buffer"]              # trajectory lengths after every done(), the open one last
    assert line["sample"] == gold["sample"]              # memory positions in draw order: the reference's own std::shuffle draws
    assert line["trains"] == line["curr_train_step"] == gold["trains"] == 4
    assert np.abs(actions - gold["actions"]).max() < 2e-4
    np.testing.assert_allclose(last_rewards, gold["last_rewards"], atol=1e-7)
    np.testing.assert_array_equal(last_done, gold["last_done"])
    np.testing.assert_allclose(last_values, gold["last_values"], atol=5e-3)
    np.testing.assert_allclose(last_next, gold["last_next_values"], atol=5e-3)
    np.testing.assert_allclose(last_logp, gold["last_log_prob"], atol=5e-3)
    np.testing.assert_allclose(params[:H * S].reshape(H, S)[0], gold["after_actor_w0_row0"], atol=3e-5)
    assert line["metric_names"] == ["actor_loss", "critic_loss", "steps"]                        # ppo_gae.cpp:205-207
    assert abs(line["steps_meter"] - np.mean(lengths)) < 1e-5                                      # episode_steps_meter, window 64
    assert line["loss_meter_adds"] == gold["trains"] * epoch                                       # one add per epoch (:185-186)
    assert line["meter_known_answers"] == [1.5, 1.5, 2.0]                                         # test_metrics.cpp:20-25
    assert line["steps_string"] == "steps = %.6f" % np.mean(lengths)                                # metrics.cpp:52-56,70-74
    # RandomAgent / ConstantAgent through the same factory: the random agent's stream IS the reference's (golden from the compiled reference)
    g0 = tl.golden_io.load()
    assert np.array_equal(np.array(line["random_actions"], np.float32).reshape(3, 12), g0["random_agent_actions"])
    assert line["constant_action"] == [0.25, 0.25] and line["constant_missing"] == "action_value"       # agent_factory.cpp:76
    assert line["missing_key"] == "gamma" and line["unknown_name"] == "no_such_agent"            # agent_factory.cpp:27,208-209
    assert line["ckpt_equal"] is True and line["count_parameters"] == pa.size + pc.size == 330521
    assert np.isfinite(line["actor_loss"]) and np.isfinite(line["critic_loss"])

    # ---- against the Python agent (same kernels through ctypes): bit for bit ----
    agent = PpoGaeAgent(1234, [S], [A], hidden_size=H, gamma=0.99, lam=0.95, epsilon=0.2, entropy_factor=0.01, critic_loss_factor=0.5,
                        epoch=epoch, batch_size=batch_size, train_every=train_every, replay_buffer_size=replay, learning_rate=1e-3,
                        clip_grad_norm=0.5, device=0)
    tl.load_pattern(agent.actor, tl.ao.ACTOR_SHAPES, 100)
    tl.load_pattern(agent.critic, tl.ao.CRITIC_SHAPES, 200)
    agent.fused.load_modules(agent.actor, agent.critic)
    draws = []

    def cxx_shuffle(index):
        filtered = [i for i, t in enumerate(agent.replay_buffer.memory) if len(t) > 1]
        order = tl.shuffled_for(line["sample"][len(draws)], filtered)
        draws.append(order)
        return order
    agent.replay_buffer.shuffle = cxx_shuffle
    k_act = 0
    py_actions = []
    for k, L in enumerate(lengths):
        for t in range(L):
            a = agent.act(torch.from_numpy(tl.state_of(S, k, t)), tl.reward_of(k, t), uniform=torch.from_numpy(gold["uniform"][k_act]))
            py_actions.append(a.cpu().numpy())
            k_act += 1
        agent.done(torch.from_numpy(tl.state_of(S, k, L)), tl.reward_of(k, L))
    want = torch.cat([agent._trainer.vector(PARAMS, ACTOR), agent._trainer.vector(PARAMS, CRITIC)]).cpu().numpy()
    assert np.array_equal(np.stack(py_actions), actions), float(np.abs(np.stack(py_actions) - actions).max())
    assert np.array_equal(params, want), float(np.abs(params - want).max())
    assert np.abs(want - np.concatenate([pa, pc])).max() > 1e-4   # it did train
