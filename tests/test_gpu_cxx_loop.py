"""examples/train_loop_main.cpp — the reference's train() loop (src/train.cpp:41-83: reset, `while (!step.done) step =
env->do_step(agent->act(step.state, step.reward))`, agent->done, env->reset, get_metrics, save) on the two COMPILED adapters
(robot_walk_hip.hpp: Environment / factory; ppo_gae_agent_hip.hpp: Agent / factory), torch-free — against the same loop driven
through the Python adapters (evomotion_amd.VecRobotWalk with one env, evomotion_amd.PpoGaeAgent): same episode lengths, same
train() calls, and — with the C++ buffer's trajectory draws plugged into the Python buffer — the same weights bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest

from test_gpu_cxx_host import make_params

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")
SKEL = os.path.join(ROOT, "evomotion_amd", "data", "robot_walk_spider.skel")


def test_cxx_train_loop_equals_the_python_driven_loop(tmp_path):
    import torch
    from evomotion_amd import PpoGaeAgent, VecRobotWalk
    from evomotion_amd.ppo import ACTOR, CRITIC, PARAMS
    if not os.path.exists(os.path.join(BUILD, "train_loop_main")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    S, A, episodes, seed, batch_size, train_every, epoch = 371, 12, 10, 1234, 3, 3, 2
    pa, pc = make_params(S, A, 256, True, 1000), make_params(S, A, 256, False, 500000)
    weights, dump, out = str(tmp_path / "w.bin"), str(tmp_path / "final.bin"), str(tmp_path / "saves")
    os.makedirs(out)
    np.concatenate([pa, pc]).tofile(weights)
    p = subprocess.run([os.path.join(BUILD, "train_loop_main"), "--skeleton", SKEL, "--episodes", str(episodes), "--seed", str(seed), "--batch-size",
                        str(batch_size), "--train-every", str(train_every), "--epoch", str(epoch), "--weights", weights, "--dump", dump, "--out", out],
                       capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    got = np.fromfile(dump, np.float32)
    assert line["state_space"] == S and line["action_space"] == A and line["parameters_count"] == 330521
    assert len(line["lengths"]) == episodes and line["steps"] == sum(line["lengths"]) and line["trains"] >= 2
    assert line["metrics"].startswith("Save -1, actor_loss = ") and ", steps = " in line["metrics"]      # the progress bar's text (train.cpp:67-76)
    # agent->save(out / "save_0") (train.cpp:82-85): the reference's four files (ppo_gae.cpp:192-197), written without LibTorch by
    # examples/th_archive.hpp — torch reads them, names and bits as the final weights the program dumped
    save0 = os.path.join(out, "save_0")
    assert sorted(os.listdir(save0)) == ["actor.th", "actor_optimizer.th", "critic.th", "critic_optimizer.th"]
    from evomotion_amd.checkpoint import load_adam_th, load_th
    sa, sc = load_th(os.path.join(save0, "actor.th")), load_th(os.path.join(save0, "critic.th"))
    assert list(sa.keys())[0] == "head.0.weight" and list(sa.keys())[-1] == "sigma.0.bias" and list(sc.keys())[-1] == "critic.6.bias"
    saved = np.concatenate([t.numpy().ravel() for t in list(sa.values()) + list(sc.values())])
    assert np.array_equal(saved, got)
    st_a, opt_a = load_adam_th(os.path.join(save0, "actor_optimizer.th"))
    assert opt_a["lr"] == pytest.approx(1e-3) and len(st_a) == 12 and all(s_ is not None and s_[0] == line["trains"] * epoch for s_ in st_a)

    # ---- the same loop through the Python adapters ----
    env = VecRobotWalk(1, seed=seed, device=0)
    agent = PpoGaeAgent(seed, [S], [A], hidden_size=256, gamma=0.99, lam=0.95, epsilon=0.2, entropy_factor=0.01, critic_loss_factor=0.5, epoch=epoch,
                        batch_size=batch_size, train_every=train_every, replay_buffer_size=64, learning_rate=1e-3, clip_grad_norm=0.5, device=0)
    agent.fused.set_weights(pa, pc)
    flat = torch.from_numpy(np.concatenate([pa, pc])).cuda()
    for mod, vec in ((agent.actor, flat[:pa.size]), (agent.critic, flat[pa.size:])):    # the trainer starts from the modules
        o = 0
        with torch.no_grad():
            for prm in mod.parameters():
                prm.copy_(vec[o:o + prm.numel()].view_as(prm))
                o += prm.numel()
    draws = []

    def cxx_shuffle(index):
        filtered = [i for i, t in enumerate(agent.replay_buffer.memory) if len(t) > 1]
        drawn = [filtered.index(pos) for pos in line["sample"][len(draws)]]
        draws.append(drawn)
        return drawn + [i for i in index if i not in drawn]
    agent.replay_buffer.shuffle = cxx_shuffle
    st = env.reset()
    lengths = []
    for e in range(episodes):
        n = 0
        while not bool(st.done[0]):
            st = env.do_step(agent.act(st.state[0], float(st.reward[0]))[None])
            n += 1
        agent.done(st.state[0], float(st.reward[0]))
        st = env.reset()
        lengths.append(n)
    assert lengths == line["lengths"]
    assert agent.curr_train_step == line["trains"] == len(draws)
    want = torch.cat([agent._trainer.vector(PARAMS, ACTOR), agent._trainer.vector(PARAMS, CRITIC)]).cpu().numpy()
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    assert np.abs(want - np.concatenate([pa, pc])).max() > 1e-5
    assert env.errors() == (0, 0) or list(env.errors()) == [0, 0]
