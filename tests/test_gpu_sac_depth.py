"""BASELINE configs[4] at its stated depth: SAC with replay_buffer_size = 1024 slots x 4096 envs (4.2 M transitions, 6.5 GB
ring).  No oracle of that size: size-independent properties (every drawn row is a stored row, its next state is the state one
slot later for the same env, draws are distinct, the ring wrapped), then one SoftActorCriticAgent::train call on the device
with a 4096-row batch drawn from it.  Plus the agent-level rules around the memory: no update before has_enough(batch), one
effective train() per update on the captured-graph path, optimiser state through save() / load() in the reference's archive format."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S, A = 371, 12
MASK = (1 << 20) - 1


def synth_state(base, t):
    """state of push t: exact small integers as fp32, cheap to recompute for any (t, env) instead of keeping 1027 x 6 MB"""
    return ((base + 7 * t) & MASK).float()


def test_config5_ring_depth_and_one_update():
    import torch
    from evomotion_amd import VecSacAgent
    C, N, B = 1024, 4096, 4096
    agent = VecSacAgent(1234, [S], [A], batch_size=B, replay_buffer_size=C, train_every=4, n_envs=N, device=0, use_graph=False)
    ring = agent.replay
    g = torch.Generator(device="cuda").manual_seed(0)
    base = torch.randint(0, MASK + 1, (N, S), device="cuda", generator=g, dtype=torch.int64)
    pushes = C + 2  # the ring wraps: pushes 0 and 1 are overwritten
    valid_count = torch.zeros(pushes, dtype=torch.int64)
    for t in range(pushes):
        valid = (((torch.arange(N, device="cuda") * 2654435761 + t * 40503) >> 7) % 5 != 0).to(torch.uint8)  # 80 % valid, recomputable
        valid_count[t] = int(valid.sum()) if t >= 2 else 0
        ring.push(synth_state(base, t), torch.full((N, A), float(t % 251) / 256.0, device="cuda"),
                  torch.full((N,), float(t), device="cuda"), ((torch.arange(N, device="cuda") + t) % 97 == 0).to(torch.uint8),
                  valid, synth_state(base, t + 1))
    st = ring.stats()
    assert st["pushes"] == pushes and st["live_slots"] == C
    assert st["transitions"] == int(valid_count.sum()) > 3_000_000       # > 2^31 / 1484 B rows: 64-bit offsets are exercised
    states, actions, rewards, done, nxt, idx = ring.sample(B, 12345, want_index=True)
    t_of, env = rewards.long(), idx[:, 1].long()
    assert int(t_of.min()) >= 2 and int(t_of.max()) <= pushes - 1         # nothing from the two overwritten pushes
    assert int(t_of.max()) > C // 2 and int(t_of.min()) < C // 2          # draws spread over the whole ring
    assert torch.equal(states, synth_state(base[env], t_of[:, None])) and torch.equal(nxt, synth_state(base[env], t_of[:, None] + 1))
    assert torch.equal(actions[:, 0], (t_of % 251).float() / 256.0)   # (exact in fp32)
    assert torch.equal(done, ((env + t_of) % 97 == 0).float())
    assert bool(((((env * 2654435761 + t_of * 40503) >> 7) % 5) != 0).all())   # only valid rows are stored
    assert len(set(map(tuple, idx.cpu().numpy()))) == B                    # distinct draws
    # one train() call on the device with a full-depth batch (soft_actor_critic.cpp:93-170)
    assert agent.has_enough()
    from evomotion_amd.qnet import PARAMS
    q_before = agent.twinq.vector(PARAMS, 0).clone()
    agent.update()
    torch.cuda.synchronize()
    losses = {k: float(v) for k, v in agent.last_losses.items()}
    assert all(np.isfinite(v) for v in losses.values()), losses
    assert not torch.equal(agent.twinq.vector(PARAMS, 0), q_before)
    assert agent.train_steps == 1


def test_no_update_before_the_memory_holds_a_batch():
    """soft_actor_critic.cpp:64: `global_step % train_every == train_every - 1 && replay_buffer.has_enough(batch_size)`"""
    import torch
    from evomotion_amd import VecRobotWalk, VecSacAgent
    n = 64
    env = VecRobotWalk(n, seed=5, device=0)
    env.reset()
    agent = VecSacAgent(7, [S], [A], batch_size=256, replay_buffer_size=16, train_every=1, n_envs=n, device=0, use_graph=False)
    for k in range(3):      # 3 x 64 = 192 < 256 stored transitions: no update yet
        agent.step(env)
    assert agent.train_steps == 0 and agent.replay.stats()["transitions"] <= 192
    for k in range(3):
        agent.step(env)
    assert agent.replay.stats()["transitions"] >= 256 and agent.train_steps >= 1
    # an all-invalid ring (every env settling inside reset()) holds nothing: has_enough() is False again
    empty = VecSacAgent(7, [S], [A], batch_size=8, replay_buffer_size=4, train_every=1, n_envs=n, device=0, use_graph=False)
    z = torch.zeros
    for k in range(5):
        empty.replay.push(z(n, S, device="cuda"), z(n, A, device="cuda"), z(n, device="cuda"), z(n, dtype=torch.uint8, device="cuda"),
                          z(n, dtype=torch.uint8, device="cuda"), z(n, S, device="cuda"))
        empty._pushes_since_check += 1
    assert not empty.has_enough()


def test_graph_path_first_update_is_one_train_call_and_state_round_trips(tmp_path):
    """captured-graph path: the two warm-up calls must leave no trace — after the first update the networks equal those of an
    eager agent after ONE train() on the same batch and draws; save() / load() carries weights and all four Adam states"""
    import torch
    from evomotion_amd import VecSacAgent
    from evomotion_amd.qnet import PARAMS
    from evomotion_amd.ppo import ACTOR, ACTOR_DEV_STEP, EXP_AVG, PARAMS as PP
    n, B = 64, 128
    mk = lambda graph: VecSacAgent(11, [S], [A], batch_size=B, replay_buffer_size=8, train_every=1, n_envs=n, device=0, use_graph=graph)
    eager, graph = mk(False), mk(True)
    g = torch.Generator(device="cuda").manual_seed(3)
    for t in range(6):
        args = (torch.randn(n, S, device="cuda", generator=g), torch.rand(n, A, device="cuda", generator=g) * 2 - 1,
                torch.randn(n, device="cuda", generator=g), torch.zeros(n, dtype=torch.uint8, device="cuda"), None,
                torch.randn(n, S, device="cuda", generator=g))
        eager.replay.push(*args)
        graph.replay.push(*args)
    torch.manual_seed(99); eager.update()
    torch.manual_seed(99); graph.update()   # (the warm-up calls draw uniforms too: the draws differ, the step count must not)
    torch.cuda.synchronize()
    assert eager.train_steps == graph.train_steps == 1
    assert graph.twinq.adam_step(0) == eager.twinq.adam_step(0) == 1 and graph._actor_tr.adam_step(ACTOR_DEV_STEP) == 1
    assert int(graph._ent_step.item()) == 1
    # one Adam step moves every weight by at most lr: three steps would show up as up to 3 lr
    q0 = mk(False).twinq.vector(PARAMS, 0)
    assert float((graph.twinq.vector(PARAMS, 0) - q0).abs().max()) <= 1.001e-3
    # save / load: weights and all four Adam states survive, in the reference's own archive format
    graph.save(str(tmp_path))
    for f in ("actor.th", "critic_1.th", "target_critic_2.th", "entropy.th", "actor_optimizer.th", "critic_1_optimizer.th",
              "critic_2_optimizer.th", "entropy_optimizer.th"):
        assert os.path.isfile(tmp_path / f), f
    other = mk(False)
    other.load(str(tmp_path))
    torch.cuda.synchronize()
    assert torch.equal(other.twinq.vector(PARAMS, 1), graph.twinq.vector(PARAMS, 1))
    assert torch.equal(other._actor_tr.vector(PP, ACTOR), graph._actor_tr.vector(PP, ACTOR))
    assert torch.equal(other._actor_tr.vector(EXP_AVG, ACTOR), graph._actor_tr.vector(EXP_AVG, ACTOR))
    assert other.twinq.adam_step(1) == 1 and int(other._ent_step.item()) == 1 and other._actor_tr.adam_step(ACTOR_DEV_STEP) == 1
    assert torch.equal(other._ent_state, graph._ent_state) and other.train_steps == 1
    # like the reference's load_torch, a missing archive is an error, not a silent restart of the optimiser
    os.remove(tmp_path / "critic_2_optimizer.th")
    with pytest.raises(RuntimeError, match="Could not find"):
        mk(False).load(str(tmp_path))
