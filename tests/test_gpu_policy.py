"""Fused MFMA actor-critic forward (evm_policy_forward) against the reference's golden vectors and the oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import agent_oracle as ao  # noqa: E402
import golden_io  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return golden_io.load()


@pytest.fixture(scope="module")
def fused():
    import torch
    from evomotion_amd import FusedActorCritic
    assert torch.cuda.is_available()
    f = FusedActorCritic(371, 12, 256, 0)
    pa, pc = ao.pattern_params(ao.ACTOR_SHAPES, 100), ao.pattern_params(ao.CRITIC_SHAPES, 200)
    f.set_weights(np.concatenate([pa[n].ravel() for n, _ in ao.ACTOR_SHAPES]),
                  np.concatenate([pc[n].ravel() for n, _ in ao.CRITIC_SHAPES]))
    return f, pa, pc


def test_forward_matches_reference_golden(gold, fused):
    import torch
    f, pa, pc = fused
    x = torch.from_numpy(gold["X"]).cuda()
    u = torch.full((8, 12), 0.5, device="cuda")
    action, logp, value, mu, sigma = f.forward(x, uniform=u, want_dist=True)
    # fp32 MFMA is a k-ordered fma chain; the reference's CPU GEMM sums in another order: 2e-5 absolute
    np.testing.assert_allclose(mu.cpu().numpy(), gold["mu"], atol=2e-5)
    np.testing.assert_allclose(sigma.cpu().numpy(), gold["sigma"], atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(value.cpu().numpy()[:, None], gold["value"], atol=5e-5)
    # single state (Agent::act path)
    a1, l1, v1, m1, s1 = f.forward(x[:1].contiguous(), uniform=u[:1].contiguous(), want_dist=True)
    np.testing.assert_allclose(m1.cpu().numpy()[0], gold["mu_1d"], atol=2e-5)


def test_sampling_and_logp_against_oracle(gold, fused):
    import torch
    f, pa, pc = fused
    rng = np.random.default_rng(0)
    n = 333  # ragged: not a multiple of the 32-row tile
    x = rng.uniform(-1, 1, (n, 371)).astype(np.float32)
    u = rng.uniform(0, 1, (n, 12)).astype(np.float32)
    action, logp, value, mu, sigma = f.forward(torch.from_numpy(x).cuda(), uniform=torch.from_numpy(u).cuda(), want_dist=True)
    mu_o, sg_o = ao.actor_forward(x, pa)
    v_o = ao.critic_forward(x, pc)
    np.testing.assert_allclose(mu.cpu().numpy(), mu_o, atol=3e-5)
    np.testing.assert_allclose(sigma.cpu().numpy(), sg_o, atol=3e-5, rtol=3e-5)
    np.testing.assert_allclose(value.cpu().numpy(), v_o[:, 0], atol=1e-4)
    # sample / log-pdf evaluated at the kernel's own (mu, sigma): isolates the epilogue from the GEMM rounding
    m, s = mu.cpu().numpy(), sigma.cpu().numpy()
    a_o = ao.tn_sample(m, s, u)
    a = action.cpu().numpy()
    assert (np.abs(a) <= 1).all() and np.isfinite(a).all()
    np.testing.assert_allclose(a, a_o, atol=5e-4)
    lp_o = ao.tn_log_pdf(a, m, s)
    np.testing.assert_allclose(logp.cpu().numpy(), lp_o, atol=2e-4, rtol=2e-4)


def test_unaligned_observation_buffer_takes_the_scalar_staging_path(fused):
    """A caller buffer that is only 4-byte aligned (the C ABI promises nothing more) must give the same outputs as the
    16-byte aligned float4 staging path."""
    import torch
    f, pa, pc = fused
    g = torch.Generator(device="cpu").manual_seed(5)
    n = 96
    x = (torch.rand(n, 371, generator=g) * 2 - 1)
    u = torch.rand(n, 12, generator=g).cuda()
    aligned = x.cuda()
    buf = torch.zeros(n * 371 + 1, device="cuda")
    buf[1:] = aligned.reshape(-1)
    shifted = buf[1:].view(n, 371)
    assert shifted.data_ptr() % 16 == 4 and shifted.is_contiguous()
    ref = f.forward(aligned, uniform=u, want_dist=True)
    got = f.forward(shifted, uniform=u, want_dist=True)
    for r, o in zip(ref, got):
        assert torch.equal(r, o)


def test_matches_torch_modules_at_full_batch(fused):
    import torch
    from evomotion_amd import ActorModule, CriticModule, FusedActorCritic
    torch.manual_seed(1234)
    actor, critic = ActorModule([371], [12], 256).cuda().eval(), CriticModule([371], 256).cuda().eval()
    f = FusedActorCritic(371, 12, 256, 0)
    f.load_modules(actor, critic)
    x = torch.randn(4096, 371, device="cuda")
    action, logp, value, mu, sigma = f.forward(x, want_dist=True)
    with torch.no_grad():
        m_t, s_t = actor(x)
        v_t = critic(x)
    assert (mu - m_t).abs().max() < 5e-5 and (sigma - s_t).abs().max() < 5e-5
    assert (value - v_t[:, 0]).abs().max() < 1e-4
    assert action.abs().max() <= 1 and torch.isfinite(logp).all()
    # built-in counter-based generator: a new draw every call, same call sequence -> same stream
    a2, _, _ = f.forward(x)
    assert not torch.equal(a2, action)
    g = FusedActorCritic(371, 12, 256, 0)
    g.load_modules(actor, critic)
    b1, _, _ = g.forward(x)
    assert torch.equal(b1, action)
    # empirical mean of samples tracks mu for small sigma dims
    assert (action - mu).abs().mean() < (sigma.mean() * 2)


def test_weights_from_th_checkpoints(fused, tmp_path):
    """evm_policy_forward with weights read from `.th` files equals the same weights set directly; agent save/load."""
    import torch
    from evomotion_amd import ActorModule, CriticModule, FusedActorCritic, VecPpoGaeAgent, save_th
    f, pa, pc = fused
    a, c = ActorModule([371], [12], 256), CriticModule([371], 256)
    with torch.no_grad():
        for n, p in a.named_parameters():
            p.copy_(torch.from_numpy(pa[n]))
        for n, p in c.named_parameters():
            p.copy_(torch.from_numpy(pc[n]))
    save_th(a, str(tmp_path / "actor.th"))
    save_th(c, str(tmp_path / "critic.th"))
    f2 = FusedActorCritic(371, 12, 256, 0)
    f2.load_th(str(tmp_path / "actor.th"), str(tmp_path / "critic.th"))
    g = torch.Generator(device="cpu").manual_seed(9)
    x = (torch.rand(100, 371, generator=g) * 2 - 1).cuda()
    u = torch.rand(100, 12, generator=g).cuda()
    for r, o in zip(f.forward(x, uniform=u, want_dist=True), f2.forward(x, uniform=u, want_dist=True)):
        assert torch.equal(r, o)
    ag = VecPpoGaeAgent(7, [371], [12], horizon=4, epoch=1)
    ag.save(str(tmp_path))
    ag2 = VecPpoGaeAgent(8, [371], [12], horizon=4, epoch=1)
    ag2.load(str(tmp_path))
    for r, o in zip(ag.fused.forward(x, uniform=u), ag2.fused.forward(x, uniform=u)):
        assert torch.equal(r, o)


def test_actor_only_forward_is_the_same_actor(fused):
    """d_value == NULL (SoftActorCriticAgent::act): the critic workgroups are not launched, the actor outputs are unchanged."""
    import torch
    f, pa, pc = fused
    g = torch.Generator(device="cpu").manual_seed(21)
    x = (torch.rand(333, 371, generator=g) * 2 - 1).cuda()
    u = torch.rand(333, 12, generator=g).cuda()
    a0, l0, v0, m0, s0 = f.forward(x, uniform=u, want_dist=True)
    a1, l1, v1, m1, s1 = f.forward(x, uniform=u, want_dist=True, actor_only=True)
    assert v1 is None
    for r, o in ((a0, a1), (l0, l1), (m0, m1), (s0, s1)):
        assert torch.equal(r, o)


def test_16_row_tiles_equal_32_row_tiles_to_rounding(fused, gold):
    """evm_policy_set_tile_rows: the 16-row form (v_mfma_f32_16x16x4_f32; chosen by itself when the 32-row grid would leave
    CUs idle, e.g. SAC's actor-only act at 4096 rows) against the 32-row form on full, ragged and single-row batches, both
    networks and actor only; same draws (the built-in generator is keyed by row, not by tile) -> same actions to rounding.
    Both forms are also held to the reference's golden vectors."""
    import torch
    f, pa, pc = fused
    g = torch.Generator(device="cpu").manual_seed(77)
    try:
        for n in (1, 15, 16, 17, 333, 4096):
            x = (torch.rand(n, 371, generator=g) * 2 - 1).cuda()
            u = torch.rand(n, 12, generator=g).cuda()
            for actor_only in (False, True):
                f.set_tile_rows(32)
                r32 = f.forward(x, uniform=u, want_dist=True, actor_only=actor_only)
                f.set_tile_rows(16)
                r16 = f.forward(x, uniform=u, want_dist=True, actor_only=actor_only)
                names = ("action", "logp", "value", "mu", "sigma")
                tol = {"action": 2e-4, "logp": 2e-4, "value": 2e-5, "mu": 5e-6, "sigma": 5e-6}
                for name, a, b in zip(names, r32, r16):
                    if a is None:
                        assert b is None
                        continue
                    assert torch.isfinite(b).all()
                    d = float((a - b).abs().max())
                    assert d <= tol[name], (n, actor_only, name, d)
        x = torch.from_numpy(gold["X"]).cuda()
        u = torch.full((8, 12), 0.5, device="cuda")
        for rows in (16, 32):
            f.set_tile_rows(rows)
            action, logp, value, mu, sigma = f.forward(x, uniform=u, want_dist=True)
            np.testing.assert_allclose(mu.cpu().numpy(), gold["mu"], atol=2e-5)
            np.testing.assert_allclose(sigma.cpu().numpy(), gold["sigma"], atol=2e-5, rtol=2e-5)
            np.testing.assert_allclose(value.cpu().numpy()[:, None], gold["value"], atol=5e-5)
    finally:
        f.set_tile_rows(0)


def test_bf16_split_layers_equal_fp32_mfma_layers_to_rounding(fused, gold, monkeypatch):
    """The 32-row form runs its hidden layers as six bf16 MFMA products per fp32 product (exact three-way split of both operands);
    EVM_POLICY_SPLIT=0 at creation keeps the fp32 MFMA.  Same weights (host packer and device repack), same inputs: outputs agree to
    fp32 rounding, and the split form is held to the reference's golden vectors like the other."""
    import torch
    from evomotion_amd import ActorModule, CriticModule, FusedActorCritic
    f, pa, pc = fused
    monkeypatch.setenv("EVM_POLICY_SPLIT", "0")
    f0 = FusedActorCritic(371, 12, 256, 0)
    monkeypatch.delenv("EVM_POLICY_SPLIT")
    f1 = FusedActorCritic(371, 12, 256, 0)
    flat = lambda prm, shapes: np.concatenate([prm[n].ravel() for n, _ in shapes])
    for g_ in (f0, f1):
        g_.set_weights(flat(pa, ao.ACTOR_SHAPES), flat(pc, ao.CRITIC_SHAPES))
        g_.set_tile_rows(32)
    gen = torch.Generator(device="cpu").manual_seed(123)
    for n in (1, 33, 4096):
        x = ((torch.rand(n, 371, generator=gen) * 2 - 1) * 3).cuda()
        u = torch.rand(n, 12, generator=gen).cuda()
        r0 = f0.forward(x, uniform=u, want_dist=True)
        r1 = f1.forward(x, uniform=u, want_dist=True)
        for name, a, b, tol in zip(("action", "logp", "value", "mu", "sigma"), r0, r1, (2e-4, 2e-4, 2e-5, 5e-6, 5e-6)):
            assert float((a - b).abs().max()) <= tol, (n, name, float((a - b).abs().max()))
    x = torch.from_numpy(gold["X"]).cuda()
    u = torch.full((8, 12), 0.5, device="cuda")
    _, _, value, mu, sigma = f1.forward(x, uniform=u, want_dist=True)
    np.testing.assert_allclose(mu.cpu().numpy(), gold["mu"], atol=2e-5)
    np.testing.assert_allclose(sigma.cpu().numpy(), gold["sigma"], atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(value.cpu().numpy()[:, None], gold["value"], atol=5e-5)
    # the device-side repack writes the same planes as the host packer
    a, c = ActorModule([371], [12], 256).cuda(), CriticModule([371], 256).cuda()
    with torch.no_grad():
        for n_, p_ in a.named_parameters():
            p_.copy_(torch.from_numpy(pa[n_]))
        for n_, p_ in c.named_parameters():
            p_.copy_(torch.from_numpy(pc[n_]))
    f2 = FusedActorCritic(371, 12, 256, 0)
    f2.load_modules(a, c)
    f2.set_tile_rows(32)
    x = ((torch.rand(64, 371, generator=gen) * 2 - 1)).cuda()
    u = torch.rand(64, 12, generator=gen).cuda()
    for r, o in zip(f1.forward(x, uniform=u, want_dist=True), f2.forward(x, uniform=u, want_dist=True)):
        assert torch.equal(r, o)


def test_device_side_weight_repack_equals_host_packing(fused):
    import torch
    from evomotion_amd import ActorModule, CriticModule, FusedActorCritic
    f, pa, pc = fused  # weights went through the host packer (evm_policy_set_weights)
    a, c = ActorModule([371], [12], 256).cuda(), CriticModule([371], 256).cuda()
    with torch.no_grad():
        for n, p in a.named_parameters():
            p.copy_(torch.from_numpy(pa[n]))
        for n, p in c.named_parameters():
            p.copy_(torch.from_numpy(pc[n]))
    f2 = FusedActorCritic(371, 12, 256, 0)
    f2.load_modules(a, c)  # evm_policy_set_weights_device
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.rand(200, 371, generator=g) * 2 - 1).cuda()
    u = torch.rand(200, 12, generator=g).cuda()
    for r, o in zip(f.forward(x, uniform=u, want_dist=True), f2.forward(x, uniform=u, want_dist=True)):
        assert torch.equal(r, o)


def test_rollout_and_update_smoke():
    import torch
    from evomotion_amd import VecPpoGaeAgent, VecRobotWalk
    env = VecRobotWalk(256, seed=1)
    env.reset()
    agent = VecPpoGaeAgent(1234, [371], [12], horizon=8, epoch=2)
    assert agent.count_parameters() == 330521
    w0 = agent.actor.head[0].weight.detach().clone()
    for _ in range(2):
        buf = agent.rollout(env)
        assert torch.isfinite(buf["states"]).all() and set(buf["valid"].unique().tolist()) <= {0.0, 1.0, 2.0}
        al, cl = agent.update()
        assert np.isfinite(al) and np.isfinite(cl)
    agent.sync_modules()  # the HIP trainer owns the weights; the torch modules are refreshed on demand
    assert not torch.equal(w0, agent.actor.head[0].weight)


def _flat(m):
    import torch
    return torch.cat([p.detach().reshape(-1) for p in m.parameters()])


def test_sac_hip_update_matches_torch_update():
    """two SoftActorCriticAgent::train calls on the device against the all-autograd restatement (tests/torch_ref.py::sac_train,
    pinned on the CPU to the reference's golden train() call) from the same weights, batch and uniform draws"""
    import copy
    import torch
    import torch_ref
    from evomotion_amd import VecSacAgent
    from evomotion_amd.qnet import PARAMS
    kw = dict(batch_size=512, epoch=1, replay_buffer_size=4, train_every=2, n_envs=64, use_graph=False)
    hip = VecSacAgent(21, [371], [12], **kw)

    class Ref:  # torch modules + torch.optim.Adam, the reference's own structure
        pass
    ref = Ref()
    for name in ("actor", "critic_1", "critic_2", "target_critic_1", "target_critic_2", "entropy"):
        setattr(ref, name, copy.deepcopy(getattr(hip, name)))
    opts = [torch.optim.Adam(m.parameters(), lr=1e-3) for m in (ref.actor, ref.critic_1, ref.critic_2, ref.entropy)]
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    s = (torch.rand(512, 371, device="cuda", generator=g) * 2 - 1) * 1.5
    a = torch.rand(512, 12, device="cuda", generator=g) * 2 - 1
    r = torch.rand(512, device="cuda", generator=g) * 2 - 1
    d = (torch.rand(512, device="cuda", generator=g) < 0.1).float()
    n = (torch.rand(512, 371, device="cuda", generator=g) * 2 - 1) * 1.5
    for dst, src in zip(hip._batch, (s, a, r, d, n)):
        dst.copy_(src)
    for it in range(2):
        u_next = torch.rand(512, 12, device="cuda", generator=g)
        u_curr = torch.rand(512, 12, device="cuda", generator=g)
        lh = hip._train_once_hip(u_next=u_next, u_curr=u_curr)
        lr_ = torch_ref.sac_train(ref.actor, ref.critic_1, ref.critic_2, ref.target_critic_1, ref.target_critic_2, ref.entropy,
                                  opts[0], opts[1], opts[2], opts[3], s, a, r.unsqueeze(-1), d.unsqueeze(-1), n, hip.gamma, hip.tau,
                                  hip.target_entropy, u_next=u_next, u_curr=u_curr)
        for k in ("critic_1", "critic_2", "actor", "entropy"):
            assert abs(float(lh[k]) - float(lr_[k])) < 2e-4 * max(1.0, abs(float(lr_[k]))), (it, k, float(lh[k]), float(lr_[k]))
    hip.sync_modules()  # the trainers own the weights; the torch modules are refreshed on demand
    pairs = [(hip.twinq.vector(PARAMS, 0), _flat(ref.critic_1)), (hip.twinq.vector(PARAMS, 1), _flat(ref.critic_2)),
             (hip.twinq.vector(PARAMS, 2), _flat(ref.target_critic_1)), (hip.twinq.vector(PARAMS, 3), _flat(ref.target_critic_2)),
             (_flat(hip.actor), _flat(ref.actor)), (_flat(hip.entropy), _flat(ref.entropy))]
    for i, (x, y) in enumerate(pairs):
        dd = (x - y).abs()
        # two Adam steps of 1e-3: rounding-level gradient differences move a parameter only where |g| ~ 1e-8
        assert float(dd.max()) <= 2.1e-3 and float((dd > 5e-5).float().mean()) < 1e-3, (i, float(dd.max()), float((dd > 5e-5).float().mean()))
    assert torch.equal(_flat(hip.critic_1), hip.twinq.vector(PARAMS, 0))
    assert torch.equal(_flat(hip.target_critic_2), hip.twinq.vector(PARAMS, 3))
    # the rollout kernel already has the new actor
    x = s[:64]
    u = torch.full((64, 12), 0.5, device="cuda")
    _, _, _, mu_k, _ = hip.fused.forward(x, uniform=u, want_dist=True, actor_only=True)
    hip.actor.eval()
    with torch.no_grad():
        mu_t, _ = hip.actor(x)
    np.testing.assert_allclose(mu_k.cpu().numpy(), mu_t.cpu().numpy(), atol=2e-5)


def test_sac_hip_agent_graph_replay():
    """act + ring + the hybrid update captured into a HIP graph: runs, moves the weights, keeps them finite"""
    import torch
    from evomotion_amd import VecRobotWalk, VecSacAgent
    from evomotion_amd.qnet import PARAMS
    n = 128
    env = VecRobotWalk(n, seed=5)
    env.reset()
    ag = VecSacAgent(11, [371], [12], batch_size=64, epoch=1, replay_buffer_size=16, train_every=2, n_envs=n, use_graph=True)
    w0 = ag.twinq.vector(PARAMS, 0).clone()
    t0 = ag.twinq.vector(PARAMS, 2).clone()
    for _ in range(12):
        st = ag.step(env)
    assert ag._graph is not None and ag.train_steps >= 5
    w1, t1 = ag.twinq.vector(PARAMS, 0), ag.twinq.vector(PARAMS, 2)
    assert torch.isfinite(w1).all() and torch.isfinite(t1).all() and not torch.equal(w0, w1) and not torch.equal(t0, t1)
    assert ag.twinq.adam_step(0) == ag.train_steps  # the device step counter advanced once per (replayed) update
    assert all(torch.isfinite(p).all() for p in ag.actor.parameters())


def test_sac_hip_agent_save_load_round_trip(tmp_path):
    """SoftActorCriticAgent::save / load (soft_actor_critic.cpp:181-223) with the weights owned by the device trainers"""
    import torch
    from evomotion_amd import VecRobotWalk, VecSacAgent
    from evomotion_amd.ppo import PARAMS as PP, ACTOR
    from evomotion_amd.qnet import PARAMS
    n = 64
    env = VecRobotWalk(n, seed=9)
    env.reset()
    ag = VecSacAgent(5, [371], [12], batch_size=64, epoch=1, replay_buffer_size=8, train_every=2, n_envs=n, use_graph=False)
    for _ in range(6):
        ag.step(env)
    assert ag.train_steps >= 2
    ag.save(str(tmp_path))
    ag2 = VecSacAgent(6, [371], [12], batch_size=64, epoch=1, replay_buffer_size=8, train_every=2, n_envs=n, use_graph=False)
    ag2.load(str(tmp_path))
    for i in range(4):
        assert torch.equal(ag.twinq.vector(PARAMS, i), ag2.twinq.vector(PARAMS, i))
    assert torch.equal(ag._actor_tr.vector(PP, ACTOR), ag2._actor_tr.vector(PP, ACTOR))
    assert torch.equal(ag.entropy.log_alpha.detach(), ag2.entropy.log_alpha.detach())
    x = torch.rand(32, 371, device="cuda") * 2 - 1
    u = torch.rand(32, 12, device="cuda")
    for r, o in zip(ag.fused.forward(x, uniform=u, actor_only=True)[:2], ag2.fused.forward(x, uniform=u, actor_only=True)[:2]):
        assert torch.equal(r, o)


def test_random_agent_on_device():
    """RandomAgent (debug_agents.cpp:28-30) at the headline batch: shape, range, determinism per seed, different seeds differ,
    and the env accepts its actions as they are (device-resident, contiguous fp32)."""
    import torch
    from evomotion_amd import RandomAgent, VecRobotWalk
    st = torch.zeros(4096, 371, device="cuda")
    a = RandomAgent([12], "cuda", seed=1234).act(st)
    b = RandomAgent([12], "cuda", seed=1234).act(st)
    c = RandomAgent([12], "cuda", seed=1235).act(st)
    assert a.shape == (4096, 12) and a.dtype == torch.float32 and a.is_cuda and a.is_contiguous()
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert float(a.min()) >= -1.0 and float(a.max()) < 1.0
    assert abs(float(a.mean())) < 0.02 and abs(float(a.var()) - 1.0 / 3.0) < 0.02  # U[-1, 1)
    ra = RandomAgent([12], "cuda", seed=5)
    a1, a2 = ra.act(st), ra.act(st)
    assert not torch.equal(a1, a2)  # the stream advances
    env = VecRobotWalk(64, seed=3, device=0)
    s0 = env.reset()
    s1 = env.do_step(RandomAgent([12], "cuda", seed=9).act(s0.state))
    assert torch.isfinite(s1.state).all()
    env.close()
