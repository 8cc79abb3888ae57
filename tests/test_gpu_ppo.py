"""PPO / GAE update on the device (evm_ppo_*) against PyTorch autograd of the reference's formulas (tests/torch_ref.py::ppo_train,
itself pinned to the reference's golden train call in test_agent_host.py) and against that golden call directly."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import agent_oracle as ao  # noqa: E402
import golden_io  # noqa: E402
import torch_ref  # noqa: E402  (tests/torch_ref.py: autograd restatement of PpoGaeAgent::train)

pytestmark = pytest.mark.gpu

HP = dict(gamma=0.99, lam=0.95, epsilon=0.2, entropy_factor=0.01, critic_loss_factor=0.5, clip_grad_norm=0.5)


@pytest.fixture(scope="module")
def gold():
    return golden_io.load()


def _modules(seed=0, pattern=False, S=371, A=12):
    import torch
    from evomotion_amd import agent
    torch.manual_seed(seed)
    actor = agent.ActorModule([S], [A], 256).cuda()
    critic = agent.CriticModule([S], 256).cuda()
    if pattern:
        for mod, shapes, base in ((actor, ao.ACTOR_SHAPES, 100), (critic, ao.CRITIC_SHAPES, 200)):
            p = ao.pattern_params(shapes, base)
            with torch.no_grad():
                for name, t in mod.named_parameters():
                    t.copy_(torch.from_numpy(p[name]))
    else:
        # LayerNorm weights away from (1, 0) so that their gradients and their use in the backward pass are exercised
        with torch.no_grad():
            for mod in (actor, critic):
                for m in mod.modules():
                    if isinstance(m, torch.nn.LayerNorm):
                        m.weight.add_(0.3 * torch.randn_like(m.weight))
                        m.bias.add_(0.2 * torch.randn_like(m.bias))
    return actor, critic


def _rollout(T, N, seed=1, p_invalid=0.2, p_done=0.1):
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    r = lambda *s: torch.rand(*s, device="cuda", generator=g)
    states = (r(T, N, 371) * 2 - 1) * 1.5
    actions = (r(T, N, 12) * 2 - 1) * 0.98
    rewards = r(T, N) * 2 - 1
    values = r(T, N) * 2 - 1
    next_values = r(T, N) * 2 - 1
    mask = (r(T, N) > p_invalid).to(torch.uint8)
    done = (r(T, N) < p_done).to(torch.uint8)
    return states, actions, rewards, done, values, next_values, mask


def _trainer(actor, critic, max_rows, S=371, A=12):
    from evomotion_amd import FusedActorCritic
    from evomotion_amd.ppo import FusedPpoTrainer
    f = FusedActorCritic(S, A, 256, 0)
    tr = FusedPpoTrainer(f, max_rows)
    tr.set_modules(actor, critic)
    return f, tr


def test_gae_matches_torch():
    import torch
    from evomotion_amd import agent
    actor, critic = _modules()
    T, N = 13, 300
    _, _, rewards, done, values, next_values, mask = _rollout(T, N)
    f, tr = _trainer(actor, critic, T * N)
    adv, ret, n_glob = tr.gae(rewards, done, values, next_values, mask, 0.99, 0.95)
    tb = lambda x: x.transpose(0, 1).contiguous().unsqueeze(-1)
    mb = tb(mask) == 1
    dn = torch.where(mb, tb(done).float(), torch.ones((), device="cuda"))
    m2, a2, r2 = torch_ref.gae_advantages(tb(rewards), dn, tb(values), tb(next_values), 0.99, 0.95, mask=mb)
    assert float(n_glob) == float(mb.sum())
    sel = mb.squeeze(-1).transpose(0, 1)
    np.testing.assert_allclose(adv[sel].cpu().numpy(), a2.squeeze(-1).transpose(0, 1)[sel].cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(ret[sel].cpu().numpy(), r2.squeeze(-1).transpose(0, 1)[sel].cpu().numpy(), atol=2e-5)


def _torch_grads(actor, critic, states, actions, logp_old, adv, returns, mask, n_glob):
    """the losses of ppo_gae.cpp:155-179 on flat rows, gradients by autograd"""
    import torch
    from evomotion_amd import agent
    actor.zero_grad(); critic.zero_grad()
    mu, sigma = actor(states)
    lp = agent.truncated_normal_log_pdf(actions, mu, sigma)
    ent = agent.truncated_normal_entropy(mu, sigma)
    ratios = torch.exp(lp - logp_old)
    a = adv.unsqueeze(-1)
    s1 = ratios * a
    s2 = torch.clamp(ratios, 1.0 - HP["epsilon"], 1.0 + HP["epsilon"]) * a
    mb = mask.bool()
    a_loss = -torch.mean(torch.masked_select(torch.min(s1, s2) + HP["entropy_factor"] * ent, mb.unsqueeze(-1).expand_as(ent)))
    a_loss.backward()
    v = critic(states).squeeze(-1)
    c_loss = HP["critic_loss_factor"] * torch.mean(torch.masked_select(torch.pow(v - returns, 2.0), mb))
    c_loss.backward()
    flat = lambda m: torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    return flat(actor), flat(critic), float(a_loss), float(c_loss)


# (rows, state size, action size): the robot_walk sizes on full, ragged and single tiles; a generic skeleton's sizes (odd input
# width, head GEMM with 10 of 32 columns, a ragged tile; 16 actions fill the 32 head columns)
@pytest.mark.parametrize("rows,S,A", [(1000, 371, 12), (28, 371, 12), (4096, 371, 12), (77, 101, 5), (160, 64, 16)])
def test_gradients_match_autograd(rows, S, A):
    import torch
    from evomotion_amd import agent
    from evomotion_amd.ppo import GRADS, ACTOR, CRITIC
    actor, critic = _modules(seed=3, S=S, A=A)
    f, tr = _trainer(actor, critic, rows, S=S, A=A)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    states = (torch.rand(rows, S, device="cuda", generator=g) * 2 - 1) * 1.5
    with torch.no_grad():
        mu, sigma = actor(states)
        actions = agent.truncated_normal_sample(mu, sigma, u=torch.rand(rows, A, device="cuda", generator=g))
        # old log-probabilities around the current ones: ratios on both sides of the clip range
        logp_old = agent.truncated_normal_log_pdf(actions, mu, sigma) + 0.3 * (torch.rand(rows, A, device="cuda", generator=g) * 2 - 1)
    adv = torch.randn(rows, device="cuda", generator=g)
    returns = torch.randn(rows, device="cuda", generator=g)
    mask = (torch.rand(rows, device="cuda", generator=g) > 0.25).to(torch.uint8)
    n_glob = float(mask.sum())
    from evomotion_amd._lib import lib, check
    from evomotion_amd.ppo import _ptr
    check(lib.evm_ppo_grads(tr._h, rows, _ptr(states), _ptr(actions), _ptr(logp_old), _ptr(adv), _ptr(returns), _ptr(mask), n_glob,
                            HP["epsilon"], HP["entropy_factor"], HP["critic_loss_factor"], 0, tr._stream()))
    ga, gc = tr.vector(GRADS, ACTOR), tr.vector(GRADS, CRITIC)
    la, lc = tr.losses()
    ra, rc, rla, rlc = _torch_grads(actor, critic, states, actions, logp_old, adv, returns, mask, n_glob)
    assert abs(la - rla) < 2e-5 * max(1.0, abs(rla)) and abs(lc - rlc) < 2e-5 * max(1.0, abs(rlc))
    # per parameter tensor, relative to the tensor's largest gradient
    for flat, ref, mod in ((ga, ra, actor), (gc, rc, critic)):
        o = 0
        for name, p in mod.named_parameters():
            n = p.numel()
            a, b = flat[o:o + n], ref[o:o + n]
            scale = float(b.abs().max())
            err = float((a - b).abs().max())
            assert err <= 2e-4 * scale + 1e-8, (name, err, scale)
            o += n


def test_train_call_matches_reference_golden(gold):
    """the reference's own PpoGaeAgent::train on a padded [4, 7] batch, epoch = 2 (tests/golden/agent_golden.txt)"""
    import torch
    from evomotion_amd import agent
    actor, critic = _modules(pattern=True)
    f, tr = _trainer(actor, critic, 28)
    t = lambda k: torch.from_numpy(gold[k]).cuda()
    tm = lambda x: x.transpose(0, 1).contiguous()  # [B, T, ...] -> time-major
    done = t("ppo_done")
    B, T = done.shape[:2]
    mask = torch.eq(torch.cat([torch.ones(B, 1, 1, device="cuda"), (1.0 - done)[:, : T - 1]], 1), 1.0)  # ppo_gae.cpp:127-132
    tr.train(tm(t("ppo_states")), tm(t("ppo_actions")), tm(t("ppo_rewards").squeeze(-1)), tm(done.squeeze(-1)).to(torch.uint8),
             tm(t("ppo_log_prob")), tm(t("ppo_curr_values").squeeze(-1)), tm(t("ppo_next_values").squeeze(-1)),
             tm(mask.squeeze(-1)).to(torch.uint8), epoch=2, learning_rate=1e-3, **HP)
    x = torch.from_numpy(gold["X"]).cuda()
    u = torch.full((8, 12), 0.5, device="cuda")
    _, _, value, mu, sigma = f.forward(x, uniform=u, want_dist=True)  # the rollout kernel already has the new weights
    np.testing.assert_allclose(mu.cpu().numpy(), gold["ppo_after_mu"], atol=5e-5)
    np.testing.assert_allclose(sigma.cpu().numpy(), gold["ppo_after_sigma"], atol=5e-5, rtol=5e-5)
    # The value head is ill-conditioned in this call by construction: returns = normalised advantages + V and V is the
    # critic's own output, so sum(value - returns) = -sum(adv) = 0 in exact arithmetic and the gradients of critic.6.bias
    # and critic.5.bias (LayerNorm beta) are pure rounding noise (~5e-8 here) that Adam turns into steps of up to lr each
    # (tests/diag/diag_ppo.py: every other gradient agrees with autograd to ~1e-6 relative).  Two epochs: 2 lr on the bias.
    np.testing.assert_allclose(value.cpu().numpy(), gold["ppo_after_value"].ravel(), atol=1e-3)
    tr.params_into(actor, critic)
    np.testing.assert_allclose(actor.head[0].weight[0].detach().cpu().numpy(), gold["ppo_after_actor_w0_row0"], atol=5e-6)
    assert np.abs(mu.cpu().numpy() - gold["mu"]).max() > 1e-4  # it moved


def test_epochs_in_lock_step_with_autograd_and_adam():
    """Four epochs on a rollout-sized batch (T = 8, N = 512).  Before every epoch the torch modules are set to the
    trainer's weights, so each epoch checks (a) the gradients against autograd at identical weights and (b) the clip +
    Adam step against clip_grad_norm_ + torch.optim.Adam fed with autograd's gradients."""
    import torch
    from evomotion_amd import agent
    from evomotion_amd._lib import lib, check
    from evomotion_amd.ppo import GRADS, PARAMS, ACTOR, CRITIC, _ptr
    actor, critic = _modules(seed=7)
    T, N = 8, 512
    rows = T * N
    states, actions, rewards, done, values, next_values, mask = _rollout(T, N, seed=9)
    f, tr = _trainer(actor, critic, rows)
    with torch.no_grad():
        mu, sigma = actor(states.reshape(rows, 371))
        logp = agent.truncated_normal_log_pdf(actions.reshape(rows, 12), mu, sigma) + 0.2 * (torch.rand(rows, 12, device="cuda") * 2 - 1)
    adv, ret, ng = tr.gae(rewards, done, values, next_values, mask, 0.99, 0.95)
    ng = float(ng)
    st, ac = states.reshape(rows, 371), actions.reshape(rows, 12)
    oa = torch.optim.Adam(actor.parameters(), lr=1e-3)
    oc = torch.optim.Adam(critic.parameters(), lr=1e-3)
    for ep in range(4):
        tr.params_into(actor, critic)
        check(lib.evm_ppo_grads(tr._h, rows, _ptr(st), _ptr(ac), _ptr(logp), _ptr(adv.reshape(-1)), _ptr(ret.reshape(-1)),
                                _ptr(mask.reshape(-1)), ng, HP["epsilon"], HP["entropy_factor"], HP["critic_loss_factor"], 0, tr._stream()))
        ga, gc = tr.vector(GRADS, ACTOR), tr.vector(GRADS, CRITIC)
        ra, rc, la, lc = _torch_grads(actor, critic, st, ac, logp, adv.reshape(-1), ret.reshape(-1), mask.reshape(-1), ng)
        assert float((ga - ra).abs().max()) <= 2e-5 * float(ra.abs().max()), ep
        assert float((gc - rc).abs().max()) <= 2e-5 * float(rc.abs().max()), ep
        check(lib.evm_ppo_apply(tr._h, 1e-3, HP["clip_grad_norm"], tr._stream()))
        torch.nn.utils.clip_grad_norm_(actor.parameters(), HP["clip_grad_norm"]); oa.step()
        torch.nn.utils.clip_grad_norm_(critic.parameters(), HP["clip_grad_norm"]); oc.step()
        for net, mod in ((ACTOR, actor), (CRITIC, critic)):
            d = (tr.vector(PARAMS, net) - torch.cat([p.detach().reshape(-1) for p in mod.parameters()])).abs()
            # a step is lr * m / (sqrt(v) + 1e-8): only where |g| is down at 1e-8 does rounding noise in g change it
            assert float(d.max()) <= 2.1e-3 and float((d > 2e-5).float().mean()) < 2e-4, (ep, net, float(d.max()))


def test_update_matches_torch_update():
    """the whole train() call (GAE + 3 epochs) against torch_ref.ppo_train with torch.optim.Adam, both free-running.  The
    clipped surrogate is discontinuous in the weights and the networks amplify weight noise through two LayerNorms
    (tests/diag/diag_ppo*.py), so after three epochs the outputs agree to ~1e-3, not to rounding."""
    import torch
    from evomotion_amd import agent
    actor, critic = _modules(seed=7)
    T, N = 8, 512
    states, actions, rewards, done, values, next_values, mask = _rollout(T, N, seed=9)
    f, tr = _trainer(actor, critic, T * N)
    with torch.no_grad():
        mu, sigma = actor(states.reshape(T * N, 371))
        logp = (agent.truncated_normal_log_pdf(actions.reshape(T * N, 12), mu, sigma)
                + 0.2 * (torch.rand(T * N, 12, device="cuda") * 2 - 1)).reshape(T, N, 12)
    la, lc = tr.train(states, actions, rewards, done, logp, values, next_values, mask, epoch=3, learning_rate=1e-3, **HP)
    oa = torch.optim.Adam(actor.parameters(), lr=1e-3)
    oc = torch.optim.Adam(critic.parameters(), lr=1e-3)
    tb = lambda x: x.transpose(0, 1).contiguous()
    mb = tb(mask).unsqueeze(-1) == 1
    dn = torch.where(mb, tb(done).float().unsqueeze(-1), torch.ones((), device="cuda"))
    x = states[0, :256]
    with torch.no_grad():
        mu0, _ = actor(x)
    ra, rc = torch_ref.ppo_train(actor, critic, oa, oc, tb(states), tb(actions), tb(rewards).unsqueeze(-1), dn, tb(logp),
                             tb(values).unsqueeze(-1), tb(next_values).unsqueeze(-1), mask=mb, epoch=3, **HP)
    assert abs(la - ra) < 1e-4 * max(1.0, abs(ra)) and abs(lc - rc) < 1e-4 * max(1.0, abs(rc))
    _, _, value, mu, sigma = f.forward(x, uniform=torch.full((256, 12), 0.5, device="cuda"), want_dist=True)
    actor.eval(); critic.eval()
    with torch.no_grad():
        rmu, rsig = actor(x)
        rv = critic(x).squeeze(-1)
    moved = float((rmu - mu0).abs().max())
    assert moved > 2e-2  # the update is much larger than the tolerance below
    np.testing.assert_allclose(mu.cpu().numpy(), rmu.cpu().numpy(), atol=2e-3)
    np.testing.assert_allclose(sigma.cpu().numpy(), rsig.cpu().numpy(), atol=2e-3, rtol=2e-3)
    np.testing.assert_allclose(value.cpu().numpy(), rv.cpu().numpy(), atol=2e-3)


def test_agent_update_save_load_round_trip(tmp_path):
    """VecPpoGaeAgent with the HIP update: weights and Adam state survive save() / load() (ppo_gae.cpp:192-204)"""
    import torch
    from evomotion_amd import VecPpoGaeAgent, VecRobotWalk
    from evomotion_amd.ppo import EXP_AVG, EXP_AVG_SQ, PARAMS
    env = VecRobotWalk(128, seed=3)
    env.reset()
    ag = VecPpoGaeAgent(11, [371], [12], horizon=8, epoch=2)
    ag.rollout(env)
    al, cl = ag.update()
    assert np.isfinite(al) and np.isfinite(cl)
    ag.save(str(tmp_path))
    ag2 = VecPpoGaeAgent(12, [371], [12], horizon=8, epoch=2)
    ag2.load(str(tmp_path))
    x = (torch.rand(64, 371, device="cuda") * 2 - 1)
    u = torch.rand(64, 12, device="cuda")
    for r, o in zip(ag.fused.forward(x, uniform=u), ag2.fused.forward(x, uniform=u)):
        assert torch.equal(r, o)
    # the next update continues from the same optimiser state: same rollout buffer -> same weights afterwards
    ag2._buf = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in ag._buf.items()}
    ag.update(); ag2.update()
    for net in (0, 1):
        assert ag._trainer.adam_step(net) == ag2._trainer.adam_step(net) == 4
        for what in (PARAMS, EXP_AVG, EXP_AVG_SQ):
            assert torch.equal(ag._trainer.vector(what, net), ag2._trainer.vector(what, net))


def test_full_size_gradients_are_the_sum_of_their_halves():
    """BASELINE size (4096 envs x horizon 32 = 131 072 rows): the losses are sums over rows divided by the global count, so
    the gradient of the whole batch equals the sum of the gradients of its two halves (each computed with the global
    count) — exercises the split-K reductions and every tile at full size without an oracle of that size."""
    import torch
    from evomotion_amd import agent
    from evomotion_amd._lib import lib, check
    from evomotion_amd.ppo import GRADS, ACTOR, CRITIC, _ptr
    rows = 131072
    actor, critic = _modules(seed=11)
    f, tr = _trainer(actor, critic, rows)
    g = torch.Generator(device="cuda"); g.manual_seed(12)
    states = (torch.rand(rows, 371, device="cuda", generator=g) * 2 - 1) * 1.5
    f.load_modules(actor, critic)
    u = torch.rand(rows, 12, device="cuda", generator=g)
    actions, logp, _ = f.forward(states, uniform=u)
    logp_old = logp + 0.3 * (torch.rand(rows, 12, device="cuda", generator=g) * 2 - 1)
    adv = torch.randn(rows, device="cuda", generator=g)
    returns = torch.randn(rows, device="cuda", generator=g)
    mask = (torch.rand(rows, device="cuda", generator=g) > 0.25).to(torch.uint8)
    n_glob = float(mask.sum())

    def grads(lo, hi):
        check(lib.evm_ppo_grads(tr._h, hi - lo, _ptr(states[lo:hi]), _ptr(actions[lo:hi]), _ptr(logp_old[lo:hi]), _ptr(adv[lo:hi]),
                                _ptr(returns[lo:hi]), _ptr(mask[lo:hi]), n_glob, HP["epsilon"], HP["entropy_factor"],
                                HP["critic_loss_factor"], 0, tr._stream()))
        la, lc = tr.losses()
        return tr.vector(GRADS, ACTOR).clone(), tr.vector(GRADS, CRITIC).clone(), la, lc

    fa, fc, la, lc = grads(0, rows)
    ha1, hc1, la1, lc1 = grads(0, rows // 2)
    ha2, hc2, la2, lc2 = grads(rows // 2, rows)
    assert abs(la - (la1 + la2)) < 1e-6 * max(1.0, abs(la)) and abs(lc - (lc1 + lc2)) < 1e-6 * max(1.0, abs(lc))
    for full, parts in ((fa, ha1 + ha2), (fc, hc1 + hc2)):
        assert torch.isfinite(full).all()
        assert float((full - parts).abs().max()) <= 2e-5 * float(full.abs().max())
    # and the same call twice gives the same bits (fixed reduction order)
    fa2, fc2, _, _ = grads(0, rows)
    assert torch.equal(fa, fa2) and torch.equal(fc, fc2)


def test_selected_rows_only_equals_all_rows_with_a_mask():
    """FusedPpoTrainer.train() runs its epochs on the selected rows alone (compact_rows, the default); rows outside the mask weigh
    nothing in either loss, so one epoch from the same weights must give the same gradients and losses as the pass over every row
    with the mask (to rounding: the same terms, summed in other tiles) — here with 40 % of the rows unselected, a ragged last
    tile, and the first / last rows unselected."""
    import torch
    from evomotion_amd import agent
    T, N = 8, 500
    states, actions, rewards, done, values, next_values, mask = _rollout(T, N, seed=11, p_invalid=0.4)
    mask[0, :7] = 0
    mask[-1, -9:] = 0
    res = []
    for compact in (False, True):
        actor, critic = _modules(seed=5)
        f, tr = _trainer(actor, critic, T * N)
        tr.compact_rows = compact
        with torch.no_grad():
            mu, sigma = actor(states.reshape(T * N, 371))
            logp = agent.truncated_normal_log_pdf(actions.reshape(T * N, 12), mu, sigma).reshape(T, N, 12) + 0.05
        la, lc = tr.train(states, actions, rewards, done, logp, values, next_values, mask, epoch=1, learning_rate=1e-3, **HP)
        res.append((la, lc, tr.vector(1, 0).clone(), tr.vector(1, 1).clone()))
    (la0, lc0, ga0, gc0), (la1, lc1, ga1, gc1) = res
    assert abs(la0 - la1) <= 2e-5 * max(1.0, abs(la0)) and abs(lc0 - lc1) <= 2e-5 * max(1.0, abs(lc0)), (la0, la1, lc0, lc1)
    for g0, g1 in ((ga0, ga1), (gc0, gc1)):
        assert float(g0.abs().max()) > 0
        assert float((g0 - g1).abs().max()) <= 2e-5 * float(g0.abs().max()), float((g0 - g1).abs().max()) / float(g0.abs().max())


def test_select_rows_gathers_exactly_the_masked_rows_in_order():
    """evm_ppo_select_rows against torch.index_select: ragged sizes (the scan deals 1024 slices), nothing / everything / one row
    selected; the dense copies are bit-equal to the source rows, in source order, and the mask handed back is all ones."""
    import ctypes
    import torch
    actor, critic = _modules(seed=1)
    f, tr = _trainer(actor, critic, 5000)
    g = torch.Generator(device="cuda"); g.manual_seed(17)

    class Dev:  # a device buffer of the trainer, seen through the CUDA array interface (no copy, no second HIP runtime)
        def __init__(self, addr, shape, typestr):
            self.__cuda_array_interface__ = {"data": (int(addr.value), False), "shape": tuple(shape), "typestr": typestr, "version": 2}

    def view(addr, shape, dtype=torch.float32):
        torch.cuda.current_stream().synchronize()
        return torch.as_tensor(Dev(addr, shape, "|u1" if dtype == torch.uint8 else "<f4"), device="cuda").clone()

    for rows, p_sel in ((1, 1.0), (37, 0.5), (1024, 0.5), (1025, 0.7), (4999, 0.6), (5000, 1.0), (777, 0.0), (3000, None)):
        st = torch.rand(rows, 371, device="cuda", generator=g)
        ac = torch.rand(rows, 12, device="cuda", generator=g)
        lp = torch.rand(rows, 12, device="cuda", generator=g)
        adv = torch.rand(rows, device="cuda", generator=g)
        ret = torch.rand(rows, device="cuda", generator=g)
        if p_sel is None:
            mask = torch.zeros(rows, dtype=torch.uint8, device="cuda"); mask[1234] = 1
        else:
            mask = (torch.rand(rows, device="cuda", generator=g) < p_sel).to(torch.uint8)
        n, (s_st, s_ac, s_lp, s_adv, s_ret, s_mask) = tr.select_rows(st, ac, lp, adv, ret, mask)
        idx = torch.nonzero(mask).squeeze(1)
        assert n == idx.numel(), (rows, n, idx.numel())
        if n == 0:
            continue
        assert torch.equal(view(s_st, (n, 371)), st.index_select(0, idx))
        assert torch.equal(view(s_ac, (n, 12)), ac.index_select(0, idx))
        assert torch.equal(view(s_lp, (n, 12)), lp.index_select(0, idx))
        assert torch.equal(view(s_adv, (n,)), adv.index_select(0, idx))
        assert torch.equal(view(s_ret, (n,)), ret.index_select(0, idx))
        assert bool((view(s_mask, (n,), torch.uint8) == 1).all())


def test_an_empty_selection_is_a_no_op_that_says_so():
    """a rollout whose mask selects no transition must not move anything.  The count of selected transitions stays on
    the device (no host read in the update), so the no-op happens there: after one real update (the Adam moments are non-zero:
    residual momentum exists) an update with an all-zero mask leaves weights and both moments bit for bit where they were and
    reports NaN losses; a real update afterwards works as before."""
    import math
    import torch
    from evomotion_amd import agent
    actor, critic = _modules(seed=3)
    T, N = 4, 256
    states, actions, rewards, done, values, next_values, mask = _rollout(T, N, seed=4)
    f, tr = _trainer(actor, critic, T * N)
    with torch.no_grad():
        mu, sigma = actor(states.reshape(T * N, 371))
        logp = agent.truncated_normal_log_pdf(actions.reshape(T * N, 12), mu, sigma).reshape(T, N, 12)
    la, lc = tr.train(states, actions, rewards, done, logp, values, next_values, mask, epoch=2, learning_rate=1e-3, **HP)
    assert math.isfinite(la) and math.isfinite(lc)
    before = [[tr.vector(w, net).clone() for w in (0, 2, 3)] for net in (0, 1)]
    assert float(before[0][1].abs().max()) > 0 and float(before[1][2].abs().max()) > 0        # momentum to drift on
    la0, lc0 = tr.train(states, actions, rewards, done, logp, values, next_values, torch.zeros_like(mask), epoch=3, learning_rate=1e-3, **HP)
    assert math.isnan(la0) and math.isnan(lc0)
    after = [[tr.vector(w, net) for w in (0, 2, 3)] for net in (0, 1)]
    for net in (0, 1):
        for b, a_ in zip(before[net], after[net]):
            assert torch.equal(b, a_)
    la2, lc2 = tr.train(states, actions, rewards, done, logp, values, next_values, mask, epoch=1, learning_rate=1e-3, **HP)
    assert math.isfinite(la2) and math.isfinite(lc2) and not torch.equal(tr.vector(0, 0), before[0][0])
