"""PyTorch-autograd restatements of the reference's update steps — TEST INFRASTRUCTURE (comparison side only).

  gae_advantages / ppo_train     PpoGaeAgent::train, evo_motion_networks/src/agents/ppo_gae.cpp:117-190
  sac_train                      SoftActorCriticAgent::train, src/agents/soft_actor_critic.cpp:93-170

Pinned on the CPU to the reference's own golden train() calls (tests/test_agent_host.py, tests/test_sac_host.py); the HIP
trainers (evm_ppo_*, evm_q_*, evm_sac_*) are compared against them on the GPU.  The product (evomotion_amd/) has one
backend, the HIP one, and never imports this module."""
import torch
from torch import nn

from evomotion_amd.agent import truncated_normal_entropy, truncated_normal_log_pdf, truncated_normal_sample
from evomotion_amd.sac import soft_update


def _dist_ready():
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


def masked_mean_std(x, mask):
    """mean and unbiased std of x[mask] over ALL ranks (one all_gather of three numbers per rank; Chan's merge)."""
    sel = torch.masked_select(x, mask)
    if not _dist_ready():  # single process: literally the reference's two calls (ppo_gae.cpp:148-149)
        return sel.mean(), sel.std(), torch.tensor(float(sel.numel()), device=x.device, dtype=torch.float64)
    n = torch.tensor(float(sel.numel()), device=x.device, dtype=torch.float64)
    mean = sel.double().mean() if sel.numel() else torch.zeros((), device=x.device, dtype=torch.float64)
    m2 = ((sel.double() - mean) ** 2).sum() if sel.numel() else torch.zeros((), device=x.device, dtype=torch.float64)
    if _dist_ready():
        trip = torch.stack([n, mean, m2])
        allt = [torch.zeros_like(trip) for _ in range(torch.distributed.get_world_size())]
        torch.distributed.all_gather(allt, trip)
        n, mean, m2 = allt[0]
        for t in allt[1:]:
            nb, mb, m2b = t
            tot = n + nb
            if float(tot) == 0:
                continue
            d = mb - mean
            mean = mean + d * nb / tot
            m2 = m2 + m2b + d * d * n * nb / tot
            n = tot
    std = torch.sqrt(m2 / torch.clamp(n - 1, min=1.0))
    return mean.to(x.dtype), std.to(x.dtype), n


def gae_advantages(rewards, done, curr_values, next_values, gamma, lam, mask=None):
    """[B,T,1] tensors -> (mask, normalised advantages, returns).

    mask=None reproduces the reference exactly: trajectories are padded with done = 1 and the mask is the
    shifted done flag (ppo_gae.cpp:127-132).  The vectorised rollout passes an explicit transition mask instead
    (rows contain settle steps in the middle, not only trailing padding); a masked step contributes nothing and
    stops the backward recursion, which is what the shifted mask does for trailing padding."""
    B, T = rewards.shape[:2]
    explicit = mask is not None
    if not explicit:
        mask = torch.eq(torch.cat([torch.ones(B, 1, 1, device=rewards.device), (1.0 - done)[:, : T - 1]], 1), 1.0)
    deltas = rewards + (1.0 - done) * gamma * next_values - curr_values
    g = torch.zeros(B, 1, device=rewards.device)
    adv = []
    for t in range(T - 1, -1, -1):
        g = deltas[:, t] * mask[:, t] + gamma * lam * (1.0 - done[:, t]) * g
        if explicit:
            g = g * mask[:, t]
        adv.append(g)
    adv = torch.stack(adv, 1).flip([1])
    mean, std, _ = masked_mean_std(adv, mask)
    adv = (adv - mean) / (std + 1e-8)
    return mask, adv, adv + curr_values  # returns = NORMALISED advantages + V


def _all_reduce_grads(params):
    if not _dist_ready():
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    torch.distributed.all_reduce(flat)  # losses are normalised by the global count, so SUM is the global gradient
    o = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[o:o + n].view_as(p))
        o += n


def _all_reduce_grads_mean(params):
    """Losses that are local means (SAC, soft_actor_critic.cpp:127-153): equal shards per rank, so the global-mean
    gradient is the rank average."""
    if not _dist_ready():
        return
    _all_reduce_grads(params)
    w = torch.distributed.get_world_size()
    for p in params:
        p.grad.div_(w)


def ppo_train(actor, critic, actor_opt, critic_opt, states, actions, rewards, done, log_prob, curr_values, next_values,
              gamma, lam, epsilon, entropy_factor, critic_loss_factor, epoch, clip_grad_norm, mask=None):
    """One PpoGaeAgent::train() call on padded [B,T,*] tensors; returns the last (actor_loss, critic_loss)."""
    actor.train()
    critic.train()
    mask, adv, returns = gae_advantages(rewards, done, curr_values, next_values, gamma, lam, mask)
    adv, returns = adv.detach(), returns.detach()
    n_local = mask.sum()
    n_glob = n_local.double().clone()
    if _dist_ready():
        torch.distributed.all_reduce(n_glob)
    scale = (n_local.double() / n_glob).float() if _dist_ready() else None
    a_loss = c_loss = None
    for _ in range(epoch):
        mu, sigma = actor(states)
        lp = truncated_normal_log_pdf(actions, mu, sigma)
        ent = truncated_normal_entropy(mu, sigma)
        value = critic(states)
        ratios = torch.exp(lp - log_prob)
        s1 = ratios * adv
        s2 = torch.clamp(ratios, 1.0 - epsilon, 1.0 + epsilon) * adv
        a_loss = -torch.mean(torch.masked_select(torch.min(s1, s2) + entropy_factor * ent, mask.expand_as(ent)))
        actor_opt.zero_grad()
        (a_loss * scale if scale is not None else a_loss).backward()
        _all_reduce_grads(list(actor.parameters()))
        nn.utils.clip_grad_norm_(actor.parameters(), clip_grad_norm)
        actor_opt.step()
        c_loss = critic_loss_factor * torch.mean(torch.masked_select(torch.pow(value - returns, 2.0), mask))
        critic_opt.zero_grad()
        (c_loss * scale if scale is not None else c_loss).backward()
        _all_reduce_grads(list(critic.parameters()))
        nn.utils.clip_grad_norm_(critic.parameters(), clip_grad_norm)
        critic_opt.step()
    return float(a_loss.detach()), float(c_loss.detach())



def sac_train(actor, critic_1, critic_2, target_critic_1, target_critic_2, entropy, actor_opt, critic_1_opt, critic_2_opt,
              entropy_opt, states, actions, rewards, done, next_states, gamma, tau, target_entropy,
              u_next=None, u_curr=None, grad_hook=None):
    """One SoftActorCriticAgent::train() call (soft_actor_critic.cpp:93-170).  `u_next` / `u_curr` supply the two
    at::rand draws; `grad_hook(params)` is called between backward and step (data-parallel all-reduce)."""
    with torch.no_grad():
        next_mu, next_sigma = actor(next_states)
        next_action = truncated_normal_sample(next_mu, next_sigma, -1.0, 1.0, u=u_next)
        next_logp = truncated_normal_log_pdf(next_action, next_mu, next_sigma, -1.0, 1.0).sum(-1, keepdim=True)
        tq = torch.min(target_critic_1(next_states, next_action), target_critic_2(next_states, next_action))
        target_v = tq - entropy.alpha() * next_logp
        target_q = rewards + (1.0 - done) * gamma * target_v

    def step(opt, loss, params):
        opt.zero_grad()
        loss.backward()
        if grad_hook is not None:
            grad_hook(params)
        opt.step()

    loss_c1 = torch.nn.functional.mse_loss(critic_1(states, actions), target_q)
    step(critic_1_opt, loss_c1, list(critic_1.parameters()))
    loss_c2 = torch.nn.functional.mse_loss(critic_2(states, actions), target_q)
    step(critic_2_opt, loss_c2, list(critic_2.parameters()))

    mu, sigma = actor(states)
    curr_action = truncated_normal_sample(mu, sigma, -1.0, 1.0, u=u_curr)
    curr_logp = truncated_normal_log_pdf(curr_action, mu, sigma, -1.0, 1.0).sum(-1, keepdim=True)
    q = torch.min(critic_1(states, curr_action), critic_2(states, curr_action))
    loss_actor = torch.mean(entropy.alpha().detach() * curr_logp - q)
    step(actor_opt, loss_actor, list(actor.parameters()))

    loss_entropy = -torch.mean(entropy.log_alpha * (curr_logp.detach() + target_entropy))
    step(entropy_opt, loss_entropy, list(entropy.parameters()))

    soft_update(target_critic_1, critic_1, tau)
    soft_update(target_critic_2, critic_2, tau)
    return dict(actor=loss_actor.detach(), critic_1=loss_c1.detach(), critic_2=loss_c2.detach(), entropy=loss_entropy.detach())


