"""Soft actor-critic rows of the path — mirrors of the reference's evo_motion_networks pieces.

  QNetworkModule                 evo_motion_networks/src/networks/q_net.cpp:8-43
  EntropyParameter               evo_motion_networks/src/networks/entropy.cpp:7-15
  hard_update / soft_update      evo_motion_networks/src/functions.cpp:151-171
  SoftActorCriticAgent::train    evo_motion_networks/src/agents/soft_actor_critic.cpp:93-170
  SoftActorCriticAgent::act      soft_actor_critic.cpp:47-62 (batched: fused HIP actor forward + device replay ring)
  ReplayBuffer                   evo_motion_networks/src/replay_buffer.cpp:16-52,146-153 -> evm_replay_* (HIP)

The update runs on PyTorch-ROCm autograd over mirrors with the reference's parameter names and reproduces the
reference's post-train() outputs (tests/golden/sac_golden.txt, tests/test_sac_host.py).
"""
import math

import torch
from torch import nn

from .agent import ActorModule, init_weights, truncated_normal_log_pdf, truncated_normal_sample


class QNetworkModule(nn.Module):
    """q_network.0 … q_network.9 as in the reference: three Linear→Mish→LayerNorm blocks and a Linear(hidden, 1)."""

    def __init__(self, state_space, action_space, hidden_size):
        super().__init__()
        h = hidden_size
        self.q_network = nn.Sequential(
            nn.Linear(state_space[0] + action_space[0], h), nn.Mish(), nn.LayerNorm(h, eps=1e-5),
            nn.Linear(h, h), nn.Mish(), nn.LayerNorm(h, eps=1e-5),
            nn.Linear(h, h), nn.Mish(), nn.LayerNorm(h, eps=1e-5),
            nn.Linear(h, 1))
        self.apply(init_weights)

    def forward(self, state, action):
        return self.q_network(torch.cat([state, action], -1))


class EntropyParameter(nn.Module):
    def __init__(self, initial_alpha=1.0, nb_parameters=1):
        super().__init__()
        self.log_alpha = nn.Parameter(torch.full((nb_parameters,), math.log(initial_alpha)))

    def alpha(self):
        return self.log_alpha.exp()


def hard_update(to, frm):
    with torch.no_grad():
        for (_, t), (_, f) in zip(to.named_parameters(), frm.named_parameters()):
            t.copy_(f)


def soft_update(to, frm, tau):
    """to <- tau * from + (1 - tau) * to, with the reference's double-precision (1.0 - tau) (functions.cpp:169)."""
    with torch.no_grad():
        for (_, t), (_, f) in zip(to.named_parameters(), frm.named_parameters()):
            t.copy_(tau * f + (1.0 - tau) * t)


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
