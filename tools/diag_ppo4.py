"""Diagnostic: gradient parity of the HIP PPO update at every epoch with the torch modules forced to the trainer's weights."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from evomotion_amd import agent
from evomotion_amd.ppo import GRADS, PARAMS, ACTOR, CRITIC, _ptr
from evomotion_amd._lib import lib, check
import test_gpu_ppo as T
HP = T.HP
actor, critic = T._modules(seed=7)
Tn, N = 8, 512
states, actions, rewards, done, values, next_values, mask = T._rollout(Tn, N, seed=9)
f, tr = T._trainer(actor, critic, Tn * N)
with torch.no_grad():
    mu, sigma = actor(states.reshape(Tn * N, 371))
    logp = (agent.truncated_normal_log_pdf(actions.reshape(Tn * N, 12), mu, sigma) + 0.2 * (torch.rand(Tn * N, 12, device="cuda") * 2 - 1)).reshape(Tn, N, 12)
adv, ret, ng = tr.gae(rewards, done, values, next_values, mask, 0.99, 0.95)
rows = Tn * N
st, ac, lp = states.reshape(rows, 371), actions.reshape(rows, 12), logp.reshape(rows, 12)
for ep in range(4):
    tr.params_into(actor, critic)
    check(lib.evm_ppo_grads(tr._h, rows, _ptr(st), _ptr(ac), _ptr(lp), _ptr(adv.reshape(-1)), _ptr(ret.reshape(-1)), _ptr(mask.reshape(-1)), ng,
                            HP["epsilon"], HP["entropy_factor"], HP["critic_loss_factor"], 0, tr._stream()))
    ga, gc = tr.vector(GRADS, ACTOR), tr.vector(GRADS, CRITIC)
    ra, rc, la, lc = T._torch_grads(actor, critic, st, ac, lp, adv.reshape(-1), ret.reshape(-1), mask.reshape(-1), ng)
    print("epoch", ep, "norms %.6f %.6f | max grad err actor %.3e critic %.3e | max |g| %.3e %.3e" % (float(ra.norm()), float(rc.norm()), float((ga-ra).abs().max()), float((gc-rc).abs().max()), float(ra.abs().max()), float(rc.abs().max())))
    check(lib.evm_ppo_apply(tr._h, 1e-3, 0.5, tr._stream()))
