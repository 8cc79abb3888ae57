"""Diagnostic: HIP PPO update vs torch autograd + torch.optim.Adam, epoch by epoch (parameter differences per tensor)."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from evomotion_amd import agent
from evomotion_amd.ppo import GRADS, PARAMS, ACTOR, CRITIC
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
oa = torch.optim.Adam(actor.parameters(), lr=1e-3); oc = torch.optim.Adam(critic.parameters(), lr=1e-3)
for ep in range(3):
    tr.epoch(st, ac, lp, adv.reshape(-1), ret.reshape(-1), mask.reshape(-1), ng, HP["epsilon"], HP["entropy_factor"], HP["critic_loss_factor"], 1e-3, 0.5)
    ga, gc = tr.vector(GRADS, ACTOR), tr.vector(GRADS, CRITIC)
    ra, rc, la, lc = T._torch_grads(actor, critic, st, ac, lp, adv.reshape(-1), ret.reshape(-1), mask.reshape(-1), ng)
    print("epoch", ep, "grad norms torch %.6f %.6f hip %.6f %.6f  max grad err %.3e %.3e" % (float(ra.norm()), float(rc.norm()), float(ga.norm()), float(gc.norm()), float((ga-ra).abs().max()), float((gc-rc).abs().max())))
    torch.nn.utils.clip_grad_norm_(actor.parameters(), 0.5); oa.step()
    torch.nn.utils.clip_grad_norm_(critic.parameters(), 0.5); oc.step()
    for net, mod, nm in ((ACTOR, actor, "actor"), (CRITIC, critic, "critic")):
        th = tr.vector(PARAMS, net); o = 0
        for name, p in mod.named_parameters():
            n = p.numel(); d = (th[o:o+n] - p.detach().reshape(-1)).abs()
            print("   %-7s %-16s max dtheta %.3e  >1e-4: %d  >1e-5: %d of %d" % (nm, name, float(d.max()), int((d > 1e-4).sum()), int((d > 1e-5).sum()), n))
            o += n
x = states[0, :256]
_, _, value, mu_k, sg_k = f.forward(x, uniform=torch.full((256, 12), 0.5, device="cuda"), want_dist=True)
actor.eval(); critic.eval()
with torch.no_grad():
    rmu, rsig = actor(x)
a3, c3 = copy.deepcopy(actor), copy.deepcopy(critic)
tr.params_into(a3, c3)
with torch.no_grad():
    mmu, msig = a3(x)
print("kernel vs torch-updated module:", float((mu_k - rmu).abs().max()), " hip-theta-in-torch vs torch-updated:", float((mmu - rmu).abs().max()),
      " kernel vs hip-theta-in-torch:", float((mu_k - mmu).abs().max()))
