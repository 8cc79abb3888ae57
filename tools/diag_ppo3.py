import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from evomotion_amd import agent
from evomotion_amd.ppo import GRADS, PARAMS, ACTOR, CRITIC
import test_gpu_ppo as TT
HP = TT.HP
actor, critic = TT._modules(seed=7)
T, N = 8, 512
states, actions, rewards, done, values, next_values, mask = TT._rollout(T, N, seed=9)
f, tr = TT._trainer(actor, critic, T * N)
with torch.no_grad():
    mu, sigma = actor(states.reshape(T * N, 371))
    logp = (agent.truncated_normal_log_pdf(actions.reshape(T * N, 12), mu, sigma)
            + 0.2 * (torch.rand(T * N, 12, device="cuda") * 2 - 1)).reshape(T, N, 12)
EP = int(sys.argv[1]) if len(sys.argv) > 1 else 3
la, lc = tr.train(states, actions, rewards, done, logp, values, next_values, mask, epoch=EP, learning_rate=1e-3, **HP)
oa = torch.optim.Adam(actor.parameters(), lr=1e-3)
oc = torch.optim.Adam(critic.parameters(), lr=1e-3)
tb = lambda x: x.transpose(0, 1).contiguous()
mb = tb(mask).unsqueeze(-1) == 1
dn = torch.where(mb, tb(done).float().unsqueeze(-1), torch.ones((), device="cuda"))
ra, rc = agent.ppo_train(actor, critic, oa, oc, tb(states), tb(actions), tb(rewards).unsqueeze(-1), dn, tb(logp),
                         tb(values).unsqueeze(-1), tb(next_values).unsqueeze(-1), mask=mb, epoch=EP, **HP)
print("losses", la, ra, lc, rc)
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
    rv = critic(x).squeeze(-1)
a3, c3 = copy.deepcopy(actor), copy.deepcopy(critic)
tr.params_into(a3, c3)
with torch.no_grad():
    mmu, msig = a3(x)
d = (mu_k - rmu).abs()
print("kernel vs torch-updated:", float(d.max()), "count>3e-4", int((d > 3e-4).sum()), " hip-theta-in-torch vs torch-updated:", float((mmu - rmu).abs().max()),
      " kernel vs hip-theta-in-torch:", float((mu_k - mmu).abs().max()), "value", float((value - rv).abs().max()))
i = int(d.argmax()) // 12
print("row", i, "mu_k", mu_k[i].tolist(), "\nrmu", rmu[i].tolist())
for name, p in actor.named_parameters():
    a4 = copy.deepcopy(actor)
    with torch.no_grad():
        dict(a4.named_parameters())[name].copy_(dict(a3.named_parameters())[name])
        m4, _ = a4(x)
    dd = (dict(a3.named_parameters())[name] - p).abs()
    print("swap %-16s -> out diff %.3e   (param diff max %.3e, mean %.3e)" % (name, float((m4 - rmu).abs().max()), float(dd.max()), float(dd.mean())))
