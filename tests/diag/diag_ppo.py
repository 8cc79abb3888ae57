import os, sys, copy
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import agent_oracle as ao, golden_io
from evomotion_amd import agent
from evomotion_amd.ppo import FusedPpoTrainer, GRADS, PARAMS, ACTOR, CRITIC, _ptr
from evomotion_amd._lib import lib, check
import test_gpu_ppo as T
gold = golden_io.load()
actor, critic = T._modules(pattern=True)
f, tr = T._trainer(actor, critic, 28)
t = lambda k: torch.from_numpy(gold[k]).cuda()
tm = lambda x: x.transpose(0, 1).contiguous()
done = t("ppo_done"); B, TT = done.shape[:2]
mask = torch.eq(torch.cat([torch.ones(B, 1, 1, device="cuda"), (1.0 - done)[:, : TT - 1]], 1), 1.0)
HP = T.HP
# torch path (GPU)
a2, c2 = copy.deepcopy(actor), copy.deepcopy(critic)
# epoch-1 gradients
import torch_ref
mk, adv, ret = torch_ref.gae_advantages(t("ppo_rewards"), done, t("ppo_curr_values"), t("ppo_next_values"), 0.99, 0.95)
adv_m, ret_m, ng = tr.gae(tm(t("ppo_rewards").squeeze(-1)), tm(done.squeeze(-1)).to(torch.uint8), tm(t("ppo_curr_values").squeeze(-1)), tm(t("ppo_next_values").squeeze(-1)), tm(mask.squeeze(-1)).to(torch.uint8), 0.99, 0.95)
print("n", ng, "adv err", float((tm(adv.squeeze(-1)) - adv_m)[tm(mask.squeeze(-1))].abs().max()))
rows = 28
st = tm(t("ppo_states")).reshape(rows, 371); ac = tm(t("ppo_actions")).reshape(rows, 12); lp = tm(t("ppo_log_prob")).reshape(rows, 12)
mk8 = tm(mask.squeeze(-1)).to(torch.uint8).reshape(-1)
check(lib.evm_ppo_grads(tr._h, rows, _ptr(st), _ptr(ac), _ptr(lp), _ptr(adv_m.reshape(-1)), _ptr(ret_m.reshape(-1)), _ptr(mk8), ng, HP["epsilon"], HP["entropy_factor"], HP["critic_loss_factor"], 0, tr._stream()))
ga, gc = tr.vector(GRADS, ACTOR), tr.vector(GRADS, CRITIC)
ra, rc, la, lc = T._torch_grads(a2, c2, st, ac, lp, adv_m.reshape(-1), ret_m.reshape(-1), mk8, ng)
for flat, ref, mod, nm in ((ga, ra, a2, "actor"), (gc, rc, c2, "critic")):
    o = 0
    print(nm, "grad norm", float(ref.norm()), float(flat.norm()))
    for name, p in mod.named_parameters():
        n = p.numel(); a, b = flat[o:o+n], ref[o:o+n]
        small = (b.abs() < 1e-7).sum().item()
        flips = ((a * b) < 0).sum().item()
        print("  %-18s max|g| %.3e  max err %.3e  |g|<1e-7: %d  sign flips: %d" % (name, float(b.abs().max()), float((a-b).abs().max()), small, flips))
        o += n
