"""Diagnostic (-DEVM_PSTAMPS build): phase cycles of k_policy_forward, wave 0 of the actor workgroups (mean over tiles).
    make -C evomotion_amd/csrc pstamps && cp build/libevm_pstamps.so evomotion_amd/libevomotion_hip.so && python tools/pstamps.py [16|32]
(16: the 16-row form, k_policy_forward16; stamps of its first 256 actor tiles)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from evomotion_amd import FusedActorCritic, ActorModule, CriticModule
n = 4096
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nb = 128 if rows == 32 else 256
ao = len(sys.argv) > 2  # any third argument: actor only (half the workgroups)
pol = FusedActorCritic(371, 12, 256, 0)
torch.manual_seed(0)
pol.load_modules(ActorModule([371], [12], 256).cuda(), CriticModule([371], 256).cuda())
pol.set_tile_rows(rows)
obs = torch.randn(n, 371, device="cuda")
acc = np.zeros(6)
K = 20
for k in range(K + 5):
    out = pol.forward(obs, seed=k, want_dist=True, actor_only=ao)
    torch.cuda.synchronize()
    if k >= 5:
        st = out[3].view(torch.int64).cpu().numpy().reshape(-1)[: nb * 8].reshape(nb, 8)[:, :7].astype(np.float64)
        acc += np.diff(st, axis=1).mean(axis=0)
acc /= K
for name, v in zip(["stage observations", "layer 1 GEMM", "epilogue 1 (Mish + LayerNorm)", "layer 2 GEMM", "epilogue 2", "heads"], acc):
    print("%-32s %8.0f cycles" % (name, v))
print("sum %.0f cycles (%d-row tiles%s)" % (acc.sum(), rows, ", actor only" if ao else ""))
