"""Larger batches than the speculation slots cover (GPU box): step time, solver queries and speculation counters.
    python tools/big_batch.py <n_envs>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evomotion_amd import VecRobotWalk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
env = VecRobotWalk(n, seed=5); env.reset(); env.stagger_episodes()
print("created", n, flush=True)
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.rand(n, 12, device="cuda", generator=g) * 2 - 1 for _ in range(4)]
for k in range(100): env.step_autoreset(acts[k % 4])
torch.cuda.synchronize(); print("100 steps", flush=True)
env.penetration_queries(); print("pq", flush=True); env.speculation_counters(); print("sc", flush=True)
torch.cuda.synchronize(); t0 = time.time()
K = 200
for k in range(K): env.step_autoreset(acts[k % 4])
torch.cuda.synchronize(); dt = time.time() - t0
q = env.penetration_queries(); s = env.speculation_counters()
print("n=%d: %.3f ms per step; solver queries %.1f per step (predicted %.1f, urgent entries %.0f); speculation runs %.1f, used %.1f, waits run out %d; errors %s"
      % (n, dt / K * 1e3, q / K, env.predicted_penetration_queries / K, env.urgent_entries / K, s[0] / K, s[1] / K, s[2], env.errors()))
