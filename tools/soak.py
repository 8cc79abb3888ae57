"""Soak run on the GPU box: a long random-action rollout and a long PPO training run at 4096 envs, checked for finite states,
bounded poses, bounded solver residual and finite weights.  Not a benchmark.
    python tools/soak.py [dynamics steps] [ppo steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evomotion_amd import VecRobotWalk, VecPpoGaeAgent
n = 4096
n_dyn = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
n_ppo = int(sys.argv[2]) if len(sys.argv) > 2 else 6400
env = VecRobotWalk(n, seed=99)
env.reset(); env.stagger_episodes()
g = torch.Generator(device="cuda"); g.manual_seed(1)
bank = [torch.rand(n, 12, device="cuda", generator=g) * 2 - 1 for _ in range(61)]
t0 = time.time(); worst_res = 0.0; episodes = 0
for k in range(n_dyn):
    st = env.step_autoreset(bank[k % 61])
    if k % 2000 == 1999:
        assert torch.isfinite(st.state).all() and torch.isfinite(st.reward).all(), k
        p = env.body_poses()
        assert p[..., :3].abs().max() < 50.0 and ((p[..., 3:].norm(dim=-1) - 1).abs().max() < 1e-4), k
        worst_res = max(worst_res, float(env.residual()))
episodes = env.stats()
errs = env.errors()   # (version waits that timed out, manifolds left out of a step): both must stay zero
print("dynamics: %d calls x %d envs in %.1f s, stats %s, worst batch residual %.3g, error counters %s, member pairs %d"
      % (n_dyn, n, time.time() - t0, episodes, worst_res, errs, env.n_pairs))
assert errs[0] == 0, errs
q = env.penetration_queries(); sp = env.speculation_counters()
print("penetration-solver queries %d (predicted %d); speculation blocks: %d runs, %d answers used, %d waits ran out" % (q, env.predicted_penetration_queries, sp[0], sp[1], sp[2]))
assert sp[2] == 0, sp
agent = VecPpoGaeAgent(7, [env.state_dim], [env.action_dim], hidden_size=256, device=0, horizon=32, epoch=8, learning_rate=3e-4)
t0 = time.time(); updates = 0
for k in range(n_ppo // 32):
    agent.rollout(env); agent.update(); updates += 1
torch.cuda.synchronize()
for net in (0, 1):
    from evomotion_amd.ppo import PARAMS
    w = agent._trainer.vector(PARAMS, net)
    assert torch.isfinite(w).all(), net
print("ppo: %d rollout steps, %d updates (8 epochs each) in %.1f s, weights finite, |theta|_max %.3g" % (n_ppo, updates, time.time() - t0, float(w.abs().max())))
print("soak ok")
