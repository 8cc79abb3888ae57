"""Does the PPO path learn?  Mean reward per valid transition (robot_walk: the root's forward velocity) and mean episode length
(do_step transitions per episode end) over windows of the training run.  A functional check, not a benchmark.
    python tools/learning_curve.py [updates] [lr]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evomotion_amd import VecRobotWalk, VecPpoGaeAgent
n = 4096
updates = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-4
env = VecRobotWalk(n, seed=5)
env.reset(); env.stagger_episodes()
agent = VecPpoGaeAgent(11, [env.state_dim], [env.action_dim], hidden_size=256, device=0, horizon=32, epoch=8, learning_rate=lr)
win = max(1, updates // 10)
acc_r = acc_n = acc_d = 0.0
t0 = time.time()
for k in range(updates):
    b = agent.rollout(env)
    m = b["valid_u8"] == 1
    acc_r += float(b["rewards"][m].sum()); acc_n += float(m.sum()); acc_d += float((b["done_u8"][m] != 0).sum())
    agent.update()
    if (k + 1) % win == 0:
        print("updates %5d..%5d: mean reward %.4f per step, mean episode length %.1f steps   (%.0f s)" %
              (k + 1 - win, k + 1, acc_r / max(acc_n, 1), acc_n / max(acc_d, 1), time.time() - t0), flush=True)
        acc_r = acc_n = acc_d = 0.0
