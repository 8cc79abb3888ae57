"""Does the PPO path learn?  Mean reward per valid transition (robot_walk: the root's forward velocity) and mean episode length
(do_step transitions per episode end) over windows of the training run.  A functional check, not a benchmark.
    python tools/learning_curve.py [updates] [lr]          PPO (VecPpoGaeAgent)
    python tools/learning_curve.py sac [steps] [lr]        SAC (VecSacAgent: 4096-row update every 4 steps)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evomotion_amd import VecRobotWalk, VecPpoGaeAgent
n = 4096
if len(sys.argv) > 1 and sys.argv[1] == "sac":
    from evomotion_amd import VecSacAgent
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    lr = float(sys.argv[3]) if len(sys.argv) > 3 else 3e-4
    env = VecRobotWalk(n, seed=5)
    env.reset(); env.stagger_episodes()
    sac = VecSacAgent(11, [env.state_dim], [env.action_dim], batch_size=4096, epoch=1, learning_rate=lr, replay_buffer_size=1024,
                      train_every=4, n_envs=n, device=0)
    win = max(1, steps // 10)
    acc_r = torch.zeros((), device="cuda"); acc_n = torch.zeros((), device="cuda"); acc_d = torch.zeros((), device="cuda")
    t0 = time.time()
    for k in range(steps):
        st = sac.step(env)
        m = st.valid == 1
        acc_r += (st.reward * m).sum(); acc_n += m.sum(); acc_d += ((st.done != 0) & m).sum()
        if (k + 1) % win == 0:
            print("steps %6d..%6d: mean reward %.4f per step, mean episode length %.1f steps   (%.0f s)" %
                  (k + 1 - win, k + 1, float(acc_r / acc_n.clamp(min=1)), float(acc_n / acc_d.clamp(min=1)), time.time() - t0), flush=True)
            acc_r.zero_(); acc_n.zero_(); acc_d.zero_()
    sys.exit(0)
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
