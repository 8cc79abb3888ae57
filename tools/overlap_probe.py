"""How much of the policy forward hides behind a dynamics step when both run on their own stream?  (feasibility of starting the
next step's action-independent setup kernels while the policy network runs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evomotion_amd import VecRobotWalk, FusedActorCritic, ActorModule, CriticModule
n = 4096
env = VecRobotWalk(n, seed=1234); env.reset(); env.stagger_episodes()
pol = FusedActorCritic(371, 12, 256, 0)
torch.manual_seed(0)
pol.load_modules(ActorModule([371], [12], 256).cuda(), CriticModule([371], 256).cuda())
obs = torch.randn(n, 371, device="cuda")
act = torch.rand(n, 12, device="cuda") * 2 - 1
side = torch.cuda.Stream()
def run(mode, iters=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(iters):
        if mode == "step":
            env.step_autoreset(act)
        elif mode == "fwd":
            pol.forward(obs, seed=i)
        elif mode == "serial":
            pol.forward(obs, seed=i); env.step_autoreset(act)
        else:  # both, independent streams
            with torch.cuda.stream(side):
                pol.forward(obs, seed=i)
            env.step_autoreset(act)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
for m in ("step", "fwd", "serial", "both"):
    run(m, 50)
    print("%-7s %.4f ms per iteration" % (m, run(m)))
