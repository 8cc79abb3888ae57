"""Experiment: 4096 envs as G independent sub-batches, each stepped on its own HIP stream (software pipelining of the
step's four kernels across sub-batches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from evomotion_amd import VecRobotWalk
N, K, W = 4096, 600, 100
for G in (1, 2):
    n = N // G
    envs = [VecRobotWalk(n, seed=1234 + g * n) for g in range(G)]
    for e in envs: e.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    bank = 64
    actions = torch.rand(bank, N, 12, device="cuda", generator=gen) * 2 - 1
    acts = [[actions[b, g * n:(g + 1) * n].contiguous() for g in range(G)] for b in range(bank)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    torch.cuda.synchronize()
    def run(k, off):
        for i in range(k):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    envs[g].step_autoreset(acts[(off + i) % bank][g])
    run(W, 0)
    torch.cuda.synchronize()
    for e in envs: e.clear_stats()
    t0 = time.perf_counter()
    run(K, W)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = sum(e.stats()["env_steps"] for e in envs)
    print("issue %.1f us per call;" % (1e6 * t_issue / K / G), end=" ")
    print("G=%d  %.4f ms per step of all %d envs   %.2f M physics steps/s   %.2f M env-steps/s" % (G, 1e3 * dt / K, N, N * K / dt / 1e6, steps / dt / 1e6), flush=True)
    del envs
