"""N > 1 path of the HIP PPO update on one GPU box: two processes share cuda:0 and exchange the advantage statistics and
the gradients over gloo (the RCCL calls of a real multi-GPU run, routed through host copies); rank r trains on the envs
[r N/2, (r + 1) N/2) and the resulting weights must equal those of one process training on all N."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

HP = dict(gamma=0.99, lam=0.95, epsilon=0.2, entropy_factor=0.01, critic_loss_factor=0.5, clip_grad_norm=0.5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_ppo as T
    from evomotion_amd import agent
    actor, critic = T._modules(seed=5)
    Tn, N = 6, 256
    states, actions, rewards, done, values, next_values, mask = T._rollout(Tn, N, seed=13)
    g = torch.Generator(device="cuda"); g.manual_seed(17)
    with torch.no_grad():
        mu, sigma = actor(states.reshape(Tn * N, 371))
        logp = (agent.truncated_normal_log_pdf(actions.reshape(Tn * N, 12), mu, sigma)
                + 0.2 * (torch.rand(Tn * N, 12, device="cuda", generator=g) * 2 - 1)).reshape(Tn, N, 12)
    return T, actor, critic, (states, actions, rewards, done, logp, values, next_values, mask)


def _train(T, actor, critic, batch, lo, hi):
    import torch
    f, tr = T._trainer(actor, critic, batch[0].shape[0] * (hi - lo))
    sh = [t[:, lo:hi].contiguous() for t in batch]
    tr.train(*sh, epoch=2, learning_rate=1e-3, **HP)
    from evomotion_amd.ppo import PARAMS
    return torch.cat([tr.vector(PARAMS, 0), tr.vector(PARAMS, 1)]).cpu().numpy()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, actor, critic, batch = _setup()
    N = batch[0].shape[1]
    theta = _train(T, actor, critic, batch, rank * N // world, (rank + 1) * N // world)
    out.put((rank, theta))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hip_update_equals_single_process():
    import torch.multiprocessing as mp
    T, actor, critic, batch = _setup()
    ref = _train(T, actor, critic, batch, 0, batch[0].shape[1])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the replicas stay identical (same all-reduced gradients, same Adam state)
    assert np.array_equal(got[0], got[1])
    # and equal the single-process update up to the summation order of the gradients (two partial sums instead of one)
    d = np.abs(got[0] - ref)
    assert d.max() <= 2.1e-3 and (d > 2e-5).mean() < 5e-4, (d.max(), (d > 2e-5).mean())
