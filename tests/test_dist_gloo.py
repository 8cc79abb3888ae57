"""N>1 path on CPU: two gloo ranks exercise the only cross-rank exchanges of the design — advantage statistics
(all_gather of (n, mean, M2) + Chan merge) and the gradient all-reduce — against a single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch_ref  # noqa: E402  (autograd restatements: the comparison side)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch(seed, B, T, S=24, A=3):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g) * 2 - 1
    lens = torch.randint(2, T + 1, (B,), generator=g)
    t = torch.arange(T).view(1, T, 1)
    valid = (t < lens.view(B, 1, 1)).float()
    done = torch.where(t >= (lens.view(B, 1, 1) - 1), torch.ones(()), torch.zeros(()))
    return dict(states=r(B, T, S) * valid, actions=r(B, T, A) * 0.9 * valid, rewards=r(B, T, 1) * valid, done=done,
                log_prob=(r(B, T, A) - 1.0) * valid, curr_values=r(B, T, 1) * valid, next_values=r(B, T, 1) * valid)


def _nets(agent):
    torch.manual_seed(7)
    a, c = agent.ActorModule([24], [3], 16), agent.CriticModule([24], 16)
    return a, c, torch.optim.Adam(a.parameters(), lr=1e-3), torch.optim.Adam(c.parameters(), lr=1e-3)


HP = dict(gamma=0.99, lam=0.95, epsilon=0.2, entropy_factor=0.01, critic_loss_factor=0.5, epoch=2, clip_grad_norm=0.5)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from evomotion_amd import agent
    full = _make_batch(3, 8, 6)
    shard = {k: v[rank * 4:(rank + 1) * 4] for k, v in full.items()}  # envs are sharded by rank, nothing else
    a, c, oa, oc = _nets(agent)
    torch_ref.ppo_train(a, c, oa, oc, **shard, **HP)
    flat = torch.cat([p.detach().reshape(-1) for p in list(a.parameters()) + list(c.parameters())])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put([g.numpy() for g in gathered])
    dist.destroy_process_group()


def test_two_rank_update_equals_single_process(hip_lib):
    from evomotion_amd import agent
    torch.set_num_threads(1)
    full = _make_batch(3, 8, 6)
    a, c, oa, oc = _nets(agent)
    torch_ref.ppo_train(a, c, oa, oc, **full, **HP)
    ref = torch.cat([p.detach().reshape(-1) for p in list(a.parameters()) + list(c.parameters())]).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1])  # replicas stay identical
    # the sharded update is the single-process update (Adam turns 1e-7 gradient noise into <= lr-sized steps)
    assert np.abs(got[0] - ref).max() < 2.5e-3 and np.abs(got[0] - ref).mean() < 2e-5


# ---- SAC: gradient averaging over equal shards --------------------------------------------------------------
def _sac_nets():
    from evomotion_amd import agent, sac
    torch.manual_seed(9)
    a = agent.ActorModule([24], [3], 16)
    qs = [sac.QNetworkModule([24], [3], 16) for _ in range(4)]
    e = sac.EntropyParameter(1.0, 1)
    opts = [torch.optim.Adam(m.parameters(), lr=1e-3) for m in (a, qs[0], qs[1], e)]
    return a, qs, e, opts


def _sac_batch():
    g = torch.Generator().manual_seed(4)
    r = lambda *s: torch.rand(*s, generator=g) * 2 - 1
    return dict(states=r(16, 24), actions=r(16, 3) * 0.9, rewards=r(16, 1), done=(torch.rand(16, 1, generator=g) < 0.2).float(),
                next_states=r(16, 24), u_next=torch.rand(16, 3, generator=g), u_curr=torch.rand(16, 3, generator=g))


def _sac_flat(a, qs, e):
    return torch.cat([p.detach().reshape(-1) for m in [a] + qs + [e] for p in m.parameters()])


def _sac_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from evomotion_amd import agent, sac
    full = _sac_batch()
    shard = {k: v[rank * 8:(rank + 1) * 8] for k, v in full.items()}
    a, qs, e, opts = _sac_nets()
    torch_ref.sac_train(a, qs[0], qs[1], qs[2], qs[3], e, *opts, shard["states"], shard["actions"], shard["rewards"], shard["done"],
                  shard["next_states"], 0.99, 0.005, -3.0, u_next=shard["u_next"], u_curr=shard["u_curr"],
                  grad_hook=torch_ref._all_reduce_grads_mean)
    flat = _sac_flat(a, qs, e)
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put([g.numpy() for g in gathered])
    dist.destroy_process_group()


def test_two_rank_sac_update_equals_single_process(hip_lib):
    from evomotion_amd import sac
    torch.set_num_threads(1)
    full = _sac_batch()
    a, qs, e, opts = _sac_nets()
    torch_ref.sac_train(a, qs[0], qs[1], qs[2], qs[3], e, *opts, full["states"], full["actions"], full["rewards"], full["done"],
                  full["next_states"], 0.99, 0.005, -3.0, u_next=full["u_next"], u_curr=full["u_curr"])
    ref = _sac_flat(a, qs, e).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sac_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1])  # replicas stay identical
    # first Adam step: |update| = lr regardless of the gradient scale, so rounding noise in the averaged gradient can
    # only matter where a gradient is ~0; the sharded update is the single-process update to well below lr
    assert np.abs(got[0] - ref).max() < 2e-4
