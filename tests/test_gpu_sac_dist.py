"""N > 1 path of the device-side SAC update on one GPU box: two processes share cuda:0, each trains on half of the batch and
the gradients are averaged over gloo (host copies standing in for RCCL); the result must equal one process on the whole
batch (soft_actor_critic.cpp:93-170 with mean losses: equal shards -> rank average = global mean)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
B = 256


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(lo, hi):
    import torch
    sys.path.insert(0, ROOT)
    from evomotion_amd import VecSacAgent
    from evomotion_amd.ppo import PARAMS as PP, ACTOR
    from evomotion_amd.qnet import PARAMS
    n = hi - lo
    ag = VecSacAgent(31, [371], [12], batch_size=n, epoch=1, replay_buffer_size=4, train_every=2, n_envs=64, use_graph=False)
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    r = lambda *s: torch.rand(*s, device="cuda", generator=g)
    full = ((r(B, 371) * 2 - 1) * 1.5, r(B, 12) * 2 - 1, r(B) * 2 - 1, (r(B) < 0.1).float(), (r(B, 371) * 2 - 1) * 1.5)
    u_next, u_curr = r(B, 12), r(B, 12)
    for dst, src in zip(ag._batch, full):
        dst.copy_(src[lo:hi])
    for _ in range(2):
        ag._train_once_hip(u_next=u_next[lo:hi].contiguous(), u_curr=u_curr[lo:hi].contiguous())
    vecs = [ag.twinq.vector(PARAMS, i) for i in range(4)] + [ag._actor_tr.vector(PP, ACTOR), ag.entropy.log_alpha.detach().reshape(-1)]
    # the entropy parameter's Adam state has ONE owner (the device step) in single-process and data-parallel runs: two
    # updates = two steps, non-zero moments, and a save() / load() round trip carries them (they used to be lost)
    import tempfile
    st = ag.optimizer_state()
    assert st["entropy"]["step"] == 2 and float(st["entropy"]["state"].abs().min()) > 0.0, st["entropy"]
    with tempfile.TemporaryDirectory() as d:
        ag.save(d)
        ag2 = VecSacAgent(32, [371], [12], batch_size=n, epoch=1, replay_buffer_size=4, train_every=2, n_envs=64, use_graph=False)
        ag2.load(d)
        st2 = ag2.optimizer_state()
        assert st2["entropy"]["step"] == 2 and torch.allclose(st2["entropy"]["state"], st["entropy"]["state"], rtol=0, atol=0)
        for dst, src in zip(ag2._batch, full):
            dst.copy_(src[lo:hi])
        ag2._train_once_hip(u_next=u_next[lo:hi].contiguous(), u_curr=u_curr[lo:hi].contiguous())
        assert ag2.optimizer_state()["entropy"]["step"] == 3
    return torch.cat(vecs).cpu().numpy()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out.put((rank, _run(rank * B // world, (rank + 1) * B // world)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sac_hip_update_equals_single_process():
    import torch.multiprocessing as mp
    ref = _run(0, B)
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
    assert np.array_equal(got[0], got[1])  # the replicas stay identical
    d = np.abs(got[0] - ref)
    assert d.max() <= 2.1e-3 and (d > 5e-5).mean() < 1e-3, (d.max(), (d > 5e-5).mean())
