"""examples/train_main.cpp — the torch-free data-parallel PPO loop on the C ABI with <rccl/rccl.h> (one process per GPU, one
all-gather of the advantage statistics and ONE in-place all-reduce of the [actor | critic] gradient buffer per epoch, no host
read inside an iteration) — run here as a single rank: its communicator has one member, the collectives are real RCCL calls,
and the weights after a few iterations must equal, bit for bit, those of the same loop driven through the Python binding
(ppo.py: same C ABI, same kernels, the count of selected transitions on the device in both)."""
import json
import os
import subprocess

import numpy as np
import pytest

from test_gpu_cxx_host import make_params

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")
SKEL = os.path.join(ROOT, "evomotion_amd", "data", "robot_walk_spider.skel")


def test_cxx_rccl_training_equals_python_training_bit_for_bit(tmp_path):
    import torch
    from evomotion_amd import FusedActorCritic, VecRobotWalk
    from evomotion_amd.ppo import FusedPpoTrainer, PARAMS, ACTOR, CRITIC
    if not os.path.exists(os.path.join(BUILD, "train_main")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    n, T, iters, epoch, seed = 256, 8, 3, 2, 1234
    dump = str(tmp_path / "weights.bin")
    env_vars = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env_vars.pop(k, None)
    p = subprocess.run([os.path.join(BUILD, "train_main"), "--skeleton", SKEL, "--envs", str(n), "--horizon", str(T), "--iters", str(iters),
                        "--epoch", str(epoch), "--seed", str(seed), "--dump", dump], capture_output=True, timeout=600, env=env_vars)
    assert p.returncode == 0, p.stderr.decode()
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["world"] == 1 and line["collectives_per_iter"] == 1 + epoch and line["env_errors"] == [0, 0]
    assert np.isfinite(line["actor_loss"]) and np.isfinite(line["critic_loss"])
    got = np.fromfile(dump, np.float32)

    S, A = 371, 12
    env = VecRobotWalk(n, seed=seed, device=0)
    pol = FusedActorCritic(S, A, 256, 0)
    pa, pc = make_params(S, A, 256, True, 1000), make_params(S, A, 256, False, 500000)
    pol.set_weights(pa, pc)
    tr = FusedPpoTrainer(pol, T * n)
    assert line["allreduce_bytes_per_epoch"] == 4 * tr.grad_floats
    tr.set_flat(torch.from_numpy(pa).cuda(), torch.from_numpy(pc).cuda(), reset_optimizer=True)
    z = lambda *s, **k: torch.zeros(*s, device="cuda", **k)
    states = z(T + 1, n, S)
    actions, logp, values, rewards = z(T, n, A), z(T, n, A), z(T, n), z(T, n)
    done, valid = z(T, n, dtype=torch.uint8), z(T, n, dtype=torch.uint8)
    scratch = (z(n, A), z(n, A), z(n))
    env.reset()
    states[0].copy_(env.obs)
    for it in range(iters):
        if it:
            states[0].copy_(states[T])
        for t in range(T):
            pol.forward(states[t], seed=seed, out=(actions[t], logp[t], values[t]))
            env.step_autoreset(actions[t], reward_out=rewards[t], done_out=done[t], valid_out=valid[t], obs_out=states[t + 1])
        _, _, last_v = pol.forward(states[T], seed=seed, out=scratch)
        next_values = torch.cat([values[1:], last_v[None]])
        mask = (valid == 1).to(torch.uint8)
        la, lc = tr.train(states[:T], actions, rewards, done, logp, values, next_values.contiguous(), mask, 0.99, 0.95, 0.2, 0.01, 0.5, epoch, 1e-3, 0.5)
    want = torch.cat([tr.vector(PARAMS, ACTOR), tr.vector(PARAMS, CRITIC)]).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    assert abs(la - line["actor_loss"]) == 0.0 and abs(lc - line["critic_loss"]) == 0.0
    moved = np.abs(want - np.concatenate([pa, pc])).max()
    assert moved > 1e-4  # the update did something


def test_rendezvous_gives_up_with_a_message_and_never_takes_a_stale_id(tmp_path):
    """The ncclUniqueId travels through a file.  A rank > 0 whose rank 0 never comes must end with a message after the bounded
    wait (not block in ncclCommInitRank), and an id file left behind by an earlier run on the same port — older than this
    process — must not be accepted (ids of two different runs would block the communicator for ever)."""
    import time
    if not os.path.exists(os.path.join(BUILD, "train_main")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    idf = tmp_path / "evm_nccl_id.test"
    idf.write_bytes(b"\x07" * 128)                         # sizeof(ncclUniqueId) bytes of a previous run
    old = time.time() - 3600
    os.utime(idf, (old, old))
    env_vars = dict(os.environ, RANK="1", WORLD_SIZE="2", LOCAL_RANK="0", EVM_NCCL_ID_FILE=str(idf), EVM_RENDEZVOUS_TIMEOUT_S="2",
                    HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.time()
    p = subprocess.run([os.path.join(BUILD, "train_main"), "--skeleton", SKEL, "--envs", "64", "--horizon", "4", "--iters", "1"],
                       capture_output=True, timeout=120, env=env_vars)
    assert p.returncode != 0 and time.time() - t0 < 60
    assert b"no fresh ncclUniqueId" in p.stderr and b"after 2 s" in p.stderr, p.stderr
