"""The compiled C++ side of the boundary (examples/, no torch, no Python): the Environment-shaped adapter driven like
src/train.cpp:41-66, and the torch-free rollout program, whose observations must equal — bit for bit — those of the same
rollout driven through the Python binding (same C ABI, same kernels, same seeds)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")
SKEL = os.path.join(ROOT, "evomotion_amd", "data", "robot_walk_spider.skel")


@pytest.fixture(scope="module")
def binaries():
    if not (os.path.exists(os.path.join(BUILD, "rollout_main")) and os.path.exists(os.path.join(BUILD, "adapter_check"))):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    return BUILD


def pattern(k):
    h = (k.astype(np.uint64) * 2654435761 + 12345) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0x5bd1e995) & 0xFFFFFFFF
    h ^= h >> 15
    return (h >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0) * np.float32(2.0) - np.float32(1.0)


def make_params(S, A, H, actor, base):
    """examples/rollout_main.cpp::make_params"""
    out, k = [], base

    def linear(o, i):
        nonlocal k
        s = np.float32(1.0) / np.sqrt(np.float32(i))
        out.append(pattern(np.arange(k, k + o * i, dtype=np.uint64)) * s)
        out.append(np.zeros(o, np.float32))
        k += o * i

    def ln(n):
        out.append(np.ones(n, np.float32)); out.append(np.zeros(n, np.float32))
    linear(H, S); ln(H); linear(H, H); ln(H)
    if actor:
        linear(A, H); linear(A, H)
    else:
        linear(1, H)
    return np.concatenate(out).astype(np.float32)


def test_adapter_runs_the_train_loop(binaries):
    p = subprocess.run([os.path.join(binaries, "adapter_check"), SKEL], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    assert b"adapter_check ok" in p.stdout


@pytest.mark.parametrize("mode", ["policy", "random"])
def test_cxx_rollout_equals_python_rollout_bit_for_bit(binaries, tmp_path, mode):
    import torch
    from evomotion_amd import FusedActorCritic, VecRobotWalk
    n, steps, seed, k = 64, 8, 1234, 8
    dump = str(tmp_path / "obs.bin")
    p = subprocess.run([os.path.join(binaries, "rollout_main"), "--skeleton", SKEL, "--envs", str(n), "--steps", str(steps), "--warmup", "0",
                        "--seed", str(seed), "--mode", mode, "--dump", dump, "--dump-envs", str(k)], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["envs"] == n and line["steps"] == steps and line["env_steps_per_s"] > 0
    got = np.fromfile(dump, np.float32).reshape(steps + 1, k, 371)
    # the same rollout through the Python binding
    env = VecRobotWalk(n, seed=seed, device=0)
    want = [env.reset().state[:k].cpu().numpy().copy()]
    if mode == "policy":
        pol = FusedActorCritic(371, 12, 256, 0)
        pol.set_weights(make_params(371, 12, 256, True, 1000), make_params(371, 12, 256, False, 500000))
    for call in range(steps):
        if mode == "policy":
            action, _, _ = pol.forward(env.obs, seed=seed)
        else:
            i = np.arange(n * 12, dtype=np.uint64)
            action = torch.from_numpy(pattern_call(i, call).reshape(n, 12)).cuda()
        want.append(env.step_autoreset(action).state[:k].cpu().numpy().copy())
    want = np.stack(want)
    assert np.isfinite(got).all()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), float(np.abs(got - want).max())


def pattern_call(i, call):
    """examples/rollout_main.cpp::k_uniform_actions"""
    h = (((i + np.uint64(call) * np.uint64(0x9E3779B9)) & 0xFFFFFFFF) * 2654435761 + 12345) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0x5bd1e995) & 0xFFFFFFFF
    h ^= h >> 15
    return (h >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0) * np.float32(2.0) - np.float32(1.0)


def test_cxx_rollout_throughput_line(binaries):
    """the headline configuration from C++ alone: 4096 envs, fused policy forward + dynamics, no Python in the loop"""
    p = subprocess.run([os.path.join(binaries, "rollout_main"), "--skeleton", SKEL, "--envs", "4096", "--steps", "256", "--warmup", "64"],
                       capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    print(line)
    assert line["physics_steps_per_s"] > 1.0e6  # north_star's floor, by a wide margin
