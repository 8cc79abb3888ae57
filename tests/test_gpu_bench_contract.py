"""The driver's contract with bench.py, checked on the GPU box: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON
line with the agreed keys, the workload BASELINE.json names, a roofline object for the dominant kernel and the CPU baseline
(oracle timed on the host cores) with the pose parity beside it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=300):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # exactly one line on stdout
    return json.loads(lines[0])


def test_default_form_line():
    d = _run("--gpus", "1", "--steps", "20", "--warmup", "5")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].startswith("env-steps/sec on robot_walk") and "4096" in d["metric"] and base["metric"].startswith("env-steps/sec on robot_walk")
    assert d["unit"] == "env-steps/s" and d["value"] > 1e6          # north_star: >= 1 M env-steps/s on one MI355X
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert abs(d["value"] - d["config"]["envs_per_gpu"] * d["config"]["do_step_fraction"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and (r["traffic"] is None or r["traffic"] > 0)
    for k in ("roofline_valu", "roofline_policy"):
        assert 0 < d[k]["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "env-steps/s" and c["sample"]
    assert c["all_cores"]["cores"] >= 1 and c["all_cores"]["value"] >= 0.5 * c["value"]
    p = d["pose_parity"]
    assert p["pose_l2_max"] < 5e-6 and p["env_steps"] == 64 * 256 and "unpinned" in p["against"]      # SURVEY 8(d) config 2's sample
    assert p["reward_abs_max"] < 1e-4 and p["done_mismatches"] == 0


def test_training_mode_lines_without_the_cpu_leg():
    d = _run("--mode", "ppo", "--steps", "32", "--warmup", "32", "--no-cpu-baseline")
    assert "cpu_baseline" not in d and d["roofline_ppo_update"]["bound"] == "mfma" and 0.2 < d["roofline_ppo_update"]["frac"] < 1
    d = _run("--mode", "sac", "--steps", "32", "--warmup", "32", "--no-cpu-baseline")
    assert d["value"] > 1e6 and "sac" in d["config"]["workload"].lower()


def test_two_ranks_on_one_gpu_end_to_end():
    """The N > 1 form of the driver's call, rehearsed on the one GPU of this box: `bench.py --gpus 2` starts
    two rank processes itself (launcher, RANK / WORLD_SIZE / MASTER_* environment), each creates its 4096 envs and its agent,
    the all-reduce-of-ones check passes before anything is timed, the PPO update exchanges its advantage statistics (all-gather)
    and gradients (all-reduce) over gloo — RCCL refuses two ranks on one device — and rank 0 prints the one line, `n_gpus: 2`.
    The 8-GPU run differs in the backend only."""
    d = _run("--gpus", "2", "--backend", "gloo", "--mode", "ppo", "--steps", "32", "--warmup", "32", "--no-cpu-baseline", timeout=600)
    assert d["n_gpus"] == 2 and d["collective_ranks"] == 2 and d["backend"] == "gloo" and d["rccl_ranks"] is None
    assert d["steps"] == 32 and d["scaling"] == "weak"
    # whole-job aggregate: both ranks' transitions over the max of their times
    assert d["value"] > 5e5 and 0 < d["config"]["do_step_fraction"] <= 1.0
    assert abs(d["value"] - 2 * d["config"]["envs_per_gpu"] * d["config"]["do_step_fraction"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert d["roofline_ppo_update"]["frac"] > 0.05
