#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec on robot_walk @4096 envs/GPU (BASELINE.json).

A "step" is one evm_env_step_autoreset call over the whole batch (every lane runs one stepSimulation);
`value` counts only do_step() transitions delivered to the agent — settle steps inside reset() and the
reset's own emitted step are NOT counted — divided by the wall time of the K timed calls (max over ranks).
Actions are pre-generated uniform [-1,1) tensors resident in HBM (config 2: random actions, dynamics only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic bytes of one env physics step, by collision mode (the line must describe the workload that ran):
#   0 (floor contacts only): SURVEY.md §8(d)'s 8 300 B — body state 2 132 R + 2 132 W, observation history 408 + 408, floor contact
#     cache 1 664, counters 16, action 48, observation 1 484, reward + done 8;
#   1 (member-vs-member contacts, the default): + the pair manifolds' cache: 2.75 live pair manifolds per env with about 1.5 points
#     each (profiles/r4*_pair_stats.txt) = 4.1 points x 13 f32, read and written: + 430 B = 8 730 B.
ALG_BYTES_PER_ENV_STEP = {0: 8300, 1: 8730}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak = fp32 matrix peak on this part (MI355X_MICROARCH.md)
# Algorithmic fp32 FLOP of one env physics step (fma = 2), from the row counts of DESIGN.md §2 — table in DESIGN.md §5:
# 10 sweeps x (12 hinges x (3 lin + 2 ang rows) + 4 fixed x (3 ang + 3 lin) + 12 sliders x (2 ang + 2 lin + motor)
# + 24 p2p x 3 rows + ~8 contact points x (normal + friction)) = 149.7 k, row set-up 23.6 k, bodies 8.3 k,
# hull scans 13.4 k, integration + observation 7.4 k
# member-vs-member mode adds: the pair points' rows (4.1 points x (two-body normal 102 + friction 102) x 10 sweeps = 8.4 k) and the
# narrowphase (per env-step 3.3 box-box queries x 8.4 GJK iterations x (2 x 8 vertices x 6 + about 300 for the simplex) = 11 k and
# 1.07 foot queries x 9.1 x (459 x 6 + 300) = 29.8 k; profiles/r4*_pair_stats.txt, r4*_kstamps.txt) = 251.6 kFLOP
ALG_FLOP_PER_ENV_STEP = {0: 202.4e3, 1: 251.6e3}
POLICY_FLOP_PER_ROW = 654848.0  # actor 333 312 + critic 321 536 GEMM FLOP per act
# untimed calls after stagger_episodes().  One episode + reset cycle is ~120 calls; 192 calls (rounds 2 and 3) left the driver's
# 20-step window inside the transient that follows the staggering (do_step_fraction 0.559 there).  A 20-step window is a sample of a
# population that keeps breathing: windows taken after 512 / 1024 / 2048 / 4096 / 8192 calls gave 8.38 / 7.95 / 8.24 / 7.84 / 7.82 M
# env-steps/s (do_step_fraction 0.621 / 0.611 / 0.619 / 0.605 / 0.596) around the 8.22 M (0.606) of a 1024-step run, which averages
# 51 such windows (gpurun_out/pr_*.json).  The pre-roll is the length whose window is closest to that average.
PREROLL_CALLS = int(os.environ.get("EVM_BENCH_PREROLL", "2048"))


# the PMC profile of THIS build's kernels, per collision mode (profiles/<tag>_traffic.json, written by tools/profile_round.sh +
# tools/traffic_json.py): named explicitly — the newest file by sort order need not be the current build's
TRAFFIC_PROFILE = {1: "r4z_traffic.json", 0: "r4z0_traffic.json"}


def policy_kernel_name(rows, nets, dev):
    """The form evm_policy_forward picks for this batch (policy_kernels.hip, policy_tile_rows): 16-row tiles while 32-row
    tiles give the device at most one workgroup per CU."""
    import torch
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    split = os.environ.get("EVM_POLICY_SPLIT", "1") != "0"
    wg32 = -(-rows // 32) * nets
    if (wg32 < cus) if split else (wg32 <= cus):
        return "k_policy_forward16 (16-row tiles, v_mfma_f32_16x16x4_f32)"
    if split:
        return "k_policy_forward<1> (32-row tiles, six v_mfma_f32_32x32x16_bf16 products per fp32 product)"
    return "k_policy_forward<0> (32-row tiles, v_mfma_f32_32x32x2_f32)"


def measured_traffic(n, self_collision=0):
    """HBM-side bytes per step of the dynamics pipeline, from the committed PMC profile of this build and workload
    (TRAFFIC_PROFILE; rocprofv3 cannot run inside the timed bench); None when the file is missing or the batch size or the
    collision mode differs from the profiled one.  Returns (bytes, file name)."""
    name = TRAFFIC_PROFILE.get(int(self_collision))
    if not name:
        return (None, None)
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            t = json.load(f)
        if t["envs_per_launch"] == n and int(t.get("self_collision", 0)) == int(self_collision):
            return (t["traffic_bytes_per_launch"], name)
    except (OSError, KeyError, ValueError):
        pass
    return (None, None)


def host_cores():
    """cores this process may really use: the affinity mask, cut down by the cgroup CPU quota of a container (a GPU box hands
    out 16 cores of a much larger host) and by 16"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_worker(seed, steps, self_collision=1):
    """child process of cpu_baseline()'s all-cores leg: one oracle env, prints `steps seconds`"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    L = orc.load()
    env = orc.OracleEnv(seed=seed, lib=L, self_collision=self_collision)
    n_res = ctypes.c_int()
    t = L.orc_bench_env_steps(env.h, steps, 11, ctypes.byref(n_res))
    print(steps, t)


def bullet_probe():
    """Is Bullet3 (the library the reference's step lives in, evo_motion_model/CMakeLists.txt:14) on THIS box?  Looks for the
    header and the library the reference links; nothing is built or downloaded.  Returns a short finding."""
    import glob
    hdr = [p for p in ("/usr/include/bullet/btBulletDynamicsCommon.h", "/usr/local/include/bullet/btBulletDynamicsCommon.h",
                       "/usr/include/btBulletDynamicsCommon.h") if os.path.exists(p)]
    libs = []
    for d in ("/usr/lib", "/usr/lib64", "/usr/lib/x86_64-linux-gnu", "/usr/local/lib"):
        libs += glob.glob(os.path.join(d, "libBulletDynamics*"))
    try:
        import importlib.util
        pyb = importlib.util.find_spec("pybullet") is not None
    except Exception:
        pyb = False
    harness = os.path.join(ROOT, "oracle", "_ref", "bullet_harness")
    return {"header": hdr[0] if hdr else None, "library": libs[0] if libs else None, "pybullet": pyb,
            "harness_built": os.path.exists(harness)}


def cpu_baseline(seconds=12.0, self_collision=1):
    """The scalar CPU restatement (oracle/) timed on this box's host cores: one env on one thread (the reference's loop is one
    env, one thread driving Bullet), and — so that the comparison is not against a single core of a many-core host — one
    independent env per core on all cores at once (plain child processes, bounded by a timeout)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    L = orc.load()
    env = orc.OracleEnv(seed=1234, lib=L, self_collision=self_collision)
    n_res = ctypes.c_int()
    probe = bullet_probe()
    found = probe["header"] and probe["library"]
    # calibrate, then one bounded sample
    t = L.orc_bench_env_steps(env.h, 2000, 7, ctypes.byref(n_res))
    steps = max(2000, int(2000 * seconds / max(t, 1e-6)))
    t = L.orc_bench_env_steps(env.h, steps, 11, ctypes.byref(n_res))
    out = {
        "value": steps / t, "unit": "env-steps/s", "cores": 1, "kind": "port",
        "sample": f"{steps} do_step calls incl. {n_res.value} reset() (60 settle steps each), 1 env, 1 thread, "
                  f"{t:.1f} s of the scalar CPU restatement (oracle/, {'member-vs-member' if self_collision else 'floor-only'} contacts); "
                  + ("Bullet3 found on this box (%s) but the harness oracle/bullet_harness.cpp was not built: timed the restatement" % probe["library"]
                     if found else "Bullet3 probed for on this box and not found (no btBulletDynamicsCommon.h / libBulletDynamics, no pybullet)"),
        "bullet_probe": probe,
    }
    if probe["harness_built"]:
        # the reference's own physics library under our harness (oracle/bullet_harness.cpp, built by oracle/Makefile where Bullet3
        # exists): one env, one thread, a bounded sample — then THIS is the baseline and the restatement's figure rides along
        try:
            import subprocess
            skel = os.path.join(ROOT, "evomotion_amd", "data", "robot_walk_spider.skel")
            o = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "bullet_harness"), skel, "--bench", str(seconds), "--self-collision", str(self_collision)],
                               capture_output=True, timeout=4 * seconds + 60)
            r = json.loads(o.stdout.decode().strip().splitlines()[-1])
            out["port"] = {"value": out["value"], "sample": out["sample"]}
            out.update(value=r["bullet_env_steps_per_s"], kind="reference",
                       sample="%.1f s of do_step calls (resets included) on Bullet3 itself through oracle/bullet_harness.cpp, 1 env, 1 thread" % r["seconds"])
        except Exception as e:
            out["bullet_harness_error"] = str(e)
    try:
        import subprocess
        cores = host_cores()
        if cores > 1:
            per = max(2000, int(steps * 6.0 / max(t, 1e-6)))  # about 6 s per process at the single-core rate
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(1234 + 17 * i), str(per), str(self_collision)],
                                      stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for i in range(cores)]
            res, deadline = [], time.time() + 60.0
            for p_ in procs:
                try:
                    o, _ = p_.communicate(timeout=max(deadline - time.time(), 0.1))
                    a, b = o.decode().split()
                    res.append((int(a), float(b)))
                except Exception:
                    p_.kill()
            if len(res) == cores:
                out["all_cores"] = {"value": sum(r[0] for r in res) / max(r[1] for r in res), "unit": "env-steps/s", "cores": cores,
                                    "sample": f"{cores} independent envs, one process per core, {per} do_step calls each"}
    except Exception as e:  # the single-core figure above stands on its own
        out["all_cores"] = {"error": str(e)}
    return out


def pose_parity(device, n_oracles=64, steps=256, self_collision=1):
    """Second half of BASELINE.json's metric: per-step pose L2 of the HIP path against the CPU restatement (oracle/; Bullet3 is
    not installed, so this is parity with the restatement, not with Bullet).  Teacher-forced: every step both sides start from
    the restatement's state, take the same action, and the body poses [41][pos xyz, quat xyzw], the reward and the done flag are
    compared — SURVEY section 8(d) config 2's sample: 64 environments x 256 steps with resets.  Part of the cpu_baseline leg (rank 0,
    N = 1), outside every timed region."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from evomotion_amd import VecRobotWalk
    L = orc.load()
    env = VecRobotWalk(64, seed=4321, device=device, parameters={"self_collision": self_collision})
    env.reset()
    oracles = [orc.OracleEnv(seed=4321 + i, lib=L, self_collision=self_collision) for i in range(n_oracles)]
    for o in oracles:
        o.reset()
    rng = np.random.default_rng(7)
    pos_l2, quat_l2 = [], []
    rew_err, done_mismatch = 0.0, 0
    for k in range(steps):
        blob = env.get_state()
        for i, o in enumerate(oracles):  # the restatement free-runs, the HIP path starts every step from its state
            blob[i] = o.get_state()
        env.set_state(blob)
        act = rng.uniform(-1, 1, (64, env.action_dim)).astype(np.float32)
        st = env.do_step(torch.from_numpy(act))
        poses = env.body_poses().cpu().numpy()
        rew_g, done_g = st.reward.cpu().numpy(), st.done.cpu().numpy()
        for i, o in enumerate(oracles):
            _, rew, done = o.do_step(act[i])
            rew_err = max(rew_err, abs(float(rew_g[i]) - float(rew)))
            done_mismatch += int(bool(done_g[i]) != bool(done))
            d = poses[i] - o.poses()
            pos_l2.append(float(np.sqrt((d[:, :3] ** 2).sum())))
            quat_l2.append(float(np.sqrt((d[:, 3:] ** 2).sum())))
            if done:
                o.reset()
    env.close()
    return {"pose_l2_mean": float(np.mean(pos_l2)), "pose_l2_max": float(np.max(pos_l2)), "unit": "m (L2 over the 41 body positions, per env step)",
            "quat_l2_max": float(np.max(quat_l2)), "reward_abs_max": rew_err, "done_mismatches": done_mismatch, "env_steps": len(pos_l2),
            "against": "CPU restatement (oracle/), one step from identical state (teacher-forced); Bullet3 absent: physics parity unpinned; "
                       "the restatement and the kernel share one stated substitution: the floor is a plane (north star), the reference's a 2000 m "
                       "convex hull through GJK - one-step difference measured in DESIGN.md section 2d (median 9e-7 m, p99 1e-3 m with a "
                       "well-conditioned floor box)"}


def launch_ranks(n_ranks, argv, worker=None, timeout=None, log_dir=None):
    """`bench.py --gpus N` without a torchrun environment: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, as torch.distributed.run would) and relay rank 0's JSON line.  The parent never
    imports torch or touches HIP, and the children are started with subprocess (never exec'd from a GPU process).
    Every rank's stderr (and the stdout of ranks > 0) is kept in a file of its own under log_dir (default: a fresh directory
    under $TMPDIR) and echoed when the run fails or times out, so that a dead rank says why.
    worker: command prefix of the rank program (tests pass a stub); default = this file.  Returns the exit code."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = list(worker) if worker else [sys.executable, os.path.abspath(__file__)]
    log_dir = log_dir or tempfile.mkdtemp(prefix="evm_bench_ranks_%d_" % port)
    os.makedirs(log_dir, exist_ok=True)
    procs, logs = [], []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        err = open(os.path.join(log_dir, "rank%d.err" % r), "wb")
        out = subprocess.PIPE if r == 0 else open(os.path.join(log_dir, "rank%d.out" % r), "wb")
        logs.append((err, out))
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=out, stderr=err))
    t0 = time.time()
    out0, rc, failed = None, 0, None
    try:
        pending = set(range(n_ranks))
        while pending:
            for r in sorted(pending):
                if r == 0 and out0 is None:
                    # rank 0 prints one line at the very end; communicate() also reaps it
                    try:
                        out0, _ = procs[0].communicate(timeout=0.2)
                    except subprocess.TimeoutExpired:
                        pass
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0 and not rc:
                        rc, failed = code, r
            if rc:
                break
            if out0 is None and 0 not in pending:
                # rank 0 ended between the timed-out communicate() above and the poll(): what it printed is still in the pipe
                out0, _ = procs[0].communicate()
            if timeout and time.time() - t0 > timeout:
                rc, failed = 124, None
                break
            time.sleep(0.05)
    finally:
        for p_ in procs:  # exact PIDs of our own children only
            if p_.poll() is None:
                p_.terminate()
        for p_ in procs:
            try:
                p_.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p_.kill()
        for err, out in logs:
            err.close()
            if out is not subprocess.PIPE:
                out.close()
    if rc:
        sys.stderr.write("bench.py: %s; no result line.  Per-rank logs in %s:\n"
                         % ("rank %d exited with code %d" % (failed, rc) if failed is not None else "timed out after %.0f s" % timeout, log_dir))
        for r in range(n_ranks):
            try:
                with open(os.path.join(log_dir, "rank%d.err" % r), "rb") as f:
                    tail = f.read()[-4000:].decode(errors="replace")
            except OSError:
                tail = ""
            sys.stderr.write("---- rank %d stderr (tail) ----\n%s\n" % (r, tail if tail.strip() else "(empty)"))
        return rc
    lines = [l for l in (out0 or b"").decode().splitlines() if l.strip()]
    if not lines:
        sys.stderr.write("bench.py: rank 0 printed nothing (logs in %s)\n" % log_dir)
        return 1
    sys.stdout.write(lines[-1] + "\n")
    return 0


def main():
    if len(sys.argv) in (4, 5) and sys.argv[1] == "--cpu-worker":  # child of cpu_baseline(): no torch, no GPU
        cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) == 5 else 1)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--self-collision", type=int, choices=[0, 1], default=1,
                    help="1 (default) = member-vs-member contacts as in the reference (every pair of members except constraint "
                         "parent / child collides); 0 = floor contacts only (the north-star's plane-contact configuration)")
    ap.add_argument("--mode", choices=["dynamics", "ppo", "sac"], default="dynamics",
                    help="dynamics = BASELINE configs[1] (random actions, headline); ppo = configs[2]/[3]: fused MFMA "
                         "actor-critic forward inside the rollout, PPO update every --horizon steps; sac = configs[4]: fused "
                         "actor forward + device replay ring (replay_buffer_size 1024 slots), SAC update every --train-every steps")
    ap.add_argument("--train-every", type=int, default=4)
    ap.add_argument("--sac-batch", type=int, default=4096)
    ap.add_argument("--replay-slots", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the N>1 path with several ranks on ONE GPU)")
    ap.add_argument("--no-update", action="store_true", help="ppo mode: rollout only")
    ap.add_argument("--no-stagger", action="store_true", help="skip the episode desynchronisation + pre-roll (all envs then "
                    "leave reset() in lock step and a short run measures physics steps without any reset)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: WORLD_SIZE=%d overrides --gpus %d\n" % (world, args.gpus))
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    local_rank = dev_index

    from evomotion_amd import FusedActorCritic, RandomAgent, VecRobotWalk

    # Collective sanity BEFORE anything is timed: every rank takes part in a real all_reduce of ones and the sum must be the
    # world size — over the NCCL (= RCCL) backend this is `rccl_ranks`; a rehearsal over gloo reports `collective_ranks` only.
    rccl_ranks = collective_ranks = None
    if world > 1:
        ones = torch.ones(1, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        collective_ranks = int(ones.item())
        if collective_ranks != world:
            raise RuntimeError("bench.py: all_reduce of ones over %s gave %d, expected the world size %d" % (args.backend, collective_ranks, world))
        if args.backend == "nccl":
            rccl_ranks = collective_ranks

    n = args.envs
    env = VecRobotWalk(n, seed=1234 + rank * n, device=local_rank, parameters={"self_collision": args.self_collision})
    st0 = env.reset()
    # RandomAgent (debug_agents.cpp:28-30) fills an action bank resident in HBM, cycled by the rollout loop: config 2 times
    # the dynamics, not the generator
    random_agent = RandomAgent([env.action_dim], dev, seed=1234 + rank)
    bank = 64
    actions = torch.stack([random_agent.act(st0.state) for _ in range(bank)]).contiguous()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    agent = None
    if args.mode == "ppo":
        from evomotion_amd import VecPpoGaeAgent
        agent = VecPpoGaeAgent(1234, [env.state_dim], [env.action_dim], hidden_size=256, device=local_rank,
                               horizon=args.horizon, epoch=8, learning_rate=1e-3)

    sac = None
    if args.mode == "sac":
        from evomotion_amd import VecSacAgent
        sac = VecSacAgent(1234, [env.state_dim], [env.action_dim], batch_size=args.sac_batch, epoch=1, learning_rate=1e-3,
                          replay_buffer_size=args.replay_slots, train_every=args.train_every, n_envs=n, device=local_rank)

    def run(k_steps, offset=0):
        if sac is not None:
            for i in range(k_steps):
                sac.step(env, train=not args.no_update)
            return
        if agent is None:
            for i in range(k_steps):
                env.step_autoreset(actions[(offset + i) % bank])
            return
        done_steps = 0
        while done_steps < k_steps:  # whole horizons only: K is rounded up to a multiple of --horizon
            agent.rollout(env)
            if not args.no_update:
                agent.update()
            done_steps += args.horizon

    if not args.no_stagger:
        # episodes start in lock step (every env left reset() together): spread their phases over one episode + reset cycle
        # and roll past the staggered first episodes, untimed and independent of --warmup, so that the timed region sees
        # the steady-state mix of stepping and settling envs
        env.stagger_episodes()
        for i in range(PREROLL_CALLS):
            env.step_autoreset(actions[i % bank])
    run(args.warmup)
    barrier()
    env.clear_stats()
    env.timing_begin()
    if agent is not None:
        agent.fused.timing_begin()
    if sac is not None:
        sac.fused.timing_begin()
        sac.replay.timing_begin()
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import gc
    gc.collect()
    gc.disable()  # no collector pause between the two clock reads (the driver's 20-step region is 6 ms)
    t0 = time.perf_counter()
    ev_begin.record()  # on the stream every kernel of the timed region is launched on (torch's current stream)
    run(args.steps, args.warmup)
    ev_end.record()
    barrier()   # synchronize + (N > 1) barrier + synchronize: the K steps of every rank are done
    t1 = time.perf_counter()
    gc.enable()
    # the event read-outs come AFTER the second clock read: with them inside, a 20-step region once measured 8.5 ms of wall
    # time around 6.1 ms of kernels (gpurun_out/r3n_bench20.json of that run: ms_per_step 0.423 against launch_ms 0.307)
    ms_kernel, n_launch, ms_sweeps = env.timing_end_detail()
    region_ms = ev_begin.elapsed_time(ev_end)
    ms_policy, n_policy = agent.fused.timing_end() if agent is not None else (sac.fused.timing_end() if sac is not None else (0.0, 0))
    rp = sac.replay.timing_end() if sac is not None else None
    ppo_ms, ppo_epochs = 0.0, 0
    if agent is not None and not args.no_update:
        # one more update outside the timed region, with HIP events around each epoch (the events synchronise)
        agent._trainer.timing(True)
        agent.update()
        ppo_ms, ppo_epochs = agent._trainer.timing(False)
        torch.cuda.synchronize()
    if agent is not None:
        args.steps = n_launch
    # MFMA utilisation of the policy GEMM in every mode: in dynamics mode the fused actor-critic forward is timed here, outside
    # the headline region, on the envs' current observations (50 launches, random-init weights of init.cpp:7-21)
    if agent is None and sac is None and rank == 0:
        from evomotion_amd import ActorModule, CriticModule
        torch.manual_seed(1234)
        pol = FusedActorCritic(env.state_dim, env.action_dim, 256, local_rank)
        pol.load_modules(ActorModule([env.state_dim], [env.action_dim], 256).to(dev), CriticModule([env.state_dim], 256).to(dev))
        # steady state, as inside a rollout where the GPU never idles: the set-up above (module init, weight repack) leaves a
        # gap behind the timed region in which the clocks fall back; 50 untimed launches bring them up again
        for _ in range(50):
            pol.forward(env.obs)
        torch.cuda.synchronize()
        # the kernel's average launch duration back to back: ONE event pair around the 200 launches on their stream (an event pair
        # around every launch adds its own record / wait latency, ~2 us, to each: 28.7 us kernels then read 30.7-31.7 us)
        pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pe0.record()
        for i in range(200):
            pol.forward(env.obs, seed=i)
        pe1.record()
        torch.cuda.synchronize()
        ms_policy, n_policy = pe0.elapsed_time(pe1), 200
    elapsed = t1 - t0
    st = env.stats()
    tt = torch.tensor([elapsed, float(st["env_steps"]), float(st["resets"])], dtype=torch.float64,
                      device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
    env_steps, resets = float(tt[1].item()), float(tt[2].item())
    if rank == 0:
        # ONE denominator for `value` and `roofline.achieved`: the timed region, by HIP events on the launch stream around all of
        # its K steps (agrees with the wall clock between the barriers to the launch latency of the first kernel)
        sampled_ms = ms_kernel / max(n_launch, 1)   # mean of the individually bracketed steps (every 4th)
        launch_ms = region_ms / max(args.steps if agent is None and sac is None else n_launch, 1) if agent is None and sac is None else sampled_ms
        phys_per_launch = n  # every lane runs one stepSimulation per launch
        alg_bytes, alg_flop = ALG_BYTES_PER_ENV_STEP[int(args.self_collision)], ALG_FLOP_PER_ENV_STEP[int(args.self_collision)]
        achieved = alg_bytes * phys_per_launch / (launch_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec on robot_walk @4096 envs/GPU",
            "value": env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("robot_walk, %d envs/GPU on %d MI355X, SAC: fused MFMA actor forward, device replay ring of %d "
                             "slots, %s (configs[4])" % (n, world, args.replay_slots, "rollout + ring only" if args.no_update else
                              "SAC update (batch %d) every %d steps on the device (twin Q, targets, actor and entropy steps, fp32 MFMA), "
                              "one HIP graph" % (args.sac_batch, args.train_every))) if sac is not None else
                            ("robot_walk, %d envs/GPU on %d MI355X, HIP dynamics only, uniform random actions, "
                             "rollout form with in-band reset, %s (configs[1])" % (n, world, "member-vs-member contacts as in the reference "
                              "(self_collision=1)" if args.self_collision else "floor contacts only (self_collision=0)")) if agent is None else
                            ("robot_walk, %d envs/GPU on %d MI355X, PPO hidden_size=256, fused MFMA actor-critic forward "
                             "in the rollout, horizon %d, %s (configs[2])" % (n, world, args.horizon,
                              "rollout only" if args.no_update else
                              "HIP PPO update (fp32 MFMA forward / backward / weight gradients, epoch 8) every horizon")),
                "envs_per_gpu": n,
                "self_collision": args.self_collision,
                "physics_steps_per_s": world * n * args.steps / elapsed,
                "do_step_fraction": env_steps / (world * n * args.steps),
                "resets_started": resets,
            },
            "rccl_ranks": rccl_ranks,
            "collective_ranks": collective_ranks,
            "backend": args.backend if world > 1 else None,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(n, args.self_collision)[0], "traffic_source": measured_traffic(n, args.self_collision)[1],
                "kernel": ("k_sweeps_g (dominant: the Gauss-Seidel sweeps) + k_split_pre_a / %s: one step" % ("pairs_rec" if args.self_collision else "pre_b") if ms_sweeps > 0 else "k_env_step<7>"),
                "launch_ms": launch_ms, "sampled_step_ms": sampled_ms, "dominant_kernel_ms": (ms_sweeps / max(n_launch, 1)) if ms_sweeps > 0 else launch_ms,
                "alg_bytes_per_env_step": alg_bytes,
                "note": "algorithmic %d B per env physics step (by collision mode, bench.py) x %d envs per step; a step is a pipeline of three kernels (setup, records [+ narrowphase with member-vs-member contacts], sweeps + integration) up to "
                        "8192 envs (launch_ms = all of them, HIP events on the launch stream; dominant_kernel_ms = the Gauss-Seidel "
                        "sweeps kernel alone), one monolithic kernel above; fp32-VALU/latency bound (about 50 FLOP per algorithmic "
                        "byte); traffic = memory-side bytes per step from the committed rocprofv3 PMC passes "
                        "(traffic_source under profiles/), see DESIGN.md" % (alg_bytes, n),
            },
        }
        valu_tf = alg_flop * phys_per_launch / (launch_ms * 1e-3) / 1e12
        out["roofline_valu"] = {"bound": "valu", "achieved": valu_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": valu_tf / VALU_PEAK_TFLOPS, "traffic": None, "launch_ms": launch_ms,
                                "flop_per_env_step": alg_flop,
                                "note": "algorithmic fp32 FLOP of one stepSimulation (row counts of DESIGN.md, fma = 2) x envs per "
                                        "step / the same step time as `roofline`; peak = packed-fp32 VALU peak"}
        if agent is None and sac is None and n_policy:
            pol_ms = ms_policy / n_policy
            tf = POLICY_FLOP_PER_ROW * n / (pol_ms * 1e-3) / 1e12
            out["roofline_policy"] = {"bound": "mfma", "achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tf / VALU_PEAK_TFLOPS, "traffic": None, "kernel": policy_kernel_name(n, 2, dev),
                                      "launch_ms": pol_ms, "rows": n,
                                      "note": "200 back-to-back launches after the timed region (not part of `value`), one HIP event pair around them, 50 untimed ones before; fp32 in and out, dense fp32 "
                                              "matrix peak; 654 848 GEMM FLOP per row"}
        if sac is not None and n_policy:
            pol_ms = ms_policy / n_policy
            tf = 333312.0 * n / (pol_ms * 1e-3) / 1e12  # actor 333 312 GEMM FLOP per act
            out["roofline_policy"] = {"bound": "mfma", "achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tf / VALU_PEAK_TFLOPS, "traffic": None,
                                      "kernel": policy_kernel_name(n, 1, dev) + ", actor only",
                                      "launch_ms": pol_ms, "note": "fp32-input MFMA, dense fp32 matrix peak"}
            # replay ring: pure byte movement.  push = one rollout step of all envs read + written once; sample = per drawn
            # row state + next state + action + reward + done read and written once
            S_, A_ = env.state_dim, env.action_dim
            push_bytes = 2.0 * n * (2 * S_ + A_ + 2) * 4 + n * 2
            out["roofline_replay_push"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "traffic": None,
                                           "launch_ms": rp["ms_push"] / max(rp["n_push"], 1),
                                           "achieved": push_bytes / (rp["ms_push"] / max(rp["n_push"], 1) * 1e-3) / 1e9,
                                           "kernel": "k_replay_copy (copies + the index block)"}
            out["roofline_replay_push"]["frac"] = out["roofline_replay_push"]["achieved"] / HBM_PEAK_GBS
            if rp["n_sample"]:
                smp_bytes = 2.0 * args.sac_batch * (2 * S_ + A_ + 2) * 4
                ms = rp["ms_sample"] / rp["n_sample"]
                out["roofline_replay_sample"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "traffic": None, "launch_ms": ms,
                                                 "achieved": smp_bytes / (ms * 1e-3) / 1e9, "frac": smp_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "kernel": "k_replay_plan + k_replay_gather"}
        if agent is not None and n_policy:
            pol_ms = ms_policy / n_policy
            tf = POLICY_FLOP_PER_ROW * n / (pol_ms * 1e-3) / 1e12
            out["roofline_policy"] = {"bound": "mfma", "achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tf / VALU_PEAK_TFLOPS, "traffic": None, "kernel": policy_kernel_name(n, 2, dev),
                                      "launch_ms": pol_ms, "note": "fp32-input MFMA, dense fp32 matrix peak"}
        if agent is not None and ppo_epochs:
            ep_ms = ppo_ms / ppo_epochs
            # the rows the epoch's kernels really worked on: the selected transitions of the rollout (FusedPpoTrainer.train keeps
            # only those), not horizon x envs
            rows = int(agent._trainer.last_rows) or n * args.horizon
            # GEMM FLOP of one epoch, both networks: forward 654 848 per row + backward: the Linear(256,256)
            # dgrad and the three weight-gradient GEMMs per network (no dgrad into the observations)
            S_, A_ = env.state_dim, env.action_dim
            bwd = 2.0 * (2 * 256 * 256 + 256 * 256 + S_ * 256) * 2 + 2.0 * (2 * (2 * A_) * 256 + 2 * 256)
            tf = (POLICY_FLOP_PER_ROW + bwd) * rows / (ep_ms * 1e-3) / 1e12
            out["roofline_ppo_update"] = {"bound": "mfma", "achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": tf / VALU_PEAK_TFLOPS, "traffic": None, "launch_ms": ep_ms, "rows": rows,
                                          "rollout_rows": n * args.horizon,
                                          "kernel": "k_ppo_forward + k_ppo_loss_* + k_ppo_backward + k_ppo_wgrad (x6) + reductions + k_ppo_adam: one epoch",
                                          "note": "HIP events around evm_ppo_grads .. evm_ppo_apply; fp32-input MFMA, dense fp32 matrix peak"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(self_collision=args.self_collision)
            try:
                out["pose_parity"] = pose_parity(local_rank, self_collision=args.self_collision)
            except Exception as e:  # reported, never fatal for the throughput line
                out["pose_parity"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
