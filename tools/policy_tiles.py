"""Time evm_policy_forward at 4096 rows in its two tile forms (32 rows on v_mfma_f32_32x32x2_f32, 16 rows on
v_mfma_f32_16x16x4_f32), both networks and actor only.  HIP events around 200 launches each; prints one JSON line.
Usage (GPU box): python tools/policy_tiles.py [rows ...]            inputs N(0,1)
                 python tools/policy_tiles.py env [rows ...]        inputs = observations of a VecRobotWalk after reset() + 64 random steps
                 python tools/policy_tiles.py one 16|32 both|actor [rows]   200 launches of that one form (for a rocprofv3 --pmc pass)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evomotion_amd import ActorModule, CriticModule, FusedActorCritic  # noqa: E402

FLOP_BOTH = 654848.0  # GEMM FLOP per row, actor + critic (bench.py)


def main():
    args = sys.argv[1:]
    if args and args[0] == "one":
        tile, ao, n = int(args[1]), args[2] == "actor", int(args[3]) if len(args) > 3 else 4096
        actor, critic = ActorModule([371], [12], 256).cuda(), CriticModule([371], 256).cuda()
        f = FusedActorCritic(371, 12, 256, 0)
        f.load_modules(actor, critic)
        f.set_tile_rows(tile)
        x = torch.randn(n, 371, device="cuda")
        for _ in range(200):
            f.forward(x, actor_only=ao)
        torch.cuda.synchronize()
        return
    global ENV_OBS
    ENV_OBS = bool(args) and args[0] == "env"
    if ENV_OBS:
        args = args[1:]
    for n in ([int(a) for a in args] or [4096]):
        one(n)


ENV_OBS = False


def inputs(n):
    if not ENV_OBS:
        return torch.randn(n, 371, device="cuda")
    from evomotion_amd import VecRobotWalk
    env = VecRobotWalk(n, seed=3)
    st = env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for _ in range(64):
        st = env.step_autoreset(torch.rand(n, 12, device="cuda", generator=g) * 2 - 1)
    x = st.state.clone()
    env.close()
    return x


def one(n, reps=7, launches=100):
    actor, critic = ActorModule([371], [12], 256).cuda(), CriticModule([371], 256).cuda()
    f = FusedActorCritic(371, 12, 256, 0)
    f.load_modules(actor, critic)
    x = inputs(n)
    flop_actor = 2.0 * (371 * 256 + 256 * 256 + 256 * 24)
    cfgs = [(rows, ao) for rows in (32, 16) for ao in (False, True)]
    times = {c: [] for c in cfgs}
    for _ in range(reps):  # the forms take turns, so that clock ramps and neighbours hit all of them alike
        for rows, ao in cfgs:
            f.set_tile_rows(rows)
            for _ in range(10):
                f.forward(x, actor_only=ao)
            f.timing_begin()
            for _ in range(launches):
                f.forward(x, actor_only=ao)
            ms, k = f.timing_end()
            times[(rows, ao)].append(ms / k)
    out = {"rows": n, "inputs": "env observations" if ENV_OBS else "N(0,1)"}
    for (rows, ao), ts in times.items():
        ts = sorted(ts)
        t = ts[len(ts) // 2]
        flop = (flop_actor if ao else FLOP_BOTH) * n
        out[f"tile{rows}_{'actor' if ao else 'both'}"] = {"us_median": round(t * 1e3, 2), "us_min": round(ts[0] * 1e3, 2),
                                                          "frac_of_157.3": round(flop / t / 1e9 / 157.3, 3)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
