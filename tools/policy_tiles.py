"""Time evm_policy_forward at 4096 rows in its two tile forms (32 rows on v_mfma_f32_32x32x2_f32, 16 rows on
v_mfma_f32_16x16x4_f32), both networks and actor only.  HIP events around 200 launches each; prints one JSON line.
Usage (GPU box): python tools/policy_tiles.py [rows ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evomotion_amd import ActorModule, CriticModule, FusedActorCritic  # noqa: E402

FLOP_BOTH = 654848.0  # GEMM FLOP per row, actor + critic (bench.py)


def main():
    for n in ([int(a) for a in sys.argv[1:]] or [4096]):
        one(n)


def one(n, reps=7, launches=100):
    actor, critic = ActorModule([371], [12], 256).cuda(), CriticModule([371], 256).cuda()
    f = FusedActorCritic(371, 12, 256, 0)
    f.load_modules(actor, critic)
    x = torch.randn(n, 371, device="cuda")
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
    out = {"rows": n}
    for (rows, ao), ts in times.items():
        ts = sorted(ts)
        t = ts[len(ts) // 2]
        flop = (flop_actor if ao else FLOP_BOTH) * n
        out[f"tile{rows}_{'actor' if ao else 'both'}"] = {"us_median": round(t * 1e3, 2), "us_min": round(ts[0] * 1e3, 2),
                                                          "frac_of_157.3": round(flop / t / 1e9 / 157.3, 3)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
