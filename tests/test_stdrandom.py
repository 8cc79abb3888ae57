"""evomotion_amd/stdrandom.py — std::mt19937 + libstdc++'s std::shuffle restated, the generator of the reference's replay buffers
(evo_motion_networks/src/replay_buffer.cpp:14,21,66,83) — against (1) the container's own g++ / libstdc++ on a few hundred shuffles of
one stream, (2) the trajectory draws the COMPILED reference made in its PPO loop and (3) the transition draws of its SAC loop
(tests/golden/agent_loop_golden.txt, sac_loop_golden.txt: recorded by oracle/ref_loop.cpp / ref_sac_loop.cpp)."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("evm_stdrandom", os.path.join(ROOT, "evomotion_amd", "stdrandom.py"))
sr = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sr)

CPP = r"""
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
int main() {
    std::mt19937 g(1234);
    const unsigned first = (unsigned) g(), second = (unsigned) g();   // (two statements: argument evaluation order is unspecified)
    printf("%u %u\n", first, second);
    for (int n = 0; n < 300; n++) {
        std::vector<int> v(n);
        std::iota(v.begin(), v.end(), 0);
        std::shuffle(v.begin(), v.end(), g);
        for (int x : v) printf("%d ", x);
        printf("\n");
    }
    std::vector<int> big(70000);   // beyond 65535 elements std::shuffle draws one position per call
    std::iota(big.begin(), big.end(), 0);
    std::shuffle(big.begin(), big.end(), g);
    unsigned long long h = 1469598103934665603ull;
    for (int x : big) h = (h ^ (unsigned) x) * 1099511628211ull;
    printf("%llu\n", h);
}
"""


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_against_libstdcxx(tmp_path):
    src, exe = tmp_path / "s.cpp", tmp_path / "s"
    src.write_text(CPP)
    subprocess.check_call(["g++", "-O1", "-o", str(exe), str(src)])
    lines = subprocess.check_output([str(exe)]).decode().split("\n")
    g = sr.Mt19937(1234)
    assert [int(v) for v in lines[0].split()] == [g(), g()]
    for n in range(300):
        assert sr.std_shuffle(list(range(n)), g) == [int(v) for v in lines[1 + n].split()], n
    big = sr.std_shuffle(list(range(70000)), g)
    h = 1469598103934665603
    for x in big:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert h == int(lines[301])


def test_the_compiled_reference_sac_draws():
    import test_sac_loop as tl
    gold = tl.load_sac_loop_golden()
    S, A, H, batch_size, epoch, replay, train_every = gold["config"]
    g = sr.Mt19937(1234)                      # ReplayBuffer(replay_buffer_size, seed): rand_gen(seed), soft_actor_critic.cpp:34
    i = 0
    for k, t, gstep, size_after, train in gold["act"]:
        for e in range(epoch if train else 0):
            index = sr.std_shuffle(list(range(size_after - 1)), g)       # replay_buffer.cpp:19-21
            assert index[:batch_size] == gold["sample"][i], (k, t, e)
            i += 1
    assert i == gold["trains"] == 10


def test_the_compiled_reference_ppo_draws():
    import test_agent_loop as tl
    gold = tl.load_loop_golden()
    S, A, H, epoch, batch_size, train_every, replay = gold["config"]
    g = sr.Mt19937(1234)                      # TrajectoryReplayBuffer(replay_buffer_size, seed): ppo_gae.cpp:22
    i = 0
    for k, (gk, gstep, gmem, gfilt, gtrain) in enumerate(gold["done"]):
        if not gtrain:
            continue
        lens = list(gold["buffer"][k - 1]) if k else [0]                # trajectory lengths after the previous done(), the open one last
        lens[-1] = gold["lengths"][k]                                     # the open trajectory has this episode's steps by now
        filtered = [p for p, n in enumerate(lens) if n > 1]
        assert len(filtered) == gfilt
        index = sr.std_shuffle(list(range(len(filtered) - 1)), g)       # replay_buffer.cpp:81-83
        assert [filtered[j] for j in index[:batch_size]] == gold["sample"][i], k
        i += 1
    assert i == gold["trains"] == 4
