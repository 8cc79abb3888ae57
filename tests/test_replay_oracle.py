"""Replay-memory oracles (oracle/replay_oracle.py): the flat buffer against the lines the reference's own ReplayBuffer
printed (tests/golden/sac_golden.txt), and the ring model against the flat buffer for one env."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import replay_oracle as ro  # noqa: E402

SAC_GOLDEN = os.path.join(ROOT, "tests", "golden", "sac_golden.txt")


def golden_lines():
    return [l.rstrip("\n") for l in open(SAC_GOLDEN) if l.startswith("replay ")]


def test_flat_buffer_matches_the_reference_trace():
    lines = golden_lines()
    assert lines[0] == "replay empty 1"
    rb = ro.FlatReplayBuffer(4)
    assert rb.empty()
    k = 0
    for l in lines[1:]:
        if not l.startswith("replay after_add"):
            continue
        if not rb.empty():
            rb.update_last(float(10 + k), float(k), k == 3)
        rb.add(float(k), float(100 + k))
        m = re.match(r"replay after_add (\d+) size (\d+) has_enough2 (\d) has_enough4 (\d) :(.*)", l)
        assert int(m.group(1)) == k and int(m.group(2)) == len(rb.memory)
        assert int(m.group(3)) == int(rb.has_enough(2)) and int(m.group(4)) == int(rb.has_enough(4))
        items = re.findall(r"\(([^)]*)\)", m.group(5))
        got = [tuple(float(v) for v in it.split(",")) for it in items]
        want = [(s, a, r, float(d), n) for s, a, r, d, n in rb.memory]
        assert got == want, (k, got, want)
        k += 1
    assert k == 6 and lines[-1] == "replay sample3 never_the_newest 1 count 3"


def test_ring_holds_the_same_transitions_as_the_flat_buffer_for_one_env():
    """N = 1: after every step the ring's stored transitions equal the flat buffer's sampleable items plus the newest
    one (the ring is pushed after the env step, when that transition is already complete)."""
    rng = np.random.default_rng(3)
    C, S, A, T = 5, 3, 2, 17
    ring, flat = ro.RingOracle(C, 1, S, A), ro.FlatReplayBuffer(C + 1)
    state = rng.normal(size=S).astype(np.float32)
    for t in range(T):
        action = rng.normal(size=A).astype(np.float32)
        # reference order (soft_actor_critic.cpp:53-56 / 172): update_last(previous) happens at the next act()/done()
        flat.add(state.copy(), action.copy())
        nxt, reward, done = rng.normal(size=S).astype(np.float32), float(rng.normal()), bool(rng.random() < 0.3)
        flat.update_last(reward, nxt.copy(), done)
        ring.push(state[None], action[None], np.array([reward], np.float32), np.array([done]), None, nxt[None])
        want = flat.memory[-ring.live:]
        st, ac, rw, dn, nx, plan = ring.sample(ring.transitions(), seed=t)
        got = sorted((tuple(st[b]), tuple(ac[b]), float(rw[b]), float(dn[b]), tuple(nx[b])) for b in range(len(rw)))
        exp = sorted((tuple(s), tuple(a), np.float32(r).item(), float(d), tuple(n)) for s, a, r, d, n in want)
        assert got == exp
        state = nxt


def test_draw_is_a_permutation_and_skips_invalid_rows():
    rng = np.random.default_rng(0)
    ring = ro.RingOracle(4, 7, 2, 1)
    for t in range(6):
        valid = rng.integers(0, 3, 7).astype(np.uint8)  # codes 0, 1, 2: only 1 is a transition
        ring.push(rng.normal(size=(7, 2)), rng.normal(size=(7, 1)), rng.normal(size=7), rng.integers(0, 2, 7), valid, rng.normal(size=(7, 2)))
    m = ring.transitions()
    for seed in (0, 1, 2**40 + 5):
        plan = ring.plan(m, seed)
        assert len({tuple(p) for p in plan}) == m  # all stored transitions, each exactly once
        for s, e in plan:
            assert e in ring.valid_idx[s]
    assert sorted(ro.replay_rank(b, 1000, 9) for b in range(1000)) == list(range(1000))
    plan = ring.plan(m + 3, 5)  # more draws than transitions: ranks wrap around
    assert [tuple(p) for p in plan[m:]] == [tuple(p) for p in plan[:3]]
