"""Device replay ring (evm_replay_*) against the numpy ring oracle — bit-exact (pure copies and integer indexing)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import replay_oracle as ro  # noqa: E402

pytestmark = pytest.mark.gpu


def drive(C, N, S, A, steps, seed, with_valid=True):
    import torch
    from evomotion_amd.replay import ReplayRing
    rng = np.random.default_rng(seed)
    ring, orc = ReplayRing(C, N, S, A), ro.RingOracle(C, N, S, A)
    state = rng.normal(size=(N, S)).astype(np.float32)
    for t in range(steps):
        action = rng.uniform(-1, 1, (N, A)).astype(np.float32)
        reward = rng.normal(size=N).astype(np.float32)
        done = (rng.random(N) < 0.1).astype(np.uint8)
        valid = rng.choice(np.array([0, 1, 1, 1, 2], np.uint8), N) if with_valid else None
        nxt = rng.normal(size=(N, S)).astype(np.float32)
        cu = lambda a: torch.from_numpy(a).cuda()
        ring.push(cu(state), cu(action), cu(reward), cu(done), None if valid is None else cu(valid), cu(nxt))
        orc.push(state, action, reward, done, valid, nxt)
        state = nxt
    return ring, orc


@pytest.mark.parametrize("C,N,S,A,steps", [(4, 70, 371, 12, 3), (4, 70, 371, 12, 11), (3, 1, 5, 2, 7), (6, 1500, 19, 3, 9)])
def test_ring_matches_oracle(C, N, S, A, steps):
    ring, orc = drive(C, N, S, A, steps, seed=C * 1000 + steps)
    st = ring.stats()
    assert st == dict(transitions=orc.transitions(), live_slots=orc.live, pushes=orc.pushes)
    for batch, seed in ((1, 0), (33, 7), (orc.transitions(), 123456789012345), (orc.transitions() + 5, 3)):
        got = ring.sample(batch, seed, want_index=True)
        want = orc.sample(batch, seed)
        assert np.array_equal(got[5].cpu().numpy(), want[5])
        for g, w in zip(got[:5], want[:5]):
            assert np.array_equal(g.cpu().numpy(), w)
    # a full draw is a permutation of the stored transitions
    m = orc.transitions()
    idx = ring.sample(m, 99, want_index=True)[5].cpu().numpy()
    assert len({tuple(r) for r in idx}) == m


def test_all_rows_valid_when_no_mask_and_unaligned_sizes():
    ring, orc = drive(2, 33, 7, 3, 5, seed=1, with_valid=False)  # N*S*4 not a multiple of 16: scalar copy path
    assert ring.stats()["transitions"] == 2 * 33
    got, want = ring.sample(66, 4, want_index=True), orc.sample(66, 4)
    for g, w in zip(got[:5], want[:5]):
        assert np.array_equal(g.cpu().numpy(), w)


def test_full_size_round_trip_properties():
    """BASELINE configs[4] size (4096 envs, robot_walk widths): every drawn row is a stored row and its next state is the
    state stored one slot later for the same env (size-independent properties, no oracle loop)."""
    import torch
    from evomotion_amd.replay import ReplayRing
    C, N, S, A = 8, 4096, 371, 12
    ring = ReplayRing(C, N, S, A)
    g = torch.Generator(device="cuda").manual_seed(0)
    states = [torch.randn(N, S, device="cuda", generator=g) for _ in range(C + 3)]
    valids = []
    for t in range(C + 2):
        valid = (torch.rand(N, device="cuda", generator=g) < 0.6).to(torch.uint8)
        valids.append(valid)
        action = torch.full((N, A), float(t), device="cuda")
        ring.push(states[t], action, torch.full((N,), float(t), device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda"),
                  valid, states[t + 1])
    live = list(range(2, C + 2))  # the last C pushes
    assert ring.stats()["transitions"] == int(sum(int(valids[t].sum()) for t in live))
    st, ac, rw, dn, nx, idx = ring.sample(10000, 5, want_index=True)
    t_of = rw.long()  # reward carries the push number
    assert int(t_of.min()) >= 2 and int(t_of.max()) <= C + 1
    env = idx[:, 1].long()
    all_states = torch.stack(states)  # [T, N, S]
    assert torch.equal(st, all_states[t_of, env]) and torch.equal(nx, all_states[t_of + 1, env])
    assert torch.equal(ac[:, 0], t_of.float())
    assert bool(torch.stack(valids)[t_of, env].all())
    assert len(set(map(tuple, idx.cpu().numpy()))) == 10000  # distinct draws (fewer than the ~19.6 k stored)


def test_errors():
    from evomotion_amd import EvmError
    from evomotion_amd.replay import ReplayRing
    with pytest.raises(ValueError):  # EVM_E_INVALID <-> std::invalid_argument
        ReplayRing(0, 4, 3, 1)
    ring = ReplayRing(2, 4, 3, 1)
    with pytest.raises(EvmError):
        ring.sample(4, 0)  # empty memory
