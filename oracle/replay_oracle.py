"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatements for the replay memory of the SAC rows.

  FlatReplayBuffer   literal restatement of the reference's flat buffer, one env:
                     evo_motion_networks/src/replay_buffer.cpp:16-52 (sample / add / update_last / has_enough),
                     :146-153 (update_last_item); pinned by the `replay ...` lines of tests/golden/sac_golden.txt,
                     which the reference's own compiled class printed (oracle/ref_sac.cpp)
  RingOracle         numpy model of the device ring (evm_replay_*): time-major slots, implicit next state, ordered
                     compaction of valid rows, keyed permutation draw — bit-exact counterpart of replay_kernels.hip
"""
import numpy as np

M32 = 0xFFFFFFFF


class FlatReplayBuffer:
    def __init__(self, size):
        self.size, self.memory = size, []

    def empty(self):
        return not self.memory

    def add(self, state, action, reward=0.0, done=False, next_state=None):
        self.memory.append([state, action, reward, done, state if next_state is None else next_state])
        while len(self.memory) > self.size:
            self.memory.pop(0)

    def update_last(self, reward, next_state, done):
        self.memory[-1][2:] = [reward, done, next_state]

    def has_enough(self, batch_size):
        return len(self.memory) - 1 >= batch_size

    def sampleable(self):
        return self.memory[:-1]  # sample() shuffles indices 0 .. size-2: never the newest item


def _mix(x, key, mask):
    for r in range(3):
        x = (x * 0x9E3779B1 + key) & mask
        x ^= x >> 7
        x &= mask
        x = (x * 0x85EBCA6B + (key >> 16) + r) & mask
        x ^= x >> 11
        x &= mask
    return x


def replay_rank(b, m, seed):
    mask = 1
    while mask < m:
        mask <<= 1
    mask -= 1
    key = ((((seed ^ (seed >> 32)) & M32) * 0x27D4EB2F) + 0x165667B1) & M32
    x = b
    while True:
        x = _mix(x, key, mask)
        if x < m:
            return x


class RingOracle:
    def __init__(self, C, N, S, A):
        self.C, self.N, self.S, self.A = C, N, S, A
        self.state = np.zeros((C, N, S), np.float32)
        self.action = np.zeros((C, N, A), np.float32)
        self.reward = np.zeros((C, N), np.float32)
        self.done = np.zeros((C, N), np.float32)
        self.pending = np.zeros((N, S), np.float32)
        self.valid_idx = [np.zeros(0, np.int64) for _ in range(C)]
        self.head = self.live = self.pushes = 0

    def push(self, state, action, reward, done, valid, next_state):
        s = self.head
        self.state[s], self.action[s], self.reward[s] = state, action, reward
        self.done[s] = (np.asarray(done) != 0).astype(np.float32)
        self.pending[:] = next_state
        v = np.ones(self.N, bool) if valid is None else (np.asarray(valid) == 1)
        self.valid_idx[s] = np.nonzero(v)[0]
        self.head = (self.head + 1) % self.C
        self.live = min(self.live + 1, self.C)
        self.pushes += 1

    def slots_in_age_order(self):
        return [(self.head - self.live + j) % self.C for j in range(self.live)]

    def transitions(self):
        return sum(len(self.valid_idx[s]) for s in self.slots_in_age_order())

    def plan(self, batch, seed):
        order = self.slots_in_age_order()
        prefix = np.concatenate([[0], np.cumsum([len(self.valid_idx[s]) for s in order])]).astype(np.int64)
        m = int(prefix[-1])
        out = np.full((batch, 2), -1, np.int32)
        if m == 0:
            return out
        for b in range(batch):
            r = replay_rank(b % m, m, seed)
            j = int(np.searchsorted(prefix, r, side="right") - 1)
            out[b] = (order[j], self.valid_idx[order[j]][r - prefix[j]])
        return out

    def sample(self, batch, seed):
        plan = self.plan(batch, seed)
        newest = (self.head - 1) % self.C
        st = np.zeros((batch, self.S), np.float32); nx = np.zeros((batch, self.S), np.float32)
        ac = np.zeros((batch, self.A), np.float32); rw = np.zeros(batch, np.float32); dn = np.zeros(batch, np.float32)
        for b, (s, e) in enumerate(plan):
            if s < 0:
                continue
            st[b], ac[b], rw[b], dn[b] = self.state[s, e], self.action[s, e], self.reward[s, e], self.done[s, e]
            nx[b] = self.pending[e] if s == newest else self.state[(s + 1) % self.C, e]
        return st, ac, rw, dn, nx, plan
