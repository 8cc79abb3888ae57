"""Device-resident replay ring (evm_replay_*, include/evomotion.h) — ReplayBuffer of the reference for N envs at once
(evo_motion_networks/src/replay_buffer.cpp:16-52,146-153)."""
import ctypes

import torch

from ._lib import check, lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class ReplayRing:
    def __init__(self, capacity_slots, n_envs, state_dim, action_dim, device=0):
        self.C, self.N, self.S, self.A = capacity_slots, n_envs, state_dim, action_dim
        self.device = torch.device("cuda", device)
        self._h = ctypes.c_void_p()
        check(lib.evm_replay_create(capacity_slots, n_envs, state_dim, action_dim, device, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib.evm_replay_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def push(self, state, action, reward, done, valid, next_state):
        """One rollout step of all envs; `done` / `valid` are uint8 tensors (valid may be None: all rows count)."""
        for t, shape in ((state, (self.N, self.S)), (action, (self.N, self.A)), (next_state, (self.N, self.S))):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape
        assert reward.dtype == torch.float32 and done.dtype == torch.uint8 and (valid is None or valid.dtype == torch.uint8)
        check(lib.evm_replay_push(self._h, _ptr(state), _ptr(action), _ptr(reward), _ptr(done), _ptr(valid), _ptr(next_state),
                                  self._stream()))

    def stats(self):
        out = (ctypes.c_longlong * 3)()
        check(lib.evm_replay_stats(self._h, out, self._stream()))
        return dict(transitions=int(out[0]), live_slots=int(out[1]), pushes=int(out[2]))

    def sample(self, batch, seed, want_index=False, out=None):
        """`out` = (states, actions, rewards, done, next_states) to fill in place (static buffers of a captured update)."""
        z = lambda *s, **k: torch.empty(*s, device=self.device, **k)
        states, actions, rewards, done, nxt = out if out is not None else (z(batch, self.S), z(batch, self.A), z(batch), z(batch), z(batch, self.S))
        assert states.shape == (batch, self.S) and nxt.shape == (batch, self.S) and states.is_contiguous() and nxt.is_contiguous()
        index = z(batch, 2, dtype=torch.int32) if want_index else None
        check(lib.evm_replay_sample(self._h, batch, ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), _ptr(states), _ptr(actions),
                                    _ptr(rewards), _ptr(done), _ptr(nxt), _ptr(index), self._stream()))
        out = (states, actions, rewards, done, nxt)
        return out + (index,) if want_index else out

    def timing_begin(self):
        check(lib.evm_replay_timing_begin(self._h))

    def timing_end(self):
        a, b = ctypes.c_float(), ctypes.c_float()
        na, nb = ctypes.c_int(), ctypes.c_int()
        check(lib.evm_replay_timing_end(self._h, self._stream(), ctypes.byref(a), ctypes.byref(na), ctypes.byref(b), ctypes.byref(nb)))
        return dict(ms_push=a.value, n_push=na.value, ms_sample=b.value, n_sample=nb.value)
