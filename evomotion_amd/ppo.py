"""PPO / GAE update on the device (evm_ppo_*, csrc/ppo_kernels.hip): PpoGaeAgent::train (ppo_gae.cpp:117-190) without
PyTorch autograd.  The trainer owns the master weights, gradients and Adam moments; torch is used for the rollout
buffers and, with more than one rank, for the RCCL exchange of the advantage statistics and the gradients."""
import ctypes

import torch

from ._lib import lib, check

PARAMS, GRADS, EXP_AVG, EXP_AVG_SQ = 0, 1, 2, 3
ACTOR_DEV_STEP = 2  # adam_step(): the actor's device-side step counter (actor_apply, SAC)
ACTOR, CRITIC = 0, 1


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _dist_ready():
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


def _via_host():
    """RCCL ("nccl") exchanges device tensors directly; the gloo rehearsal of the N > 1 path goes through host copies"""
    return torch.distributed.get_backend() != "nccl"


def _all_reduce_sum(t):
    """in place.  RCCL: enqueued on the current stream behind the kernels that produced t, nothing waits on the host"""
    if _via_host():
        h = t.cpu()
        torch.distributed.all_reduce(h)
        t.copy_(h)
    else:
        torch.distributed.all_reduce(t)


def _all_gather_into(out, t):
    """out [world, ...] <- every rank's t, rank order"""
    if _via_host():
        h = torch.empty(out.numel(), dtype=out.dtype)
        torch.distributed.all_gather_into_tensor(h, t.reshape(-1).cpu())
        out.copy_(h.view(out.shape))
    else:
        torch.distributed.all_gather_into_tensor(out.view(-1), t.reshape(-1))


class DeviceCount:
    """The number of selected transitions over all ranks.  It stays on the device (evm_ppo_grads reads it there); float() is a
    host read for tests and logs."""

    def __init__(self, stats):
        self._stats = stats

    def __float__(self):
        return float(self._stats[0])


class FusedPpoTrainer:
    """fused: the FusedActorCritic whose weights the trainer keeps up to date (borrowed; must outlive the trainer)."""

    def __init__(self, fused, max_rows):
        self.fused = fused
        self.device = fused.device
        self.max_rows = int(max_rows)
        self._h = ctypes.c_void_p()
        torch.cuda.set_device(self.device)
        check(lib.evm_ppo_create(fused._h, self.max_rows, ctypes.byref(self._h)))
        na, nc = ctypes.c_size_t(), ctypes.c_size_t()
        check(lib.evm_policy_param_counts(fused._h, ctypes.byref(na), ctypes.byref(nc)))
        self.n_params = (na.value, nc.value)
        self._stats = torch.zeros(3, device=self.device, dtype=torch.float64)
        self._all_stats = None
        # actor and critic gradients are one vector: the data-parallel exchange is ONE in-place all-reduce per epoch
        gp, gn, go = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_size_t()
        check(lib.evm_ppo_grad_buffer(self._h, None, ctypes.byref(gp), ctypes.byref(gn), ctypes.byref(go)))
        self.grad_floats, self.critic_grad_offset = gn.value, go.value
        self._grads = None
        self.last_rows = 0
        # train(): run the epochs on the selected rows only (False, or EVM_PPO_COMPACT=0 for A/B runs: all rows, the mask zeroes the rest)
        import os
        self.compact_rows = os.environ.get("EVM_PPO_COMPACT", "1") != "0"

    def grad_buffer(self):
        """[grad_floats] device vector, actor gradients at 0, critic gradients at critic_grad_offset; owned by this object and
        handed to the trainer (evm_ppo_grad_buffer) the first time it is asked for"""
        if self._grads is None:
            self._grads = torch.zeros(self.grad_floats, device=self.device)
            check(lib.evm_ppo_grad_buffer(self._h, _ptr(self._grads), None, None, None))
        return self._grads

    def close(self):
        if self._h:
            lib.evm_ppo_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- parameters / optimiser state -------------------------------------------------------------------------
    def set_modules(self, actor, critic, reset_optimizer=True):
        flat = lambda m: torch.cat([p.detach().reshape(-1).float() for p in m.parameters()]).to(self.device).contiguous()
        a, c = flat(actor), flat(critic)
        assert a.numel() == self.n_params[0] and c.numel() == self.n_params[1]
        check(lib.evm_ppo_set_params(self._h, _ptr(a), _ptr(c), 1 if reset_optimizer else 0, self._stream()))
        torch.cuda.current_stream(self.device).synchronize()  # a, c may be freed

    def vector(self, what, net, out=None):
        if out is None:
            out = torch.empty(self.n_params[net], device=self.device)
        check(lib.evm_ppo_copy(self._h, what, net, 0, _ptr(out), self._stream()))
        return out

    def load_vector(self, what, net, t):
        t = t.to(self.device, torch.float32).contiguous()
        assert t.numel() == self.n_params[net]
        check(lib.evm_ppo_copy(self._h, what, net, 1, _ptr(t), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    def adam_step(self, net, set_step=-1):
        s = ctypes.c_int()
        check(lib.evm_ppo_adam_step(self._h, net, set_step, ctypes.byref(s)))
        return s.value

    def params_into(self, actor, critic):
        """copy the trainer's weights into torch modules (checkpoints, evaluation)"""
        for net, mod in ((ACTOR, actor), (CRITIC, critic)):
            flat = self.vector(PARAMS, net)
            o = 0
            with torch.no_grad():
                for p in mod.parameters():
                    n = p.numel()
                    p.copy_(flat[o:o + n].view_as(p))
                    o += n

    # ---- one PpoGaeAgent::train call ----------------------------------------------------------------------------
    def gae(self, rewards, done_u8, values, next_values, mask_u8, gamma, lam):
        """time-major [T, N] tensors -> (normalised advantages, returns), statistics over all ranks"""
        T, N = rewards.shape
        adv = torch.empty(T, N, device=self.device)
        ret = torch.empty(T, N, device=self.device)
        s = self._stream()
        check(lib.evm_ppo_gae(self._h, T, N, _ptr(rewards), _ptr(done_u8), _ptr(values), _ptr(next_values), _ptr(mask_u8),
                              gamma, lam, _ptr(adv), _ptr(self._stats), s))
        if _dist_ready():
            # every rank's (count, mean, M2) -> Chan's merge in rank order ON THE DEVICE: the update never reads the host
            w = torch.distributed.get_world_size()
            if self._all_stats is None or self._all_stats.shape[0] != w:
                self._all_stats = torch.zeros(w, 3, device=self.device, dtype=torch.float64)
            _all_gather_into(self._all_stats, self._stats)
            check(lib.evm_ppo_gae_merge(self._h, _ptr(self._all_stats), w, _ptr(self._stats), s))
        check(lib.evm_ppo_gae_normalize(self._h, T, N, None, _ptr(values), _ptr(adv), _ptr(ret), s))
        return adv, ret, DeviceCount(self._stats)

    def epoch(self, states, actions, logp_old, adv, returns, mask_u8, n_selected_global, epsilon, entropy_factor,
              critic_loss_factor, learning_rate, clip_grad_norm, states_unchanged=False):
        """one forward / backward / optimiser step over the rows; states_unchanged: `states` holds what it held in the previous
        call (later epochs of one train call), the trainer reuses its aligned copy"""
        rows = states.shape[0]
        for t in (states, actions, logp_old, adv, returns, mask_u8):
            assert t.is_cuda and t.is_contiguous()
        self._epoch_ptrs(rows, [_ptr(t) for t in (states, actions, logp_old, adv, returns, mask_u8)], n_selected_global, epsilon,
                         entropy_factor, critic_loss_factor, learning_rate, clip_grad_norm, states_unchanged)

    def _epoch_ptrs(self, rows, ptrs, n_selected_global, epsilon, entropy_factor, critic_loss_factor, learning_rate, clip_grad_norm,
                    states_unchanged):
        """epoch() on device addresses (states, actions, logp_old, adv, returns, mask): the caller's tensors or the trainer's own
        dense copies of the selected rows (select_rows)"""
        assert 1 <= rows <= self.max_rows
        self.last_rows = rows  # what the epoch's kernels work on (bench.py prices its roofline on this)
        s = self._stream()
        dist = _dist_ready()
        g = self.grad_buffer() if dist else None
        # a DeviceCount (what gae() returns) stays on the device: the loss kernels read 1 / count there
        n_sel = -1.0 if isinstance(n_selected_global, DeviceCount) else float(n_selected_global)
        check(lib.evm_ppo_grads(self._h, rows, *ptrs, n_sel, epsilon, entropy_factor, critic_loss_factor,
                                1 if states_unchanged else 0, s))
        if dist:
            # the losses are normalised by the global count, so the SUM over ranks is the global gradient.  One collective over
            # [actor | critic], in place in the buffer the gradient kernels wrote and the optimiser kernel reads, ordered by the
            # stream: no copy, no synchronize()
            _all_reduce_sum(g)
        check(lib.evm_ppo_apply(self._h, learning_rate, clip_grad_norm, s))

    def select_rows(self, states, actions, logp_old, adv, returns, mask_u8):
        """evm_ppo_select_rows: the rows inside the mask, in order, as dense copies owned by the trainer -> (count, addresses of
        states, actions, logp_old, adv, returns, mask of ones).  One host read of the count."""
        rows = states.shape[0]
        for t in (states, actions, logp_old, adv, returns, mask_u8):
            assert t.is_cuda and t.is_contiguous()
        n = ctypes.c_size_t()
        out = [ctypes.c_void_p() for _ in range(6)]
        check(lib.evm_ppo_select_rows(self._h, rows, _ptr(mask_u8), _ptr(states), _ptr(actions), _ptr(logp_old), _ptr(adv), _ptr(returns),
                                      ctypes.byref(n), *[ctypes.byref(o) for o in out], self._stream()))
        return int(n.value), out

    # ---- SAC's actor step on the same kernels ----------------------------------------------------------------------
    def actor_forward(self, states, mu=None, sigma=None):
        rows = states.shape[0]
        A = self.fused.A
        mu = torch.empty(rows, A, device=self.device) if mu is None else mu
        sigma = torch.empty(rows, A, device=self.device) if sigma is None else sigma
        check(lib.evm_ppo_actor_forward(self._h, rows, _ptr(states), _ptr(mu), _ptr(sigma), self._stream()))
        return mu, sigma

    def actor_backward(self, dmu, dsigma):
        check(lib.evm_ppo_actor_backward(self._h, dmu.shape[0], _ptr(dmu), _ptr(dsigma), self._stream()))

    def actor_apply(self, learning_rate):
        """Adam step of the actor alone, step counter on the device (replayable from a captured graph)"""
        check(lib.evm_ppo_actor_apply(self._h, learning_rate, self._stream()))

    def set_flat(self, actor_flat, critic_flat, reset_optimizer=False):
        """flat device parameter vectors -> trainer and rollout kernel, asynchronously on the current stream"""
        check(lib.evm_ppo_set_params(self._h, _ptr(actor_flat), _ptr(critic_flat), 1 if reset_optimizer else 0, self._stream()))

    def losses(self):
        a, c = ctypes.c_double(), ctypes.c_double()
        check(lib.evm_ppo_losses(self._h, ctypes.byref(a), ctypes.byref(c), self._stream()))
        return a.value, c.value

    def train(self, states, actions, rewards, done_u8, logp_old, values, next_values, mask_u8, gamma, lam, epsilon,
              entropy_factor, critic_loss_factor, epoch, learning_rate, clip_grad_norm, loss_hook=None):
        """time-major rollout: states [T, N, S], actions / logp_old [T, N, A], the rest [T, N].  loss_hook(actor_loss,
        critic_loss): called after every epoch with that epoch's losses (the LossMeter adds of ppo_gae.cpp:185-186; reading
        them synchronises the stream, so the fast path leaves it None)"""
        T, N = rewards.shape
        adv, ret, n_glob = self.gae(rewards, done_u8, values, next_values, mask_u8, gamma, lam)
        # nothing selected on any rank: the count stays on the device (no host read here), and there the update turns into a no-op —
        # zero gradients, k_ppo_adam leaves weights and moments untouched, the losses read NaN — so a caller can see that an update
        # was empty (VecPpoGaeAgent.update() returns them).  The host-side Adam step counters still advance: the next real step's
        # bias corrections are one step ahead, which the reference cannot reach (it returns before training, ppo_gae.cpp:63-66)
        S, A = states.shape[-1], actions.shape[-1]
        st, ac, lp = states.reshape(T * N, S), actions.reshape(T * N, A), logp_old.reshape(T * N, A)
        advf, retf, mk = adv.reshape(-1).contiguous(), ret.reshape(-1).contiguous(), mask_u8.reshape(-1).contiguous()
        rows, ptrs = T * N, [_ptr(t) for t in (st, ac, lp, advf, retf, mk)]
        for t in (st, ac, lp):
            assert t.is_cuda and t.is_contiguous()
        if self.compact_rows and not torch.cuda.is_current_stream_capturing():
            # Rows outside the mask (reset()'s settle calls and emissions: 30-40 % of a rollout) weigh nothing in either loss
            # (ppo_gae.cpp:167-168, 178: means over masked_select), yet every epoch would push them through forward, backward and
            # the weight-gradient GEMMs.  After GAE — which needs the time structure — the update is a sum over rows, so only the
            # selected rows are kept (in their order: the sums see the same terms, grouped into other tiles).  Costs one host read
            # of the local count per train() call.
            nv, sel = self.select_rows(st, ac, lp, advf, retf, mk)
            if 0 < nv < rows:
                rows, ptrs = nv, sel
        for ep in range(epoch):
            self._epoch_ptrs(rows, ptrs, n_glob, epsilon, entropy_factor, critic_loss_factor, learning_rate, clip_grad_norm,
                             states_unchanged=ep > 0)
            if loss_hook is not None:
                loss_hook(*self.losses())
        return self.losses()

    def timing(self, enable):
        ms, n = ctypes.c_float(), ctypes.c_int()
        check(lib.evm_ppo_timing(self._h, 1 if enable else 0, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value
