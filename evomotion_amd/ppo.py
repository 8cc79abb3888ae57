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
    if _via_host():
        h = t.cpu()
        torch.distributed.all_reduce(h)
        t.copy_(h)
    else:
        torch.distributed.all_reduce(t)


def _all_gather(t):
    w = torch.distributed.get_world_size()
    src = t.cpu() if _via_host() else t
    out = [torch.zeros_like(src) for _ in range(w)]
    torch.distributed.all_gather(out, src)
    return [o.to(t.device) for o in out]


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
        n_glob = None
        if _dist_ready():
            allt = _all_gather(self._stats)
            n, mean, m2 = allt[0]
            for t in allt[1:]:  # Chan's merge, same order on every rank
                nb, mb, m2b = t
                tot = n + nb
                if float(tot) == 0:
                    continue
                dlt = mb - mean
                mean = mean + dlt * nb / tot
                m2 = m2 + m2b + dlt * dlt * n * nb / tot
                n = tot
            self._stats.copy_(torch.stack([n, mean, m2]))
            n_glob = float(n)
        check(lib.evm_ppo_gae_normalize(self._h, T, N, _ptr(self._stats), _ptr(values), _ptr(adv), _ptr(ret), s))
        if n_glob is None:
            n_glob = float(self._stats[0])
        return adv, ret, n_glob

    def epoch(self, states, actions, logp_old, adv, returns, mask_u8, n_selected_global, epsilon, entropy_factor,
              critic_loss_factor, learning_rate, clip_grad_norm, states_unchanged=False):
        """one forward / backward / optimiser step over the rows; states_unchanged: `states` holds what it held in the previous
        call (later epochs of one train call), the trainer reuses its aligned copy"""
        rows = states.shape[0]
        assert rows <= self.max_rows
        for t in (states, actions, logp_old, adv, returns, mask_u8):
            assert t.is_cuda and t.is_contiguous()
        s = self._stream()
        check(lib.evm_ppo_grads(self._h, rows, _ptr(states), _ptr(actions), _ptr(logp_old), _ptr(adv), _ptr(returns), _ptr(mask_u8),
                                float(n_selected_global), epsilon, entropy_factor, critic_loss_factor, 1 if states_unchanged else 0, s))
        if _dist_ready():
            for net in (ACTOR, CRITIC):
                g = self.vector(GRADS, net)
                _all_reduce_sum(g)  # the losses are normalised by the global count: SUM is the global gradient
                check(lib.evm_ppo_copy(self._h, GRADS, net, 1, _ptr(g), s))
                torch.cuda.current_stream(self.device).synchronize()
        check(lib.evm_ppo_apply(self._h, learning_rate, clip_grad_norm, s))

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
              entropy_factor, critic_loss_factor, epoch, learning_rate, clip_grad_norm):
        """time-major rollout: states [T, N, S], actions / logp_old [T, N, A], the rest [T, N]"""
        T, N = rewards.shape
        adv, ret, n_glob = self.gae(rewards, done_u8, values, next_values, mask_u8, gamma, lam)
        if n_glob < 1:
            return float("nan"), float("nan")
        S, A = states.shape[-1], actions.shape[-1]
        st, ac, lp = states.reshape(T * N, S), actions.reshape(T * N, A), logp_old.reshape(T * N, A)
        for ep in range(epoch):
            self.epoch(st, ac, lp, adv.reshape(-1), ret.reshape(-1), mask_u8.reshape(-1), n_glob, epsilon, entropy_factor,
                       critic_loss_factor, learning_rate, clip_grad_norm, states_unchanged=ep > 0)
        return self.losses()

    def timing(self, enable):
        ms, n = ctypes.c_float(), ctypes.c_int()
        check(lib.evm_ppo_timing(self._h, 1 if enable else 0, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value
