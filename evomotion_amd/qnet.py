"""Twin Q networks of SAC on the device (evm_q_*, csrc/q_kernels.hip): critic_1 / critic_2, their target networks, the
critics' mse update and the soft target update of SoftActorCriticAgent::train (soft_actor_critic.cpp:100-127,166-168)."""
import ctypes

import torch

from ._lib import lib, check

PARAMS, GRADS, EXP_AVG, EXP_AVG_SQ = 0, 1, 2, 3
CRITIC_1, CRITIC_2, TARGET_1, TARGET_2 = 0, 1, 2, 3


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class FusedTwinQ:
    def __init__(self, state_dim, action_dim, hidden_size=256, max_rows=4096, device=0):
        self.device = torch.device("cuda", device)
        self.S, self.A, self.max_rows = state_dim, action_dim, int(max_rows)
        self._h = ctypes.c_void_p()
        torch.cuda.set_device(self.device)
        check(lib.evm_q_create(state_dim, action_dim, hidden_size, self.max_rows, device, ctypes.byref(self._h)))
        n = ctypes.c_size_t()
        check(lib.evm_q_param_count(self._h, ctypes.byref(n)))
        self.n_params = n.value
        self._loss = torch.zeros(2, device=self.device, dtype=torch.float64)

    def close(self):
        if self._h:
            lib.evm_q_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- parameters ---------------------------------------------------------------------------------------------
    def load_vector(self, what, net, flat):
        flat = flat.to(self.device, torch.float32).contiguous()
        assert flat.numel() == self.n_params
        check(lib.evm_q_copy(self._h, what, net, 1, _ptr(flat), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    def load_module(self, net, module):
        self.load_vector(PARAMS, net, torch.cat([p.detach().reshape(-1).float() for p in module.parameters()]))

    def vector(self, what, net, out=None):
        """copy of a trainer vector; `out` (flat fp32 device tensor) is written asynchronously on the current stream"""
        if out is None:
            out = torch.empty(self.n_params, device=self.device)
        check(lib.evm_q_copy(self._h, what, net, 0, _ptr(out), self._stream()))
        return out

    def adam_step(self, net, set_step=-1):
        s = ctypes.c_int()
        check(lib.evm_q_adam_step(self._h, net, set_step, ctypes.byref(s)))
        return s.value

    # ---- compute ------------------------------------------------------------------------------------------------
    def forward(self, nets, states, actions, out=None):
        """Q values of the networks in `nets` (iterable of indices): dict net -> [rows] tensor"""
        rows = states.shape[0]
        assert states.is_cuda and actions.is_cuda and states.is_contiguous() and actions.is_contiguous()
        assert states.shape[1] == self.S and actions.shape[1] == self.A and rows <= self.max_rows
        mask = 0
        res, arr = {}, (ctypes.c_void_p * 4)()
        for i in nets:
            mask |= 1 << i
            res[i] = out[i] if out is not None else torch.empty(rows, device=self.device)
            arr[i] = res[i].data_ptr()
        check(lib.evm_q_forward(self._h, mask, rows, _ptr(states), _ptr(actions), arr, self._stream()))
        return res

    def grads(self, states, actions, target_q):
        rows = states.shape[0]
        assert target_q.is_contiguous() and target_q.numel() == rows and rows <= self.max_rows
        check(lib.evm_q_grads(self._h, rows, _ptr(states), _ptr(actions), _ptr(target_q), self._stream()))

    def apply(self, learning_rate):
        check(lib.evm_q_apply(self._h, learning_rate, self._stream()))

    def soft_update(self, tau):
        check(lib.evm_q_soft_update(self._h, tau, self._stream()))

    def action_grad(self, states, actions, qmin=None, dqda=None):
        """min(critic_1, critic_2)(states, actions) [rows] and d(-mean(min q)) / d actions [rows, A]"""
        rows = states.shape[0]
        qmin = torch.empty(rows, device=self.device) if qmin is None else qmin
        dqda = torch.empty(rows, self.A, device=self.device) if dqda is None else dqda
        check(lib.evm_q_action_grad(self._h, rows, _ptr(states), _ptr(actions), _ptr(qmin), _ptr(dqda), self._stream()))
        return qmin, dqda

    def losses(self):
        """device tensor [2] (float64): the critics' mse losses of the last grads()"""
        check(lib.evm_q_losses(self._h, _ptr(self._loss), self._stream()))
        return self._loss


def sac_sample(mu, sigma, u, action=None, logp_sum=None):
    """truncated_normal_sample(mu, sigma, -1, 1; u) and the summed log-pdf of the sample (device kernels)"""
    rows, A = mu.shape
    action = torch.empty_like(mu) if action is None else action
    logp_sum = torch.empty(rows, device=mu.device) if logp_sum is None else logp_sum
    stream = ctypes.c_void_p(torch.cuda.current_stream(mu.device).cuda_stream)
    check(lib.evm_sac_sample(rows, A, _ptr(mu), _ptr(sigma), _ptr(u), _ptr(action), _ptr(logp_sum), stream))
    return action, logp_sum


def sac_actor_grad(mu, sigma, u, dqda, log_alpha, dmu=None, dsigma=None):
    """d mean(alpha * logp_sum - min q) / d (mu, sigma) through the reparameterised sample"""
    rows, A = mu.shape
    dmu = torch.empty_like(mu) if dmu is None else dmu
    dsigma = torch.empty_like(mu) if dsigma is None else dsigma
    stream = ctypes.c_void_p(torch.cuda.current_stream(mu.device).cuda_stream)
    check(lib.evm_sac_actor_grad(rows, A, _ptr(mu), _ptr(sigma), _ptr(u), _ptr(dqda), _ptr(log_alpha), _ptr(dmu), _ptr(dsigma), stream))
    return dmu, dsigma


def sac_target_q(rewards, done, tq1, tq2, next_logp, log_alpha, gamma, out=None):
    """r + (1 - done) * gamma * (min(tq1, tq2) - alpha * sum_a next_logp)"""
    rows, A = next_logp.shape
    out = torch.empty(rows, device=rewards.device) if out is None else out
    stream = ctypes.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream)
    check(lib.evm_sac_target_q(rows, A, _ptr(rewards), _ptr(done), _ptr(tq1), _ptr(tq2), _ptr(next_logp), _ptr(log_alpha), gamma, _ptr(out), stream))
    return out


def sac_entropy_step(logp_sum, qmin, target_entropy, learning_rate, log_alpha, adam_state, adam_step, losses):
    """one Adam step of log_alpha (in place); losses [2] <- actor loss, entropy loss"""
    stream = ctypes.c_void_p(torch.cuda.current_stream(logp_sum.device).cuda_stream)
    check(lib.evm_sac_entropy_step(logp_sum.numel(), _ptr(logp_sum), _ptr(qmin), target_entropy, learning_rate, _ptr(log_alpha),
                                   _ptr(adam_state), _ptr(adam_step), _ptr(losses), stream))
