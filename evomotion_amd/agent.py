"""Agent side of the rollout path — mirrors of the reference's evo_motion_networks pieces that the PPO loop touches.

  ActorModule / CriticModule / init_weights   evo_motion_networks/src/networks/actor.cpp:9-48, critic.cpp:8-35, init.cpp:7-21
  truncated normal log-pdf / entropy          evo_motion_networks/src/functions.cpp:53-68,113-128
  PpoGaeAgent::train                          evo_motion_networks/src/agents/ppo_gae.cpp:117-190  -> evm_ppo_* (ppo.py)
  PpoGaeAgent::act (batched, fused HIP)       evo_motion_networks/src/agents/ppo_gae.cpp:29-45  -> evm_policy_forward
  RandomAgent::act                            evo_motion_networks/src/agents/debug_agents.cpp:28-30

The rollout forward is the fused MFMA kernel behind the C ABI (`FusedActorCritic`); the PPO update is the HIP trainer
behind `evm_ppo_*` (ppo.py), with an RCCL exchange only there.  The PyTorch-autograd restatement of the update that the
tests compare against lives in tests/torch_ref.py, not in the product.
"""
import ctypes
import math

import numpy as np
import torch
from torch import nn

from ._lib import check, lib

SIGMA_MIN, SIGMA_MAX, ALPHA_BETA_BOUND = 1e-6, 1e6, 5.0  # functions.cpp:9-11


def init_weights(m):
    """init.cpp:7-21: xavier_normal(gain 0.1), bias N(0, 0.1), LayerNorm ones/zeros."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight, 1e-1)
        if m.bias is not None:
            nn.init.normal_(m.bias, 0.0, 1e-1)
    elif isinstance(m, nn.LayerNorm) and m.elementwise_affine:
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)


def _trunk(s, h):
    return nn.Sequential(nn.Linear(s, h), nn.Mish(), nn.LayerNorm(h, eps=1e-5), nn.Linear(h, h), nn.Mish(), nn.LayerNorm(h, eps=1e-5))


class ActorModule(nn.Module):
    """Same parameter names as the reference's ActorModule (head.0.weight ... mu.0.weight, sigma.0.weight)."""

    def __init__(self, state_space, action_space, hidden_size):
        super().__init__()
        self.head = _trunk(state_space[0], hidden_size)
        self.mu = nn.Sequential(nn.Linear(hidden_size, action_space[0]), nn.Tanh())
        self.sigma = nn.Sequential(nn.Linear(hidden_size, action_space[0]), nn.Softplus())
        self.apply(init_weights)

    def forward(self, state):
        h = self.head(state)
        return self.mu(h), self.sigma(h)


class CriticModule(nn.Module):
    def __init__(self, state_space, hidden_size):
        super().__init__()
        t = _trunk(state_space[0], hidden_size)
        self.critic = nn.Sequential(*list(t.children()), nn.Linear(hidden_size, 1))
        self.apply(init_weights)

    def forward(self, state):
        return self.critic(state)


def count_parameters(*modules):
    return sum(p.numel() for m in modules for p in m.parameters())


def flat_parameters(module):
    """Flat fp32 vector in named_parameters() order — the layout evm_policy_set_weights expects."""
    return torch.cat([p.detach().reshape(-1).float().cpu() for _, p in module.named_parameters()]).contiguous()


# ---- truncated normal (functions.cpp) --------------------------------------------------------------------
def _phi(z):
    return torch.exp(-0.5 * torch.pow(z, 2.0)) / math.sqrt(2.0 * math.pi)


def _theta(x):
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _alpha_beta(mu, sigma, lo, hi):
    s = torch.clamp(sigma, SIGMA_MIN, SIGMA_MAX)
    a = torch.clamp((lo - mu) / s, -ALPHA_BETA_BOUND, ALPHA_BETA_BOUND)
    b = torch.clamp((hi - mu) / s, -ALPHA_BETA_BOUND, ALPHA_BETA_BOUND)
    return s, a, b


def truncated_normal_log_pdf(x, mu, sigma, lo=-1.0, hi=1.0):
    s, a, b = _alpha_beta(mu, sigma, lo, hi)
    z = _theta(b) - _theta(a)
    return -0.5 * math.log(2.0 * math.pi) - torch.log(s) - 0.5 * torch.pow((x - mu) / s, 2.0) - torch.log(z)


def truncated_normal_entropy(mu, sigma, lo=-1.0, hi=1.0):
    s, a, b = _alpha_beta(mu, sigma, lo, hi)
    z = _theta(b) - _theta(a)
    return torch.log(math.sqrt(2.0 * math.pi * math.e) * s * z) + 0.5 * (a * _phi(a) - b * _phi(b)) / z


def truncated_normal_sample(mu, sigma, lo=-1.0, hi=1.0, u=None):
    s, a, b = _alpha_beta(mu, sigma, lo, hi)
    if u is None:
        u = torch.rand(mu.shape, device=mu.device)
    cdf = torch.clamp(_theta(a) + u * (_theta(b) - _theta(a)), 0.0, 1.0)
    return torch.clamp(math.sqrt(2.0) * torch.erfinv(2.0 * cdf - 1.0) * s + mu, lo, hi)


# ---- fused rollout forward (HIP) ----------------------------------------------------------------------------
def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class FusedActorCritic:
    """Batched PpoGaeAgent::act on the GPU: actor + critic forward, truncated-normal sample and log-pdf."""

    def __init__(self, state_dim, action_dim, hidden_size=256, device=0):
        self.device = torch.device("cuda", device)
        self.S, self.A = state_dim, action_dim
        self._h = ctypes.c_void_p()
        torch.cuda.set_device(self.device)
        check(lib.evm_policy_create(state_dim, action_dim, hidden_size, device, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib.evm_policy_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tile_rows(self, rows):
        """0 = per launch (32-row tiles unless they would leave CUs idle, then 16-row tiles), 16 / 32 = that form always."""
        check(lib.evm_policy_set_tile_rows(self._h, rows))

    def timing_begin(self):
        check(lib.evm_policy_timing_begin(self._h))

    def timing_end(self):
        ms, n = ctypes.c_float(), ctypes.c_int()
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(lib.evm_policy_timing_end(self._h, stream, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def set_actor(self, actor):
        """Actor weights only (SAC has no state-value critic); device-side repack, no host round trip."""
        self.load_modules(actor, None)

    def set_weights(self, actor_flat, critic_flat):
        a = np.ascontiguousarray(actor_flat, np.float32)
        c = np.ascontiguousarray(critic_flat, np.float32)
        self._critic_flat = c
        fp = ctypes.POINTER(ctypes.c_float)
        check(lib.evm_policy_set_weights(self._h, a.ctypes.data_as(fp), a.size, c.ctypes.data_as(fp), c.size))

    def load_modules(self, actor, critic):
        """Weights from (device) modules after an optimiser step: flattened and repacked on the device, asynchronously on
        the current stream (evm_policy_set_weights_device)."""
        flat = lambda m: None if m is None else torch.cat([p.detach().reshape(-1).float() for p in m.parameters()]).to(self.device).contiguous()
        a, c = flat(actor), flat(critic)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(lib.evm_policy_set_weights_device(self._h, _ptr(a), _ptr(c), stream))
        self._keep = (a, c)  # the kernel reads them asynchronously

    def load_th(self, actor_path, critic_path):
        """Weights from `*.th` checkpoints in the reference's format (saver.h:27-39), e.g. the reference's own files."""
        from .checkpoint import load_th
        a, c = load_th(actor_path), load_th(critic_path)
        self.set_weights(torch.cat([t.reshape(-1).float() for t in a.values()]).numpy(),
                         torch.cat([t.reshape(-1).float() for t in c.values()]).numpy())

    def forward(self, obs, uniform=None, seed=0, want_dist=False, actor_only=False, out=None):
        """actor_only: the critic network is not run and `value` is None (SoftActorCriticAgent::act).
        out = (action [n, A], logp [n, A], value [n] or None): contiguous tensors the kernel writes directly."""
        n = obs.shape[0]
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.S
        if out is not None:
            action, logp, value = out
            assert action.is_contiguous() and logp.is_contiguous() and tuple(action.shape) == tuple(logp.shape) == (n, self.A)
            assert value is None or (value.is_contiguous() and value.numel() == n)
            if actor_only:
                value = None
        else:
            action = torch.empty(n, self.A, device=self.device)
            logp = torch.empty(n, self.A, device=self.device)
            value = None if actor_only else torch.empty(n, device=self.device)
        mu = torch.empty(n, self.A, device=self.device) if want_dist else None
        sigma = torch.empty(n, self.A, device=self.device) if want_dist else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(lib.evm_policy_forward(self._h, n, _ptr(obs), _ptr(uniform), seed, _ptr(action), _ptr(logp), _ptr(value),
                                     _ptr(mu), _ptr(sigma), stream))
        return (action, logp, value, mu, sigma) if want_dist else (action, logp, value)


class RandomAgent:
    """RandomAgent (debug_agents.cpp:7-30): act() = 2 * U[0,1)^A - 1 from the global torch generator, everything else a
    no-op.  A 1-D state gives the reference's [A] action (same draws as the reference after torch.manual_seed on the
    CPU, pinned in tests/test_agent_host.py); a [N, S] batch gives [N, A], one generator call for all envs.
    `seed`: use a private generator on `device` instead of the global one (bench.py: one stream per rank)."""

    def __init__(self, action_space, device="cpu", seed=None):
        self.A, self.device = int(action_space[0]), torch.device(device)
        self.generator = None
        if seed is not None:
            self.generator = torch.Generator(device=self.device)
            self.generator.manual_seed(int(seed))

    def act(self, state, reward=None):
        shape = (self.A,) if state.dim() == 1 else (state.shape[0], self.A)
        return 2.0 * torch.rand(*shape, device=self.device, generator=self.generator) - 1.0

    def done(self, state, reward=None):
        pass

    def save(self, output_folder_path):
        pass

    def load(self, input_folder_path):
        pass

    def get_metrics(self):
        return []

    def to(self, device):
        self.device = torch.device(device)
        if self.generator is not None:
            seed = self.generator.initial_seed()
            self.generator = torch.Generator(device=self.device)
            self.generator.manual_seed(seed)
        return self

    def set_eval(self, eval_mode):
        pass

    def count_parameters(self):
        return 0


class VecPpoGaeAgent:
    """Vectorised PpoGaeAgent: fixed-horizon rollouts of N envs instead of a replay buffer of whole episodes.

    Hyper-parameter names follow agent_factory.cpp:137-146.  Transitions emitted while an env is inside reset()
    (valid == 0) are masked out exactly like the reference's padding (done = 1, shifted mask)."""

    def __init__(self, seed, state_space, action_space, hidden_size=256, gamma=0.99, lam=0.95, epsilon=0.2,
                 entropy_factor=0.01, critic_loss_factor=0.5, epoch=8, learning_rate=1e-3, clip_grad_norm=0.5,
                 device=0, horizon=32):
        """PpoGaeAgent::train runs in the HIP trainer (ppo.py, csrc/ppo_kernels.hip), which owns the master weights and the
        Adam state; the torch modules / optimisers below are the initialisation (init.cpp:7-21) and the checkpoint views."""
        torch.manual_seed(seed)
        self.device = torch.device("cuda", device)
        self.actor = ActorModule(state_space, action_space, hidden_size).to(self.device)
        self.critic = CriticModule(state_space, hidden_size).to(self.device)
        self.actor_opt = torch.optim.Adam(self.actor.parameters(), lr=learning_rate)
        self.critic_opt = torch.optim.Adam(self.critic.parameters(), lr=learning_rate)
        self.hp = dict(gamma=gamma, lam=lam, epsilon=epsilon, entropy_factor=entropy_factor,
                       critic_loss_factor=critic_loss_factor, epoch=epoch, clip_grad_norm=clip_grad_norm)
        self.fused = FusedActorCritic(state_space[0], action_space[0], hidden_size, device)
        self.fused.load_modules(self.actor, self.critic)
        self.horizon = horizon
        self.seed = seed
        # weights share `seed` across ranks; the rollout's exploration noise must not (data-parallel replicas would explore alike)
        self.noise_seed = seed
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.noise_seed = (seed ^ (torch.distributed.get_rank() * 0x9E3779B97F4A7C15)) & 0x7FFFFFFF
        self._buf = None
        self.learning_rate = learning_rate
        self._trainer = None
        self._modules_stale = False

    def _get_trainer(self, rows):
        from .ppo import FusedPpoTrainer
        if self._trainer is None or self._trainer.max_rows < rows:
            if self._trainer is not None:
                self.sync_modules()
                self._trainer.close()
            self._trainer = FusedPpoTrainer(self.fused, rows)
            self._push_to_trainer()
        return self._trainer

    def _push_to_trainer(self):
        """modules + torch optimiser states -> the trainer (construction, load())"""
        from .ppo import EXP_AVG, EXP_AVG_SQ
        tr = self._trainer
        tr.set_modules(self.actor, self.critic, reset_optimizer=True)
        for net, mod, opt in ((0, self.actor, self.actor_opt), (1, self.critic, self.critic_opt)):
            ps = list(mod.parameters())
            have = [p in opt.state and "exp_avg" in opt.state[p] for p in ps]
            if any(have):
                # per parameter: an archive (the reference's, or one written before a parameter took its first step) may hold
                # state for some parameters only — those without get zero moments; the trainer has one step count per network,
                # the largest of the archive's (torch keeps one per parameter, all equal after ordinary training)
                zero = lambda p: torch.zeros(p.numel(), device=p.device)
                tr.load_vector(EXP_AVG, net, torch.cat([opt.state[p]["exp_avg"].reshape(-1) if h else zero(p) for p, h in zip(ps, have)]))
                tr.load_vector(EXP_AVG_SQ, net, torch.cat([opt.state[p]["exp_avg_sq"].reshape(-1) if h else zero(p) for p, h in zip(ps, have)]))
                tr.adam_step(net, max(int(float(opt.state[p]["step"])) for p, h in zip(ps, have) if h))

    def sync_modules(self):
        """the trainer's weights and Adam state -> the torch modules / optimisers (save(), evaluation in torch)"""
        if self._trainer is None or not self._modules_stale:
            return
        from .ppo import EXP_AVG, EXP_AVG_SQ
        tr = self._trainer
        tr.params_into(self.actor, self.critic)
        for net, mod, opt in ((0, self.actor, self.actor_opt), (1, self.critic, self.critic_opt)):
            m, v, step = tr.vector(EXP_AVG, net), tr.vector(EXP_AVG_SQ, net), tr.adam_step(net)
            o = 0
            for p in mod.parameters():
                n = p.numel()
                opt.state[p] = dict(step=torch.tensor(float(step)), exp_avg=m[o:o + n].view_as(p).clone(),
                                    exp_avg_sq=v[o:o + n].view_as(p).clone())
                o += n
        self._modules_stale = False

    def count_parameters(self):
        return count_parameters(self.actor, self.critic)

    def _adam_states(self, module, opt):
        out = []
        for p in module.parameters():
            st = opt.state.get(p)
            out.append(None if not st or "exp_avg" not in st else (int(float(st["step"])), st["exp_avg"], st["exp_avg_sq"]))
        return out

    def save(self, output_folder_path):
        """PpoGaeAgent::save (ppo_gae.cpp:192-197): actor.th, actor_optimizer.th, critic.th, critic_optimizer.th in the
        reference's formats (checkpoint.py) — the folder loads in the reference's PpoGaeAgent::load and the other way round."""
        from .checkpoint import save_adam_th, save_th
        import os
        self.sync_modules()
        save_th(self.actor, os.path.join(output_folder_path, "actor.th"))
        save_th(self.critic, os.path.join(output_folder_path, "critic.th"))
        save_adam_th(os.path.join(output_folder_path, "actor_optimizer.th"), self._adam_states(self.actor, self.actor_opt), self.learning_rate)
        save_adam_th(os.path.join(output_folder_path, "critic_optimizer.th"), self._adam_states(self.critic, self.critic_opt), self.learning_rate)

    def load(self, input_folder_path):
        """PpoGaeAgent::load (ppo_gae.cpp:199-204): the four archives; like the reference's load_torch a missing file is an
        error ("Could not find")."""
        from .checkpoint import load_adam_th, load_into
        import os
        load_into(self.actor, os.path.join(input_folder_path, "actor.th"))
        load_into(self.critic, os.path.join(input_folder_path, "critic.th"))
        for mod, opt, name in ((self.actor, self.actor_opt, "actor_optimizer.th"), (self.critic, self.critic_opt, "critic_optimizer.th")):
            states, options = load_adam_th(os.path.join(input_folder_path, name))
            # the trainer's Adam is torch's default one (betas 0.9 / 0.999, eps 1e-8, no weight decay, no amsgrad — what the
            # reference constructs, ppo_gae.cpp:22-25); an archive that says otherwise cannot be continued faithfully
            if (tuple(round(b, 6) for b in options["betas"]) != (0.9, 0.999) or abs(options["eps"] - 1e-8) > 1e-12
                    or options["weight_decay"] != 0.0 or options["amsgrad"]):
                raise RuntimeError("%s: Adam options %r differ from the trainer's fixed defaults" % (name, options))
            ps = list(mod.parameters())
            if len(states) != len(ps):
                raise RuntimeError("%s holds %d parameters, the module has %d" % (name, len(states), len(ps)))
            opt.state.clear()
            for p, st in zip(ps, states):
                if st is not None:
                    opt.state[p] = dict(step=torch.tensor(float(st[0])), exp_avg=st[1].to(self.device).reshape(p.shape).clone(),
                                        exp_avg_sq=st[2].to(self.device).reshape(p.shape).clone())
            self.learning_rate = options["lr"]   # AdamOptions travel with the archive (torch::optim::Adam::load)
            for g in opt.param_groups:
                g["lr"] = options["lr"]
        self.fused.load_modules(self.actor, self.critic)
        if self._trainer is not None:
            self._push_to_trainer()
        self._modules_stale = False

    def rollout(self, env, last=None):
        """`horizon` calls of policy forward + evm_env_step_autoreset; everything stays on the device."""
        T, N, S, A = self.horizon, env.n_envs, env.state_dim, env.action_dim
        if self._buf is None:
            z = lambda *s, **k: torch.zeros(*s, device=self.device, **k)
            # states[t] = the observation acted on at step t; the env writes observation t + 1 straight into states[t + 1]
            self._buf = dict(states_all=z(T + 1, N, S), actions=z(T, N, A), logp=z(T, N, A), values=z(T, N), rewards=z(T, N),
                             done=z(T, N), valid=z(T, N), next_values=z(T, N), done_u8=z(T, N, dtype=torch.uint8),
                             valid_u8=z(T, N, dtype=torch.uint8), scratch=(z(N, A), z(N, A), z(N)))
        b = self._buf
        b["states"] = b["states_all"][:T]
        b["states_all"][0].copy_(env.obs)  # one copy per horizon
        for t in range(T):
            # two launches per step, no copies: the kernels read and write the rows of the rollout buffer
            self.fused.forward(b["states_all"][t], seed=self.noise_seed, out=(b["actions"][t], b["logp"][t], b["values"][t]))
            env.step_autoreset(b["actions"][t], reward_out=b["rewards"][t], done_out=b["done_u8"][t], valid_out=b["valid_u8"][t],
                               obs_out=b["states_all"][t + 1])
        b["done"].copy_(b["done_u8"])
        b["valid"].copy_(b["valid_u8"])
        _, _, last_v = self.fused.forward(b["states_all"][T], seed=self.noise_seed, out=b["scratch"])
        b["next_values"][:-1] = b["values"][1:]
        b["next_values"][-1] = last_v
        return b

    def update(self):
        """PpoGaeAgent::train (ppo_gae.cpp:117-190) on the device.  Only do_step transitions (valid == 1) are trained on:
        settle calls (0) and reset()'s own emission (2) are not transitions."""
        b = self._buf
        T, N = b["rewards"].shape
        tr = self._get_trainer(T * N)
        mask = (b["valid_u8"] == 1).to(torch.uint8)
        hp = self.hp
        out = tr.train(b["states"], b["actions"], b["rewards"], b["done_u8"], b["logp"], b["values"], b["next_values"], mask,
                       hp["gamma"], hp["lam"], hp["epsilon"], hp["entropy_factor"], hp["critic_loss_factor"], hp["epoch"],
                       self.learning_rate, hp["clip_grad_norm"])
        self._modules_stale = True
        return out


# ---- the reference's Agent surface for PPO: act / done / check_train over whole episodes ------------------------------------
class TrajectoryReplayBuffer:
    """TrajectoryReplayBuffer (evo_motion_networks/src/replay_buffer.cpp:73-138,176-189): a FIFO of whole trajectories.

      new_trajectory()   push an empty trajectory, evict from the front beyond `size` trajectories            (:105-110)
      add / update_last  append a step to / rewrite (reward, done, next_value) of the newest trajectory's last step (:112-123,176-186)
      enough_trajectory  at least batch_size trajectories of more than one step                               (:139-146)
      sample             the trajectories of more than one step, ALL BUT THE LAST of them shuffled, the first batch_size
                         taken — fewer when there are not that many                                             (:80-98)

    A step is a dict(state, action, reward, done, log_prob, curr_value, next_value) of device tensors / Python scalars
    (ppo_episode_step, replay_buffer.h:22-30).  The reference shuffles with its own std::mt19937(seed) + std::shuffle
    (replay_buffer.cpp:66,83): stdrandom.py restates both (held to g++ / libstdc++ and to the draws recorded from the compiled
    reference, tests/test_stdrandom.py), so the default draws ARE the reference's for the same seed; `shuffle(index_list)` stays
    a hook for tests."""

    def __init__(self, size, seed):
        from .stdrandom import Mt19937, std_shuffle
        self.size, self.memory = int(size), []
        self._rng = Mt19937(seed)
        self.shuffle = lambda index: std_shuffle(list(index), self._rng)

    def empty(self):
        return not self.memory

    def trajectory_empty(self):
        return self.empty() or not self.memory[-1]

    def new_trajectory(self):
        self.memory.append([])
        while len(self.memory) > self.size:
            self.memory.pop(0)
        return self.memory[-1]

    def push_closed(self, trajectory):
        """a finished trajectory that was filled outside the buffer (several environments at once: PpoGaeAgent with n_envs > 1)
        becomes the newest entry — where the reference's single environment would have left it — with the same FIFO eviction"""
        self.memory.append(trajectory)
        while len(self.memory) > self.size:
            self.memory.pop(0)

    def add(self, step, trajectory=None):
        (self.memory[-1] if trajectory is None else trajectory).append(step)

    def update_last(self, reward, done, next_value, trajectory=None):
        st = (self.memory[-1] if trajectory is None else trajectory)[-1]
        st["reward"], st["done"], st["next_value"] = reward, done, next_value

    def _filtered(self):
        return [t for t in self.memory if len(t) > 1]

    def enough_trajectory(self, batch_size):
        return len(self._filtered()) >= batch_size

    def sample(self, batch_size):
        filtered = self._filtered()
        index = self.shuffle(list(range(len(filtered) - 1)))
        return [filtered[i] for i in index[:batch_size]]


class PpoGaeAgent(VecPpoGaeAgent):
    """PpoGaeAgent with the reference's Agent surface (evo_motion_networks/include/evo_motion_networks/agent.h:16-35,
    src/agents/ppo_gae.cpp:29-115): act(state, reward) -> action, done(state, reward), whole episodes in a
    TrajectoryReplayBuffer, a train() call every `train_every` finished EPISODES on `batch_size` trajectories padded to the
    longest one (done = 1 in the padding, the shifted mask of ppo_gae.cpp:127-132).  The forward pass is the fused HIP kernel,
    train() the HIP trainer behind evm_ppo_* (time-major [T][B] with the same mask) — this class is the bookkeeping around
    them.  n_envs = 1 is the reference call for call.  With n_envs > 1 (act / done take a batch and an env index) every
    environment fills an open trajectory of its own OUTSIDE the buffer, which enters it when done() closes it: the buffer
    then holds finished episodes only, so that `sample()` — whose "all but the last one" rule is right for one environment,
    where the last entry is the episode that just ended — can never pick an episode that is still running (its last step would
    carry the placeholders reward = 0, done = False, next_value = curr_value) and eviction can never drop a trajectory an
    environment still appends to.  The fixed-horizon VecPpoGaeAgent.rollout()/update() stays the fast path for thousands of
    environments.

    `reward` of act() is the reward of the PREVIOUS transition (ppo_gae.cpp:38); done() gets the terminal state before the
    environment is reset (src/train.cpp:64-65)."""

    def __init__(self, seed, state_space, action_space, hidden_size=256, gamma=0.99, lam=0.95, epsilon=0.2, entropy_factor=0.01,
                 critic_loss_factor=0.5, epoch=8, batch_size=32, train_every=8, replay_buffer_size=1024, learning_rate=1e-3,
                 clip_grad_norm=0.5, device=0, n_envs=1):
        super().__init__(seed, state_space, action_space, hidden_size, gamma, lam, epsilon, entropy_factor, critic_loss_factor, epoch,
                         learning_rate, clip_grad_norm, device, horizon=1)
        self.batch_size, self.train_every, self.n_envs = int(batch_size), int(train_every), int(n_envs)
        self.replay_buffer = TrajectoryReplayBuffer(replay_buffer_size, seed)
        self.curr_train_step = self.global_curr_step = 0
        self.curr_episode_step = [0] * self.n_envs
        self._open = [None] * self.n_envs      # the open trajectory of every env (n_envs = 1: a list inside replay_buffer.memory)
        self._act_calls = 0
        self.S, self.A = int(state_space[0]), int(action_space[0])
        from .metrics import LossMeter
        # ppo_gae.cpp:23-24: the three meters of get_metrics(), windows of 64
        self.actor_loss_meter, self.critic_loss_meter = LossMeter("actor_loss", 64), LossMeter("critic_loss", 64)
        self.episode_steps_meter = LossMeter("steps", 64)

    # -- Agent interface ---------------------------------------------------------------------------------------------------
    def _forward(self, states, uniform=None):
        obs = states.to(device=self.device, dtype=torch.float32).reshape(-1, self.S).contiguous()
        if uniform is not None:
            uniform = uniform.to(device=self.device, dtype=torch.float32).reshape(-1, self.A).contiguous()
        self._act_calls += 1
        return obs, self.fused.forward(obs, uniform=uniform, seed=(self.noise_seed + 7919 * self._act_calls) & 0x7FFFFFFF)

    def act(self, state, reward, uniform=None):
        """state [S] (one env, like the reference) or [n_envs, S]; reward a float or [n_envs].  uniform: the U[0,1) draws
        of truncated_normal_sample (the reference's at::rand), else the kernel's counter-based generator."""
        single = state.dim() == 1
        obs, (action, logp, value) = self._forward(state, uniform)
        rewards = [float(reward)] if single else [float(r) for r in torch.as_tensor(reward).reshape(-1).tolist()]
        assert obs.shape[0] == self.n_envs == len(rewards)
        for e in range(self.n_envs):
            if self._open[e] is None:           # `if (replay_buffer.empty()) new_trajectory()` + the one done() opens
                self._open[e] = self.replay_buffer.new_trajectory() if self.n_envs == 1 else []
            traj = self._open[e]
            if traj:
                self.replay_buffer.update_last(rewards[e], False, value[e].clone(), traj)
            self.replay_buffer.add(dict(state=obs[e].clone(), action=action[e].clone(), reward=0.0, done=False, log_prob=logp[e].clone(),
                                        curr_value=value[e].clone(), next_value=value[e].clone()), traj)
            self.curr_episode_step[e] += 1
        return action[0].clone() if single else action.clone()

    def done(self, state, reward, env=0):
        """the episode of environment `env` has ended in `state` (its terminal observation) with `reward`"""
        obs, (_, _, value) = self._forward(state.reshape(1, -1), None)
        traj = self._open[env]
        self.replay_buffer.update_last(float(reward), True, value[0].clone(), traj)
        if self.n_envs > 1:
            self.replay_buffer.push_closed(traj)    # now the newest entry, as the reference's just-finished episode is
        self.check_train()
        self._open[env] = self.replay_buffer.new_trajectory() if self.n_envs == 1 else None   # (n_envs > 1: opened by its next act())
        self.global_curr_step += 1
        self.episode_steps_meter.add(float(self.curr_episode_step[env]))
        self.curr_episode_step[env] = 0

    def check_train(self):
        if not (self.global_curr_step % self.train_every == self.train_every - 1 and self.replay_buffer.enough_trajectory(self.batch_size)):
            return None
        episodes = self.replay_buffer.sample(self.batch_size)
        B, T = len(episodes), max(len(t) for t in episodes)
        z = lambda *s, **k: torch.zeros(*s, device=self.device, **k)
        # time-major [T][B]; padding: zeros, done = 1 (ppo_gae.cpp:93-103)
        states, actions, logp = z(T, B, self.S), z(T, B, self.A), z(T, B, self.A)
        rewards, values, next_values = z(T, B), z(T, B), z(T, B)
        done = torch.ones(T, B, device=self.device, dtype=torch.uint8)
        for b, traj in enumerate(episodes):
            L = len(traj)
            states[:L, b] = torch.stack([s["state"] for s in traj])
            actions[:L, b] = torch.stack([s["action"] for s in traj])
            logp[:L, b] = torch.stack([s["log_prob"] for s in traj])
            values[:L, b] = torch.stack([s["curr_value"].reshape(()) for s in traj])
            next_values[:L, b] = torch.stack([s["next_value"].reshape(()) for s in traj])
            rewards[:L, b] = torch.tensor([s["reward"] for s in traj], device=self.device)
            done[:L, b] = torch.tensor([1 if s["done"] else 0 for s in traj], device=self.device, dtype=torch.uint8)
        # mask[t] = 1 at t = 0, else 1 - done[t - 1] (ppo_gae.cpp:127-132)
        mask = torch.cat([torch.ones(1, B, device=self.device, dtype=torch.uint8), 1 - done[:-1]], 0).contiguous()
        tr = self._get_trainer(T * B)
        hp = self.hp
        out = tr.train(states, actions, rewards, done, logp, values, next_values, mask, hp["gamma"], hp["lam"], hp["epsilon"],
                       hp["entropy_factor"], hp["critic_loss_factor"], hp["epoch"], self.learning_rate, hp["clip_grad_norm"],
                       loss_hook=self._meter_losses)
        self._modules_stale = True
        self.curr_train_step += 1
        return out

    def _meter_losses(self, actor_loss, critic_loss):
        """after every epoch of train(), like ppo_gae.cpp:185-186"""
        self.actor_loss_meter.add(actor_loss)
        self.critic_loss_meter.add(critic_loss)

    def get_metrics(self):
        """ppo_gae.cpp:205-207"""
        return [self.actor_loss_meter, self.critic_loss_meter, self.episode_steps_meter]

    def to(self, device):
        return self

    def set_eval(self, eval_mode):
        pass
