"""Soft actor-critic rows of the path — mirrors of the reference's evo_motion_networks pieces.

  QNetworkModule                 evo_motion_networks/src/networks/q_net.cpp:8-43
  EntropyParameter               evo_motion_networks/src/networks/entropy.cpp:7-15
  hard_update / soft_update      evo_motion_networks/src/functions.cpp:151-171
  SoftActorCriticAgent::train    evo_motion_networks/src/agents/soft_actor_critic.cpp:93-170
  SoftActorCriticAgent::act      soft_actor_critic.cpp:47-62 (batched: fused HIP actor forward + device replay ring)
  ReplayBuffer                   evo_motion_networks/src/replay_buffer.cpp:16-52,146-153 -> evm_replay_* (HIP)

The update runs on the device (qnet.py / csrc/q_kernels.hip and the PPO trainer's actor kernels); the module mirrors here
carry the reference's parameter names for initialisation and checkpoints.  The PyTorch-autograd restatement of train() that
the tests pin to the reference's golden call (tests/golden/sac_golden.txt) lives in tests/torch_ref.py.
"""
import math

import numpy as np

import torch
from torch import nn

from .agent import ActorModule, init_weights


class QNetworkModule(nn.Module):
    """q_network.0 … q_network.9 as in the reference: three Linear→Mish→LayerNorm blocks and a Linear(hidden, 1)."""

    def __init__(self, state_space, action_space, hidden_size):
        super().__init__()
        h = hidden_size
        self.q_network = nn.Sequential(
            nn.Linear(state_space[0] + action_space[0], h), nn.Mish(), nn.LayerNorm(h, eps=1e-5),
            nn.Linear(h, h), nn.Mish(), nn.LayerNorm(h, eps=1e-5),
            nn.Linear(h, h), nn.Mish(), nn.LayerNorm(h, eps=1e-5),
            nn.Linear(h, 1))
        self.apply(init_weights)

    def forward(self, state, action):
        return self.q_network(torch.cat([state, action], -1))


class EntropyParameter(nn.Module):
    def __init__(self, initial_alpha=1.0, nb_parameters=1):
        super().__init__()
        self.log_alpha = nn.Parameter(torch.full((nb_parameters,), math.log(initial_alpha)))

    def alpha(self):
        return self.log_alpha.exp()


def hard_update(to, frm):
    with torch.no_grad():
        for (_, t), (_, f) in zip(to.named_parameters(), frm.named_parameters()):
            t.copy_(f)


def soft_update(to, frm, tau):
    """to <- tau * from + (1 - tau) * to, with the reference's double-precision (1.0 - tau) (functions.cpp:169)."""
    with torch.no_grad():
        ts, fs = [t for _, t in to.named_parameters()], [f for _, f in frm.named_parameters()]
        # two multi-tensor launches instead of four per parameter: t <- (1 - tau) t, then t <- t + tau f
        torch._foreach_mul_(ts, 1.0 - tau)
        torch._foreach_add_(ts, fs, alpha=tau)


def _rank_seed(seed):
    """Equal seeds on every rank give equal initial weights (needed) — and would give every data-parallel replica the same
    uniform draws for exploration and replay sampling.  Mix the rank into the seed used for those."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return (seed ^ (torch.distributed.get_rank() * 0x9E3779B97F4A7C15)) & 0x7FFFFFFF
    return seed


class VecSacAgent:
    """Vectorised SoftActorCriticAgent: one act() for all N envs (fused HIP actor forward), a device replay ring, and the
    reference's train() on batches drawn from it.  Parameter names follow agent_factory.cpp:112-119.

    `replay_buffer_size` counts rollout steps (ring slots of N envs each); `train_every` counts rollout steps, like the
    reference's global step counter (soft_actor_critic.cpp:64-65)."""

    def __init__(self, seed, state_space, action_space, actor_hidden_size=256, critic_hidden_size=256, batch_size=256,
                 epoch=1, learning_rate=1e-3, gamma=0.99, tau=0.005, replay_buffer_size=1024, train_every=4, n_envs=4096,
                 device=0, use_graph=True):
        """SoftActorCriticAgent::train runs on the device — target actions from the fused policy kernel, the twin critics,
        their targets, the critics' mse update and the soft update in the HIP Q trainer (qnet.py, csrc/q_kernels.hip), the
        actor step (forward, reparameterised sample, action gradient through the critics, backward, Adam) on the PPO
        trainer's actor kernels, the entropy parameter's step in its own kernel."""
        from .agent import FusedActorCritic
        from .replay import ReplayRing
        torch.manual_seed(seed)
        self.device = torch.device("cuda", device)
        mk_q = lambda: QNetworkModule(state_space, action_space, critic_hidden_size).to(self.device)
        self.actor = ActorModule(state_space, action_space, actor_hidden_size).to(self.device)
        self.critic_1, self.critic_2, self.target_critic_1, self.target_critic_2 = mk_q(), mk_q(), mk_q(), mk_q()
        self.entropy = EntropyParameter(1.0, 1).to(self.device)
        hard_update(self.target_critic_1, self.critic_1)
        hard_update(self.target_critic_2, self.critic_2)
        # the update is ~10^3 small launches; it is captured once into a HIP graph and replayed (capturable Adam keeps
        # its step counters on the device).  Data-parallel runs keep eager mode: the gradient all-reduce sits between
        # backward and step.
        self.use_graph = use_graph and not (torch.distributed.is_available() and torch.distributed.is_initialized()
                                            and torch.distributed.get_world_size() > 1)
        adam = lambda m: torch.optim.Adam(m.parameters(), lr=learning_rate, capturable=self.use_graph)
        self.actor_opt, self.critic_1_opt, self.critic_2_opt, self.entropy_opt = adam(self.actor), adam(self.critic_1), adam(self.critic_2), adam(self.entropy)
        self.target_entropy = -float(action_space[0])
        self.gamma, self.tau, self.batch_size, self.epoch, self.train_every = gamma, tau, batch_size, epoch, train_every
        self.fused = FusedActorCritic(state_space[0], action_space[0], actor_hidden_size, device)
        self.fused.set_actor(self.actor)
        self.replay = ReplayRing(replay_buffer_size, n_envs, state_space[0], action_space[0], device)
        self._prev = torch.empty(n_envs, state_space[0], device=self.device)
        self._act = (torch.empty(n_envs, action_space[0], device=self.device), torch.empty(n_envs, action_space[0], device=self.device), None)
        self.seed, self.global_step, self.train_steps = seed, 0, 0
        self.noise_seed = _rank_seed(seed)  # weights share `seed` across ranks, exploration noise and replay draws do not
        self.last_losses = None
        self._graph = None
        self._stored_margin, self._pushes_since_check = 0, 0  # has_enough(): see there
        B, S, A = batch_size, state_space[0], action_space[0]
        z = lambda *sh: torch.zeros(*sh, device=self.device)
        self._batch = (z(B, S), z(B, A), z(B), z(B), z(B, S))
        self.learning_rate = learning_rate
        from .qnet import FusedTwinQ
        self.twinq = FusedTwinQ(S, A, critic_hidden_size, B, device)
        self._push_critics()
        self._tq_out = {2: z(B), 3: z(B)}
        self._next_act = (z(B, A), z(B, A), None)
        # the actor step runs on the PPO trainer's actor kernels (same ActorModule); its critic slot is unused
        from .ppo import FusedPpoTrainer
        self._actor_tr = FusedPpoTrainer(self.fused, B)
        self._critic_dummy = z(self._actor_tr.n_params[1])
        self._abuf = dict(mu=z(B, A), sigma=z(B, A), action=z(B, A), logp=z(B), qmin=z(B), dqda=z(B, A), dmu=z(B, A), dsigma=z(B, A),
                          target_q=z(B), losses=z(2))
        self._ent_state = z(2)                                                   # Adam exp_avg, exp_avg_sq of log_alpha
        self._ent_step = torch.zeros(1, device=self.device, dtype=torch.int32)
        self._push_actor()

    def _push_actor(self):
        """the actor module -> the trainer, which owns its master weights and Adam state from here on (sync_modules() copies
        them back)"""
        flat = torch.cat([p.detach().reshape(-1) for p in self.actor.parameters()]).contiguous()
        self._actor_tr.set_flat(flat, self._critic_dummy, reset_optimizer=True)
        torch.cuda.current_stream(self.device).synchronize()

    def _push_critics(self):
        """critic / target modules -> the HIP trainer, which owns their weights from here on"""
        for i, m in enumerate((self.critic_1, self.critic_2, self.target_critic_1, self.target_critic_2)):
            self.twinq.load_module(i, m)

    def sync_modules(self):
        """the trainers' actor / critic / target weights -> the torch modules (save(), evaluation)"""
        from .qnet import PARAMS
        from .ppo import PARAMS as PP, ACTOR
        flat = self._actor_tr.vector(PP, ACTOR)
        o = 0
        with torch.no_grad():
            for p in self.actor.parameters():
                p.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
        for i, m in enumerate((self.critic_1, self.critic_2, self.target_critic_1, self.target_critic_2)):
            flat = self.twinq.vector(PARAMS, i)
            o = 0
            with torch.no_grad():
                for p in m.parameters():
                    p.copy_(flat[o:o + p.numel()].view_as(p))
                    o += p.numel()

    def _train_once_hip(self, u_next=None, u_curr=None):
        """SoftActorCriticAgent::train (soft_actor_critic.cpp:93-170) on the device; u_next / u_curr override the two
        uniform draws (tests)"""
        from .ppo import GRADS as PGRADS, ACTOR
        from .qnet import GRADS, sac_actor_grad, sac_entropy_step, sac_sample, sac_target_q
        s, a, r, d, n = self._batch
        B, A = s.shape[0], self.fused.A
        ab = self._abuf
        log_alpha = self.entropy.log_alpha.detach()
        dist_on = (not self.use_graph and torch.distributed.is_available() and torch.distributed.is_initialized()
                   and torch.distributed.get_world_size() > 1)

        def all_reduce_mean(g):
            if torch.distributed.get_backend() != "nccl":
                h = g.cpu(); torch.distributed.all_reduce(h); g.copy_(h)
            else:
                torch.distributed.all_reduce(g)
            return g / torch.distributed.get_world_size()

        # targets (:100-116): next action ~ actor(next state) from the fused policy kernel, twin target Q, entropy term
        u_next = torch.rand(B, A, device=self.device) if u_next is None else u_next
        next_action, next_logp, _ = self.fused.forward(n, uniform=u_next, actor_only=True, out=self._next_act)
        tq = self.twinq.forward((2, 3), n, next_action, out=self._tq_out)
        sac_target_q(r, d, tq[2], tq[3], next_logp, log_alpha, self.gamma, ab["target_q"])
        # critics (:118-127)
        self.twinq.grads(s, a, ab["target_q"])
        if dist_on:
            for i in (0, 1):
                self.twinq.load_vector(GRADS, i, all_reduce_mean(self.twinq.vector(GRADS, i)))
        self.twinq.apply(self.learning_rate)
        # actor (:129-153): forward, reparameterised sample, action gradient through the twin critics, loss gradient,
        # backward, Adam
        u_curr = torch.rand(B, A, device=self.device) if u_curr is None else u_curr
        self._actor_tr.actor_forward(s, ab["mu"], ab["sigma"])
        sac_sample(ab["mu"], ab["sigma"], u_curr, ab["action"], ab["logp"])
        self.twinq.action_grad(s, ab["action"], ab["qmin"], ab["dqda"])
        sac_actor_grad(ab["mu"], ab["sigma"], u_curr, ab["dqda"], log_alpha, ab["dmu"], ab["dsigma"])
        self._actor_tr.actor_backward(ab["dmu"], ab["dsigma"])
        if dist_on:
            self._actor_tr.load_vector(PGRADS, ACTOR, all_reduce_mean(self._actor_tr.vector(PGRADS, ACTOR)))
        self._actor_tr.actor_apply(self.learning_rate)
        # entropy parameter (:155-164) and the loss values
        if dist_on:
            # The entropy parameter's loss and gradient are means over the GLOBAL batch, and both are linear in the batch means of
            # logp and qmin: all-reduce those two numbers and run the same device step on a one-row "batch" holding them.  One
            # owner of the parameter's Adam state (the device one) in single-process and data-parallel runs alike, so
            # optimizer_state() / save() / load() carry it either way.
            m = all_reduce_mean(torch.stack([ab["logp"].mean(), ab["qmin"].mean()]))
            sac_entropy_step(m[0:1].contiguous(), m[1:2].contiguous(), self.target_entropy, self.learning_rate, log_alpha, self._ent_state,
                             self._ent_step, ab["losses"])
        else:
            sac_entropy_step(ab["logp"], ab["qmin"], self.target_entropy, self.learning_rate, log_alpha, self._ent_state, self._ent_step,
                             ab["losses"])
        self.twinq.soft_update(self.tau)
        lq = self.twinq.losses()
        return dict(actor=ab["losses"][0], critic_1=lq[0], critic_2=lq[1], entropy=ab["losses"][1])

    def count_parameters(self):
        from .agent import count_parameters
        return count_parameters(self.actor, self.critic_1, self.critic_2, self.target_critic_1, self.target_critic_2, self.entropy)

    def step(self, env, train=True):
        """act() + do_step + replay add/update_last + check_train() for all envs (soft_actor_critic.cpp:47-91)."""
        self._prev.copy_(env.obs)
        action, _, _ = self.fused.forward(self._prev, seed=self.noise_seed, actor_only=True, out=self._act)
        st = env.step_autoreset(action)
        self.replay.push(self._prev, action, st.reward, st.done, st.valid, st.state)
        self._pushes_since_check += 1
        if train and self.global_step % self.train_every == self.train_every - 1 and self.has_enough():
            self.update()
        self.global_step += 1
        return st

    def has_enough(self):
        """ReplayBuffer::has_enough(batch_size) (soft_actor_critic.cpp:64): no update before the memory holds a batch — with
        fewer stored transitions the sampler would repeat rows, with none it would hand zeros to four Adam steps.  The stored
        count lives on the device (the ring compacts valid rows there); reading it is a stream synchronisation, so it is
        read only when the last reading no longer guarantees a batch: one push changes the count by at most n_envs."""
        if self._pushes_since_check * self.replay.N < self._stored_margin:
            return True
        stored = self.replay.stats()["transitions"]
        self._stored_margin = max(stored - self.batch_size, 0)
        self._pushes_since_check = 0
        return stored >= self.batch_size

    def _train_once(self):
        return self._train_once_hip()

    def update(self):
        for e in range(self.epoch):
            self.replay.sample(self.batch_size, seed=(self.noise_seed << 32) ^ (self.train_steps * 1000003 + e), out=self._batch)
            if not self.use_graph:
                self.last_losses = self._train_once()
            elif self._graph is None:
                # warm-up on a side stream (allocator, optimiser state), then capture one train() call.  The warm-up calls are
                # real Adam steps: weights, moments, step counters and the entropy parameter are put back afterwards, so that
                # this first update is ONE train() call like every later one (soft_actor_critic.cpp:93-170)
                snap = self._snapshot()
                side = torch.cuda.Stream(self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._train_once()
                torch.cuda.current_stream(self.device).wait_stream(side)
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph):
                    self.last_losses = self._train_once()
                self._restore(snap)
                self._graph.replay()  # capturing records the launches without running them
            else:
                self._graph.replay()
            self.train_steps += 1

    def _snapshot(self):
        from .ppo import ACTOR, PARAMS as PP
        from .qnet import PARAMS
        return dict(opt=self.optimizer_state(), train_steps=self.train_steps, actor=self._actor_tr.vector(PP, ACTOR).clone(),
                    q=[self.twinq.vector(PARAMS, i).clone() for i in range(4)], log_alpha=self.entropy.log_alpha.detach().clone())

    def _restore(self, snap):
        from .qnet import PARAMS
        self._actor_tr.set_flat(snap["actor"], self._critic_dummy, reset_optimizer=False)
        for i in range(4):
            self.twinq.load_vector(PARAMS, i, snap["q"][i])
        with torch.no_grad():
            self.entropy.log_alpha.copy_(snap["log_alpha"])
        self.load_optimizer_state(snap["opt"])
        self.train_steps = snap["train_steps"]

    def save(self, folder):
        """SoftActorCriticAgent::save (soft_actor_critic.cpp:181-201): the six module archives and the four
        torch::optim::Adam archives (`actor_optimizer.th`, `critic_1_optimizer.th`, `critic_2_optimizer.th`,
        `entropy_optimizer.th`) in the reference's formats (checkpoint.py): the folder loads in the reference's
        SoftActorCriticAgent::load and the other way round."""
        import os
        from .checkpoint import adam_states_from_flat, save_adam_th, save_th
        self.sync_modules()
        for m, f in ((self.actor, "actor.th"), (self.critic_1, "critic_1.th"), (self.target_critic_1, "target_critic_1.th"),
                     (self.critic_2, "critic_2.th"), (self.target_critic_2, "target_critic_2.th"), (self.entropy, "entropy.th")):
            save_th(m, os.path.join(folder, f))
        st = self.optimizer_state()
        for mod, key, f in ((self.actor, "actor", "actor_optimizer.th"), (self.critic_1, "critic_1", "critic_1_optimizer.th"),
                            (self.critic_2, "critic_2", "critic_2_optimizer.th")):
            save_adam_th(os.path.join(folder, f), adam_states_from_flat(mod, st[key]["step"], st[key]["m"], st[key]["v"]), self.learning_rate)
        e = st["entropy"]
        save_adam_th(os.path.join(folder, "entropy_optimizer.th"),
                     [None if e["step"] == 0 else (e["step"], e["state"][0:1].clone(), e["state"][1:2].clone())], self.learning_rate)

    def optimizer_state(self):
        """Adam moments and step counts of the four optimisers (all four live in the HIP trainers, in single-process and
        data-parallel runs alike)"""
        from .ppo import ACTOR, ACTOR_DEV_STEP, EXP_AVG, EXP_AVG_SQ
        tr, tq = self._actor_tr, self.twinq
        out = dict(kind="hip", train_steps=self.train_steps,
                   actor=dict(m=tr.vector(EXP_AVG, ACTOR).cpu(), v=tr.vector(EXP_AVG_SQ, ACTOR).cpu(), step=tr.adam_step(ACTOR_DEV_STEP)),
                   entropy=dict(state=self._ent_state.cpu(), step=int(self._ent_step.item())))
        for i in (0, 1):
            out["critic_%d" % (i + 1)] = dict(m=tq.vector(EXP_AVG, i).cpu(), v=tq.vector(EXP_AVG_SQ, i).cpu(), step=tq.adam_step(i))
        return out

    def load_optimizer_state(self, st):
        from .ppo import ACTOR, ACTOR_DEV_STEP, EXP_AVG, EXP_AVG_SQ
        tr, tq = self._actor_tr, self.twinq
        tr.load_vector(EXP_AVG, ACTOR, st["actor"]["m"].to(self.device))
        tr.load_vector(EXP_AVG_SQ, ACTOR, st["actor"]["v"].to(self.device))
        tr.adam_step(ACTOR_DEV_STEP, int(st["actor"]["step"]))
        for i in (0, 1):
            c = st["critic_%d" % (i + 1)]
            tq.load_vector(EXP_AVG, i, c["m"].to(self.device))
            tq.load_vector(EXP_AVG_SQ, i, c["v"].to(self.device))
            tq.adam_step(i, int(c["step"]))
        self._ent_state.copy_(st["entropy"]["state"].to(self.device))
        self._ent_step.fill_(int(st["entropy"]["step"]))
        self.train_steps = int(st.get("train_steps", self.train_steps))

    def load(self, folder):
        """SoftActorCriticAgent::load (soft_actor_critic.cpp:203-223): modules and the four Adam archives — written by this
        package or by the reference; a missing file is an error like the reference's load_torch ("Could not find")."""
        import os
        from .checkpoint import adam_flat_from_states, load_adam_th, load_into
        for m, f in ((self.actor, "actor.th"), (self.critic_1, "critic_1.th"), (self.target_critic_1, "target_critic_1.th"),
                     (self.critic_2, "critic_2.th"), (self.target_critic_2, "target_critic_2.th"), (self.entropy, "entropy.th")):
            load_into(m, os.path.join(folder, f))
        self.fused.set_actor(self.actor)
        self._push_critics()
        self._push_actor()
        st = dict(kind="hip")
        for mod, key, f in ((self.actor, "actor", "actor_optimizer.th"), (self.critic_1, "critic_1", "critic_1_optimizer.th"),
                            (self.critic_2, "critic_2", "critic_2_optimizer.th")):
            states, options = load_adam_th(os.path.join(folder, f))
            step, m, v = adam_flat_from_states(mod, states)
            st[key] = dict(step=step, m=m, v=v)
            if options["lr"] != self.learning_rate:
                self._graph = None   # a captured update has the old learning rate baked into its kernel arguments
            self.learning_rate = options["lr"]
        states, _ = load_adam_th(os.path.join(folder, "entropy_optimizer.th"))
        e = states[0] if states else None
        st["entropy"] = dict(step=0 if e is None else e[0],
                             state=torch.zeros(2) if e is None else torch.cat([e[1].reshape(-1)[:1], e[2].reshape(-1)[:1]]).float())
        st["train_steps"] = st["actor"]["step"]  # one actor step per train() call
        self.load_optimizer_state(st)


# ---- the reference's Agent surface for SAC: act / done / check_train over a flat replay buffer of transitions ------------------
class ReplayBuffer:
    """ReplayBuffer (evo_motion_networks/src/replay_buffer.cpp:16-52,146-153): a FIFO of at most `size` transitions
    dict(state, action, reward, done, next_state).  The newest one is still open — add() stores it with reward 0, done false and
    next_state = state; update_last() fills those in — and sample() never returns it: the indices of all the others are shuffled
    and the first batch_size taken.  The reference shuffles with its own std::mt19937(seed) + std::shuffle (replay_buffer.cpp:14,21):
    stdrandom.py restates both (held to g++ / libstdc++ and to the draws recorded from the compiled reference,
    tests/test_stdrandom.py), so the default draws ARE the reference's for the same seed; `shuffle(index_list)` stays a hook for tests."""

    def __init__(self, size, seed):
        from .stdrandom import Mt19937, std_shuffle
        self.size, self.memory = int(size), []
        self._rng = Mt19937(seed)
        self.shuffle = lambda index: std_shuffle(list(index), self._rng)

    def empty(self):
        return not self.memory

    def add(self, item):
        self.memory.append(item)
        while len(self.memory) > self.size:
            self.memory.pop(0)

    def update_last(self, reward, next_state, done):
        last = self.memory[-1]
        last["reward"], last["next_state"], last["done"] = reward, next_state, done

    def has_enough(self, batch_size):
        return len(self.memory) - 1 >= batch_size

    def sample(self, batch_size):
        index = self.shuffle(list(range(len(self.memory) - 1)))
        return [self.memory[i] for i in index[:batch_size]]


class SoftActorCriticAgent(VecSacAgent):
    """SoftActorCriticAgent with the reference's Agent surface (evo_motion_networks/include/evo_motion_networks/agent.h:16-35,
    src/agents/soft_actor_critic.cpp:47-91,172-180) for ONE environment, call for call: act(state, reward) -> action with the
    previous transition's reward, done(state, reward), a ReplayBuffer of transitions, `epoch` train() calls on `batch_size` sampled
    transitions whenever global_curr_step % train_every == train_every - 1 and the buffer holds a batch.  The actor forward is the
    fused HIP kernel, train() the HIP update of VecSacAgent (evm_q_* / evm_ppo_actor_* / evm_sac_*) — this class is the bookkeeping
    around them.  VecSacAgent.step()/update() with the device-resident ring stays the fast path for thousands of environments.

    Kept from the reference on purpose: act() rewrites the newest transition whenever the buffer is not empty, ALSO right after
    done() — the first act() of an episode turns the terminal transition back into (reset's reward, first state of the new episode,
    done = false).  The reference's SAC therefore never trains on a done flag (tests/golden/sac_loop_golden.txt shows it)."""

    def __init__(self, seed, state_space, action_space, actor_hidden_size=256, critic_hidden_size=256, batch_size=256, epoch=1,
                 learning_rate=1e-3, gamma=0.99, tau=0.005, replay_buffer_size=4096, train_every=32, device=0):
        super().__init__(seed, state_space, action_space, actor_hidden_size=actor_hidden_size, critic_hidden_size=critic_hidden_size,
                         batch_size=batch_size, epoch=epoch, learning_rate=learning_rate, gamma=gamma, tau=tau, replay_buffer_size=2,
                         train_every=train_every, device=device, n_envs=1, use_graph=False)
        self.replay_buffer = ReplayBuffer(replay_buffer_size, seed)
        self.S, self.A = int(state_space[0]), int(action_space[0])
        self.curr_episode_step = self.curr_train_step = self.global_curr_step = 0
        self._act_calls = 0
        from .metrics import LossMeter
        # soft_actor_critic.cpp:36-38: the six meters of get_metrics(), windows of 64
        self.actor_loss_meter, self.critic_1_loss_meter = LossMeter("actor", 64), LossMeter("critic_1", 64)
        self.critic_2_loss_meter, self.entropy_loss_meter = LossMeter("critic_2", 64), LossMeter("entropy", 64)
        self.episode_steps_meter, self.rewards_meter = LossMeter("steps", 64), LossMeter("rewards", 64)

    def _obs(self, state):
        return state.to(device=self.device, dtype=torch.float32).reshape(1, self.S).contiguous()

    def act(self, state, reward, uniform=None, train_uniforms=None):
        """state [S]; reward: the PREVIOUS transition's (soft_actor_critic.cpp:53).  uniform [A]: the U[0,1) draws of
        truncated_normal_sample (the reference's at::rand), else the kernel's counter-based generator; train_uniforms: per epoch of
        the train() calls this act() may trigger, (u_next, u_curr) [batch_size, A] (tests)."""
        obs = self._obs(state)
        if uniform is not None:
            uniform = uniform.to(device=self.device, dtype=torch.float32).reshape(1, self.A).contiguous()
        self._act_calls += 1
        action, _, _ = self.fused.forward(obs, uniform=uniform, seed=(self.noise_seed + 7919 * self._act_calls) & 0x7FFFFFFF, actor_only=True)
        action = action[0].clone()
        if not self.replay_buffer.empty():
            self.replay_buffer.update_last(float(reward), obs[0], False)
        self.replay_buffer.add(dict(state=obs[0], action=action, reward=0.0, done=False, next_state=obs[0]))
        self.check_train(train_uniforms)
        self.curr_episode_step += 1
        self.global_curr_step += 1
        return action

    def check_train(self, train_uniforms=None):
        if not (self.global_curr_step % self.train_every == self.train_every - 1 and self.replay_buffer.has_enough(self.batch_size)):
            return 0
        s, a, r, d, n = self._batch
        for e in range(self.epoch):
            items = self.replay_buffer.sample(self.batch_size)
            s.copy_(torch.stack([i["state"] for i in items]))
            a.copy_(torch.stack([i["action"] for i in items]))
            n.copy_(torch.stack([i["next_state"] for i in items]))
            r.copy_(torch.tensor([i["reward"] for i in items], dtype=torch.float32))
            d.copy_(torch.tensor([1.0 if i["done"] else 0.0 for i in items], dtype=torch.float32))
            u = train_uniforms[e] if train_uniforms is not None else (None, None)
            dev = lambda t: None if t is None else t.to(device=self.device, dtype=torch.float32).contiguous()
            self.last_losses = self._train_once_hip(u_next=dev(u[0]), u_curr=dev(u[1]))
            # soft_actor_critic.cpp:164-167 (float() reads the device values: this is the reference-shaped one-env surface)
            for meter, key in ((self.actor_loss_meter, "actor"), (self.critic_1_loss_meter, "critic_1"),
                               (self.critic_2_loss_meter, "critic_2"), (self.entropy_loss_meter, "entropy")):
                meter.add(float(self.last_losses[key]))
            self.train_steps += 1
            self.curr_train_step += 1
        return self.epoch

    def done(self, state, reward):
        """the episode has ended in `state` (its terminal observation) with `reward` (soft_actor_critic.cpp:172-180)"""
        self.replay_buffer.update_last(float(reward), self._obs(state)[0], True)
        self.rewards_meter.add(float(reward))                          # :175-176
        self.episode_steps_meter.add(float(self.curr_episode_step))
        self.curr_episode_step = 0

    def get_metrics(self):
        """soft_actor_critic.cpp:223-226"""
        return [self.actor_loss_meter, self.critic_1_loss_meter, self.critic_2_loss_meter, self.entropy_loss_meter,
                self.episode_steps_meter, self.rewards_meter]

    def to(self, device):
        return self

    def set_eval(self, eval_mode):
        pass
