"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy fp32 restatement of the agent half of the path:

  ActorModule / CriticModule forward    evo_motion_networks/src/networks/actor.cpp:9-48, critic.cpp:8-35
  truncated normal sample/log-pdf/entropy evo_motion_networks/src/functions.cpp:53-68,94-128
  GAE / advantage normalisation / returns evo_motion_networks/src/agents/ppo_gae.cpp:127-151

Pinned against tests/golden/agent_golden.txt, which is produced by the reference's own compiled code
(oracle/ref_build.sh + oracle/ref_golden.cpp) — see tests/test_agent_oracle.py.
"""
import math

import numpy as np
from scipy import special

F = np.float32
SIGMA_MIN, SIGMA_MAX, AB_BOUND = F(1e-6), F(1e6), F(5.0)  # functions.cpp:9-11


# ---- deterministic parameter pattern shared with oracle/ref_golden.cpp ------------------------------
def pat(tensor, k, scale):
    k = np.asarray(k, dtype=np.uint64)
    h = (np.uint64(tensor) * np.uint64(2654435761) + k * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0x5BD1E995)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    v = (h & np.uint64(0xFFFFFF)).astype(np.float32) / F(16777216.0) - F(0.5)
    return (v * F(scale)).astype(np.float32)


ACTOR_SHAPES = [("head.0.weight", (256, 371)), ("head.0.bias", (256,)), ("head.2.weight", (256,)), ("head.2.bias", (256,)),
                ("head.3.weight", (256, 256)), ("head.3.bias", (256,)), ("head.5.weight", (256,)), ("head.5.bias", (256,)),
                ("mu.0.weight", (12, 256)), ("mu.0.bias", (12,)), ("sigma.0.weight", (12, 256)), ("sigma.0.bias", (12,))]
CRITIC_SHAPES = [("critic.0.weight", (256, 371)), ("critic.0.bias", (256,)), ("critic.2.weight", (256,)), ("critic.2.bias", (256,)),
                 ("critic.3.weight", (256, 256)), ("critic.3.bias", (256,)), ("critic.5.weight", (256,)), ("critic.5.bias", (256,)),
                 ("critic.6.weight", (1, 256)), ("critic.6.bias", (1,))]


Q_SHAPES = [("q_network.0.weight", (256, 383)), ("q_network.0.bias", (256,)), ("q_network.2.weight", (256,)), ("q_network.2.bias", (256,)),
            ("q_network.3.weight", (256, 256)), ("q_network.3.bias", (256,)), ("q_network.5.weight", (256,)), ("q_network.5.bias", (256,)),
            ("q_network.6.weight", (256, 256)), ("q_network.6.bias", (256,)), ("q_network.8.weight", (256,)), ("q_network.8.bias", (256,)),
            ("q_network.9.weight", (1, 256)), ("q_network.9.bias", (1,))]  # q_net.cpp:8-27


def pattern_params(shapes, base):
    out = {}
    for t, (name, shape) in enumerate(shapes):
        n = int(np.prod(shape))
        if len(shape) == 2:
            scale, off = F(2.0) / np.sqrt(F(shape[1])), F(0)
        elif ".2." in name or ".5." in name or ".8." in name:
            scale, off = F(0.2), F(1.0) if "weight" in name else F(0)
        else:
            scale, off = F(0.2), F(0)
        out[name] = (off + pat(base + t, np.arange(n), scale)).astype(np.float32).reshape(shape)
    return out


# ---- modules ------------------------------------------------------------------------------------------
def mish(x):
    x = x.astype(np.float32)
    with np.errstate(over="ignore"):
        return (x * np.tanh(np.log1p(np.exp(x)))).astype(np.float32)


def softplus(x):  # torch::nn::Softplus(beta=1, threshold=20)
    x = x.astype(np.float32)
    with np.errstate(over="ignore"):
        return np.where(x > F(20), x, np.log1p(np.exp(x))).astype(np.float32)


def layer_norm(x, w, b, eps=F(1e-5)):
    m = x.mean(-1, keepdims=True, dtype=np.float32)
    v = ((x - m) ** 2).mean(-1, keepdims=True, dtype=np.float32)
    return ((x - m) / np.sqrt(v + eps) * w + b).astype(np.float32)


def trunk(x, p, pre):
    h = x.astype(np.float32) @ p[pre + ".0.weight"].T + p[pre + ".0.bias"]
    h = layer_norm(mish(h), p[pre + ".2.weight"], p[pre + ".2.bias"])
    h = h @ p[pre + ".3.weight"].T + p[pre + ".3.bias"]
    return layer_norm(mish(h), p[pre + ".5.weight"], p[pre + ".5.bias"])


def actor_forward(x, p):
    h = trunk(x, p, "head")
    mu = np.tanh(h @ p["mu.0.weight"].T + p["mu.0.bias"]).astype(np.float32)
    sigma = softplus(h @ p["sigma.0.weight"].T + p["sigma.0.bias"])
    return mu, sigma


def critic_forward(x, p):
    h = trunk(x, p, "critic")
    return (h @ p["critic.6.weight"].T + p["critic.6.bias"]).astype(np.float32)


def q_forward(x, a, p):
    """QNetworkModule::forward (q_net.cpp:29-43): cat(state, action) -> 3 x (Linear, Mish, LayerNorm) -> Linear(256, 1)."""
    h = np.concatenate([x, a], axis=-1).astype(np.float32)
    for lin, ln in (("0", "2"), ("3", "5"), ("6", "8")):
        h = (h @ p["q_network.%s.weight" % lin].T + p["q_network.%s.bias" % lin]).astype(np.float32)
        h = layer_norm(mish(h), p["q_network.%s.weight" % ln], p["q_network.%s.bias" % ln])
    return (h @ p["q_network.9.weight"].T + p["q_network.9.bias"]).astype(np.float32)


# ---- truncated normal ---------------------------------------------------------------------------------
def _theta(x):
    return (F(0.5) * (F(1.0) + special.erf((x / F(math.sqrt(2.0))).astype(np.float32)))).astype(np.float32)


def _phi(z):
    return (np.exp(F(-0.5) * z * z) / F(math.sqrt(2.0 * math.pi))).astype(np.float32)


def _ab(mu, sigma, lo, hi):
    s = np.clip(sigma.astype(np.float32), SIGMA_MIN, SIGMA_MAX)
    a = np.clip((F(lo) - mu) / s, -AB_BOUND, AB_BOUND).astype(np.float32)
    b = np.clip((F(hi) - mu) / s, -AB_BOUND, AB_BOUND).astype(np.float32)
    return s, a, b


def tn_log_pdf(x, mu, sigma, lo=-1.0, hi=1.0):
    s, a, b = _ab(mu, sigma, lo, hi)
    z = _theta(b) - _theta(a)
    with np.errstate(divide="ignore"):
        return (F(-0.5 * math.log(2.0 * math.pi)) - np.log(s) - F(0.5) * ((x - mu) / s) ** 2 - np.log(z)).astype(np.float32)


def tn_entropy(mu, sigma, lo=-1.0, hi=1.0):
    s, a, b = _ab(mu, sigma, lo, hi)
    z = _theta(b) - _theta(a)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (np.log(F(math.sqrt(2.0 * math.pi * math.e)) * s * z) + F(0.5) * (a * _phi(a) - b * _phi(b)) / z).astype(np.float32)


def tn_sample(mu, sigma, u, lo=-1.0, hi=1.0):
    s, a, b = _ab(mu, sigma, lo, hi)
    cdf = np.clip(_theta(a) + u.astype(np.float32) * (_theta(b) - _theta(a)), F(0), F(1)).astype(np.float32)
    inv = (F(math.sqrt(2.0)) * special.erfinv((F(2.0) * cdf - F(1.0)).astype(np.float32))).astype(np.float32)
    return np.clip(inv * s + mu, F(lo), F(hi)).astype(np.float32)


# ---- GAE (ppo_gae.cpp:127-151) ------------------------------------------------------------------------
def gae(rewards, done, curr_values, next_values, gamma, lam):
    """[B,T,1] padded tensors (done padded with 1) -> mask, normalised advantages, returns."""
    r, d, cv, nv = [np.asarray(t, np.float32) for t in (rewards, done, curr_values, next_values)]
    B, T = r.shape[:2]
    mask = np.concatenate([np.ones((B, 1, 1), np.float32), (F(1) - d)[:, : T - 1]], 1) == 1.0
    deltas = r + (F(1) - d) * F(gamma) * nv - cv
    adv = np.zeros_like(r)
    g = np.zeros((B, 1), np.float32)
    for t in range(T - 1, -1, -1):
        g = deltas[:, t] * mask[:, t] + F(gamma) * F(lam) * (F(1) - d[:, t]) * g
        adv[:, t] = g
    sel = adv[mask]
    adv = ((adv - sel.mean(dtype=np.float32)) / (sel.std(ddof=1, dtype=np.float32) + F(1e-8))).astype(np.float32)
    return mask, adv, (adv + cv).astype(np.float32)


# ---- PpoGaeAgent::act / done / check_train and TrajectoryReplayBuffer: the bookkeeping, restated ------------------------------
class TrajectoryBufferOracle:
    """AbstractTrajectoryBuffer / TrajectoryReplayBuffer (evo_motion_networks/src/replay_buffer.cpp:73-138,176-189):
    `memory` is a list of trajectories (lists of step dicts); the newest is the one being filled."""

    def __init__(self, size):
        self.size, self.memory = size, []

    def new_trajectory(self):                 # :105-110
        self.memory.append([])
        while len(self.memory) > self.size:
            del self.memory[0]

    def add(self, step):                      # :112-115
        self.memory[-1].append(step)

    def update_last(self, reward, done, next_value):   # :117-123 + update_last_step :176-186
        last = dict(self.memory[-1][-1])
        last.update(reward=reward, done=done, next_value=next_value)
        self.memory[-1][-1] = last

    def empty(self):
        return len(self.memory) == 0

    def trajectory_empty(self):
        return self.empty() or len(self.memory[-1]) == 0

    def filtered_positions(self):             # copy_if(trajectory.size() > 1)
        return [i for i, t in enumerate(self.memory) if len(t) > 1]

    def enough_trajectory(self, batch_size):  # :139-146
        return len(self.filtered_positions()) >= batch_size

    def sample(self, batch_size, shuffled_index):
        """:80-98 — shuffled_index: what std::shuffle(rand_gen) made of iota(filtered.size() - 1) (the reference's generator
        is libstdc++'s; the test supplies the recorded result)"""
        filtered = self.filtered_positions()
        assert sorted(shuffled_index) == list(range(len(filtered) - 1))
        return [self.memory[filtered[i]] for i in shuffled_index[:batch_size]]


class PpoLoopOracle:
    """PpoGaeAgent::act / done / check_train (evo_motion_networks/src/agents/ppo_gae.cpp:29-115) around two callbacks:
    forward(state, u) -> (action, log_prob, value) and train(batch dict of padded [B, T, ...] arrays)."""

    def __init__(self, forward, train, batch_size, train_every, replay_buffer_size):
        self.forward, self.train = forward, train
        self.batch_size, self.train_every = batch_size, train_every
        self.buffer = TrajectoryBufferOracle(replay_buffer_size)
        self.global_curr_step = self.curr_train_step = self.curr_episode_step = 0

    def act(self, state, reward, u):
        action, log_prob, value = self.forward(state, u)
        if self.buffer.empty():
            self.buffer.new_trajectory()
        if not self.buffer.trajectory_empty():
            self.buffer.update_last(reward, False, value)
        self.buffer.add(dict(state=state, action=action, reward=0.0, done=False, log_prob=log_prob, curr_value=value, next_value=value))
        self.curr_episode_step += 1
        return action

    def done(self, state, reward, shuffled_index=None):
        _, _, value = self.forward(state, None)
        self.buffer.update_last(reward, True, value)
        trained = self.check_train(shuffled_index)
        self.buffer.new_trajectory()
        self.global_curr_step += 1
        self.curr_episode_step = 0
        return trained

    def will_train(self):
        return self.global_curr_step % self.train_every == self.train_every - 1 and self.buffer.enough_trajectory(self.batch_size)

    def check_train(self, shuffled_index):
        if not self.will_train():
            return False
        episodes = self.buffer.sample(self.batch_size, shuffled_index)
        T = max(len(t) for t in episodes)

        def stack(key, width, pad_value=0.0):
            out = np.full((len(episodes), T, width), pad_value, np.float32)
            for b, traj in enumerate(episodes):
                for t, st in enumerate(traj):
                    out[b, t] = np.asarray(st[key], np.float32).reshape(-1) if key != "done" else (1.0 if st["done"] else 0.0)
            return out
        S, A = len(episodes[0][0]["state"]), len(episodes[0][0]["action"])
        batch = dict(states=stack("state", S), actions=stack("action", A), rewards=stack("reward", 1), done=stack("done", 1, 1.0),
                     log_prob=stack("log_prob", A), curr_values=stack("curr_value", 1), next_values=stack("next_value", 1))
        self.train(batch)
        self.curr_train_step += 1
        return True


# ---- SoftActorCriticAgent::act / done / check_train and the flat ReplayBuffer: the bookkeeping, restated ------------------------
class ReplayBufferOracle:
    """AbstractReplayBuffer / ReplayBuffer (evo_motion_networks/src/replay_buffer.cpp:16-52,146-153): a FIFO of transitions
    dict(state, action, reward, done, next_state); the newest one is still open (its reward / done / next_state arrive with the
    next act() or with done()) and is never sampled."""

    def __init__(self, size):
        self.size, self.memory = size, []

    def empty(self):
        return not self.memory

    def add(self, item):                       # :29-33
        self.memory.append(item)
        while len(self.memory) > self.size:
            del self.memory[0]

    def update_last(self, reward, next_state, done):   # :36-40 + update_last_item :146-153
        last = dict(self.memory[-1])
        last.update(reward=reward, next_state=next_state, done=done)
        self.memory[-1] = last

    def has_enough(self, batch_size):          # :48-51
        return len(self.memory) - 1 >= batch_size

    def sample(self, batch_size, shuffled_index):
        """:16-27 — shuffled_index: what std::shuffle(rand_gen) made of iota(memory.size() - 1)"""
        assert sorted(shuffled_index) == list(range(len(self.memory) - 1))
        return [self.memory[i] for i in shuffled_index[:batch_size]]


class SacLoopOracle:
    """SoftActorCriticAgent::act / done / check_train (evo_motion_networks/src/agents/soft_actor_critic.cpp:47-91,172-180) around two
    callbacks: forward(state, u) -> action and train(batch dict of [B, ...] arrays, u_next, u_curr).

    As in the reference, act() ALWAYS rewrites the newest transition when the buffer is not empty — also right after done(): the
    first act() of an episode overwrites the terminal transition's (reward, next_state, done = true) with (the reset's reward, the
    new episode's first state, done = false).  A done flag therefore only ever sits on the newest element, which sample() never
    returns: the reference's SAC never trains on done = 1 (pinned by the buffer dumps of tests/golden/sac_loop_golden.txt)."""

    def __init__(self, forward, train, batch_size, epoch, train_every, replay_buffer_size):
        self.forward, self.train = forward, train
        self.batch_size, self.epoch, self.train_every = batch_size, epoch, train_every
        self.buffer = ReplayBufferOracle(replay_buffer_size)
        self.global_curr_step = self.curr_train_step = self.curr_episode_step = 0

    def act(self, state, reward, u, shuffles=(), train_u=()):
        """shuffles / train_u: per epoch of the train call this act() may trigger, the recorded shuffle and (u_next, u_curr)"""
        action = self.forward(state, u)
        if not self.buffer.empty():
            self.buffer.update_last(reward, state, False)
        self.buffer.add(dict(state=state, action=action, reward=0.0, done=False, next_state=state))
        trained = self.check_train(shuffles, train_u)
        self.curr_episode_step += 1
        self.global_curr_step += 1
        return action, trained

    def will_train(self):
        return self.global_curr_step % self.train_every == self.train_every - 1 and self.buffer.has_enough(self.batch_size)

    def check_train(self, shuffles, train_u):
        if not self.will_train():
            return 0
        for e in range(self.epoch):
            items = self.buffer.sample(self.batch_size, shuffles[e])
            batch = dict(states=np.stack([i["state"] for i in items]).astype(np.float32), actions=np.stack([i["action"] for i in items]).astype(np.float32),
                         rewards=np.array([[i["reward"]] for i in items], np.float32), done=np.array([[1.0 if i["done"] else 0.0] for i in items], np.float32),
                         next_states=np.stack([i["next_state"] for i in items]).astype(np.float32))
            self.train(batch, train_u[e][0], train_u[e][1])
            self.curr_train_step += 1
        return self.epoch

    def done(self, state, reward):
        self.buffer.update_last(reward, state, True)
        self.curr_episode_step = 0
