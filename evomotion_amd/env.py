"""Vectorised robot_walk environment on one MI355X — Python mirror of the reference's Environment API
(evo_motion_model/include/evo_motion_model/environment.h:35-97) on top of the C ABI.

`VecRobotWalk` keeps the reference's method names (reset / do_step / get_state_space / get_action_space);
`step` is a (state, reward, done) tuple of device tensors like the reference's `struct step`
(environment.h:20-24), with a leading env dimension.  torch is used only to own device memory and streams.
"""
import ctypes
from collections import namedtuple

import numpy as np
import torch

from . import _lib
from ._lib import EvmEnvParams, check, lib

Step = namedtuple("Step", ["state", "reward", "done"])
RolloutStep = namedtuple("RolloutStep", ["state", "reward", "done", "valid"])

# Collision mode when the parameters do not say: 1 = member-vs-member contacts as in the reference (every pair of members except
# constraint parent / child, constraint.cpp:65,147), 0 = floor contacts only (the north-star's plane-contact configuration)
SELF_COLLISION_DEFAULT = 1

# RobotWalkFactory parameter names and defaults (evo_motion_model/src/env/env_factory.cpp:74-83); "self_collision" is this
# path's own switch
_PARAM_DEFAULTS = {
    "self_collision": SELF_COLLISION_DEFAULT,
    "skeleton_json_path": _lib.DEFAULT_SKELETON,
    "initial_remaining_seconds": 1.0,
    "max_episode_seconds": 30.0,
    "target_velocity": 0.5,
    "minimal_velocity": 0.1,
    "reset_frames": 30,
}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# RobotJumpFactory parameter names and defaults (env_factory.cpp:91-100)
_JUMP_DEFAULTS = {
    "self_collision": SELF_COLLISION_DEFAULT,
    "skeleton_json_path": _lib.DEFAULT_SKELETON,
    "minimal_velocity": 0.1,
    "target_velocity": 0.5,
    "max_seconds": 30.0,
    "initial_seconds": 1.0,
    "reset_seconds": 1.0 / 6.0,
}


def get_environment(env_name, n_envs, seed=1234, device=0, parameters=None):
    """get_environment_factory(env_name, parameters)->get_env(num_threads, seed) (environment.h:96-97, env_factory.cpp:109-120)
    for n_envs environments; unknown names raise ValueError like the reference's std::invalid_argument."""
    if env_name == "robot_walk":
        return VecRobotWalk(n_envs, seed, device, parameters)
    if env_name == "robot_jump":
        return VecRobotJump(n_envs, seed, device, parameters)
    raise ValueError(env_name)


class VecRobotWalk:
    ENV_KIND = 0

    def _parameters(self, parameters):
        # keys the factory does not know are ignored, as EnvironmentFactory::generic_get_value does (env_factory.cpp:22-28), so
        # one parameter map can serve several environments; parameters={"strict": 1, ...} turns them into ValueError
        prm = dict(_PARAM_DEFAULTS)
        strict = bool(int((parameters or {}).get("strict", 0)))
        for k, v in (parameters or {}).items():
            if k == "strict":
                continue
            if k not in prm:
                if strict:
                    raise ValueError(k)
                continue
            prm[k] = type(_PARAM_DEFAULTS[k])(v)
        return prm

    def __init__(self, n_envs, seed=1234, device=0, parameters=None):
        prm = self._parameters(parameters)
        if not torch.cuda.is_available():
            raise _lib.EvmError("VecRobotWalk needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", device)
        self.n_envs = int(n_envs)
        p = EvmEnvParams(prm["initial_remaining_seconds"], prm["max_episode_seconds"], prm["target_velocity"],
                         prm["minimal_velocity"], prm["reset_frames"], self.ENV_KIND, int(prm.get("self_collision", SELF_COLLISION_DEFAULT)))
        self._h = ctypes.c_void_p()
        # int(initial_remaining_seconds / dt) in fp32 (robot_walk.cpp:30, SURVEY App. D: 59 for the default 1 s)
        self._remaining0 = int(np.float32(prm["initial_remaining_seconds"]) / (np.float32(1.0) / np.float32(60.0)))
        torch.cuda.set_device(self.device)
        check(lib.evm_env_create(prm["skeleton_json_path"].encode(), self.n_envs, device, seed, ctypes.byref(p),
                                 ctypes.byref(self._h)))
        s, a = ctypes.c_int(), ctypes.c_int()
        check(lib.evm_env_spaces(self._h, ctypes.byref(s), ctypes.byref(a)))
        self.state_dim, self.action_dim = s.value, a.value
        c = [ctypes.c_int() for _ in range(4)]
        check(lib.evm_env_counts(self._h, *[ctypes.byref(x) for x in c]))
        _, self.n_bodies, self.n_members, self.n_muscles = [x.value for x in c]
        npairs = ctypes.c_int()
        check(lib.evm_env_pairs(self._h, ctypes.byref(npairs), None))
        self.n_pairs = npairs.value
        self.obs = torch.zeros(self.n_envs, self.state_dim, device=self.device)
        self.reward = torch.zeros(self.n_envs, device=self.device)
        self.done = torch.zeros(self.n_envs, dtype=torch.uint8, device=self.device)
        self.valid = torch.zeros(self.n_envs, dtype=torch.uint8, device=self.device)

    def close(self):
        if self._h:
            lib.evm_env_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference API ---------------------------------------------------------------------------
    def get_state_space(self):
        return [self.state_dim]

    def get_action_space(self):
        return [self.action_dim]

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, mask=None):
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        check(lib.evm_env_reset(self._h, _ptr(mask), _ptr(self.obs), _ptr(self.reward), _ptr(self.done), self._stream()))
        return Step(self.obs, self.reward, self.done)

    def _action(self, action):
        if action.device != self.device or action.dtype != torch.float32 or not action.is_contiguous():
            action = action.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(action.shape) != (self.n_envs, self.action_dim):
            raise ValueError(f"action must be [{self.n_envs}, {self.action_dim}]")
        return action

    def do_step(self, action):
        action = self._action(action)
        check(lib.evm_env_step(self._h, _ptr(action), _ptr(self.obs), _ptr(self.reward), _ptr(self.done), self._stream()))
        return Step(self.obs, self.reward, self.done)

    def step_autoreset(self, action, reward_out=None, done_out=None, valid_out=None, obs_out=None):
        """`*_out`: contiguous [n_envs] device tensors (f32 / u8 / u8), e.g. rows of a rollout buffer, written directly by
        the kernel instead of the env's own buffers (no copies in the rollout loop).  obs_out [n_envs, state_dim] f32
        becomes `self.obs` (the next observation lands in the caller's rollout buffer)."""
        action = self._action(action)
        if obs_out is not None:
            assert obs_out.dtype == torch.float32 and obs_out.is_contiguous() and tuple(obs_out.shape) == (self.n_envs, self.state_dim)
            self.obs = obs_out
        r = self.reward if reward_out is None else reward_out
        d = self.done if done_out is None else done_out
        v = self.valid if valid_out is None else valid_out
        assert r.dtype == torch.float32 and d.dtype == torch.uint8 and v.dtype == torch.uint8
        assert r.is_contiguous() and d.is_contiguous() and v.is_contiguous() and r.numel() == d.numel() == v.numel() == self.n_envs
        check(lib.evm_env_step_autoreset(self._h, _ptr(action), _ptr(self.obs), _ptr(r), _ptr(d), _ptr(v), self._stream()))
        return RolloutStep(self.obs, r, d, v)

    # -- parity / checkpoint hooks ---------------------------------------------------------------
    def body_poses(self):
        out = torch.empty(self.n_envs, self.n_bodies, 7, device=self.device)
        check(lib.evm_env_get_body_poses(self._h, _ptr(out), self._stream()))
        return out

    def state_size(self):
        return lib.evm_env_state_size(self._h)

    def get_state(self):
        out = np.zeros((self.n_envs, self.state_size()), np.float32)
        check(lib.evm_env_get_state(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return out

    def set_state(self, blob):
        blob = np.ascontiguousarray(blob, np.float32)
        assert blob.shape == (self.n_envs, self.state_size())
        check(lib.evm_env_set_state(self._h, blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))

    def stagger_episodes(self, period=None):
        """Spread the envs' episode phases.  Envs created together leave reset() in lock step and, under random actions,
        fail within a step of each other (~59 do_step calls, then 60 settle calls): for a long time the whole batch is
        either stepping or settling.  This sets every env's `remaining_steps` budget (robot_walk.cpp:64-68) to a value
        spread evenly over one episode + reset cycle, so that first episodes end at evenly spread calls; every later
        episode is an ordinary one.  A vectorisation helper (the single-env reference has nothing to desynchronise)."""
        if period is None:
            period = 2 * (self._remaining0 + 1)
        blob = self.get_state()
        blob[:, -1] = 1 + (np.arange(self.n_envs) * 7919 % period)  # remaining_steps is the blob's last float
        self.set_state(blob)

    def debug_reset_begin(self, mask=None):
        check(lib.evm_env_debug_reset_begin(self._h, _ptr(mask)))

    def debug_physics_steps(self, n, mask=None):
        check(lib.evm_env_debug_physics_steps(self._h, n, _ptr(mask)))

    def body_constants(self):
        out = np.zeros((self.n_bodies, 19), np.float32)
        check(lib.evm_env_get_body_constants(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return out

    def diagnostics(self):
        out = torch.empty(self.n_envs, 2, device=self.device)
        check(lib.evm_env_get_diagnostics(self._h, _ptr(out), self._stream()))
        return out

    def residual(self, clear=True):
        """largest |delta impulse| a constraint row applied in the last of the 10 Gauss-Seidel sweeps, maximum over the whole
        batch and over the physics steps since the last clear (device-side reduction)"""
        out = ctypes.c_float()
        check(lib.evm_env_get_residual(self._h, ctypes.byref(out), 1 if clear else 0, self._stream()))
        return out.value

    def errors(self, clear=True):
        """(schedule waits that timed out, contact manifolds left out of a step) since the last clear: both must stay 0"""
        out = (ctypes.c_int * 2)()
        check(lib.evm_env_get_errors(self._h, out, 1 if clear else 0, self._stream()))
        return int(out[0]), int(out[1])

    def penetration_queries(self, clear=True):
        """narrowphase queries since the last clear that went through the penetration-depth solver (overlapping cores)"""
        out = (ctypes.c_int * 3)()
        check(lib.evm_env_get_pair_counters(self._h, out, 1 if clear else 0, self._stream()))
        self.predicted_penetration_queries = int(out[1])   # ... of which the previous step had predicted (urgent list)
        self.urgent_entries = int(out[2])                  # entries of the urgent list (predictions, right or wrong)
        return int(out[0])

    def speculation_counters(self, clear=True):
        """the urgent list's speculation blocks (penetration queries run beside the pair's own query): (runs, answers used, waits that ran out)"""
        out = (ctypes.c_int * 3)()
        check(lib.evm_env_get_speculation_counters(self._h, out, 1 if clear else 0, self._stream()))
        return int(out[0]), int(out[1]), int(out[2])

    def stats(self):
        out = (ctypes.c_longlong * 2)()
        check(lib.evm_env_get_stats(self._h, out))
        return dict(env_steps=int(out[0]), resets=int(out[1]))

    def clear_stats(self):
        check(lib.evm_env_clear_stats(self._h))

    def timing_begin(self):
        check(lib.evm_env_timing_begin(self._h, self._stream()))

    def timing_end(self):
        ms, n = ctypes.c_float(), ctypes.c_int()
        check(lib.evm_env_timing_end(self._h, self._stream(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def timing_end_detail(self):
        """(ms of whole steps, steps, ms inside the sweeps kernel — 0 when the monolithic kernel ran)"""
        ms, n, sw = ctypes.c_float(), ctypes.c_int(), ctypes.c_float()
        check(lib.evm_env_timing_end_detail(self._h, self._stream(), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(sw)))
        return ms.value, n.value, sw.value


class VecRobotJump(VecRobotWalk):
    """robot_jump (evo_motion_model/src/env/robot_jump.cpp): the same skeleton, world and step; reward max(vy, 0) + vz,
    fail on remaining < 0, reset angles within pi/3 and int(reset_seconds / dt) settle steps."""
    ENV_KIND = 1

    def _parameters(self, parameters):
        prm = dict(_JUMP_DEFAULTS)
        strict = bool(int((parameters or {}).get("strict", 0)))
        for k, v in (parameters or {}).items():
            if k == "strict":
                continue
            if k not in prm:
                if strict:
                    raise ValueError(k)
                continue
            prm[k] = type(_JUMP_DEFAULTS[k])(v)
        dt = np.float32(1.0) / np.float32(60.0)
        return dict(skeleton_json_path=prm["skeleton_json_path"], self_collision=prm["self_collision"], initial_remaining_seconds=prm["initial_seconds"],
                    max_episode_seconds=prm["max_seconds"], target_velocity=prm["target_velocity"],
                    minimal_velocity=prm["minimal_velocity"], reset_frames=int(np.float32(prm["reset_seconds"]) / dt))
