"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
ORC_LIB = os.path.join(ORC_DIR, "liborc.so")
SKEL = os.path.join(ROOT, "evomotion_amd", "data", "robot_walk_spider.skel")

fp = ctypes.POINTER(ctypes.c_float)
ip = ctypes.POINTER(ctypes.c_int)


def build():
    subprocess.check_call(["make", "-s", "-C", ORC_DIR])


def load():
    if not os.path.exists(ORC_LIB):
        build()
    L = ctypes.CDLL(ORC_LIB)
    vp = ctypes.c_void_p
    L.orc_last_error.restype = ctypes.c_char_p
    L.orc_env_create.restype = vp
    L.orc_env_create.argtypes = [ctypes.c_char_p, ctypes.c_int] + [ctypes.c_float] * 4 + [ctypes.c_int]
    L.orc_env_create_kind.restype = vp
    L.orc_env_create_kind.argtypes = [ctypes.c_char_p, ctypes.c_int] + [ctypes.c_float] * 4 + [ctypes.c_int, ctypes.c_int]
    L.orc_env_create_ex.restype = vp
    L.orc_env_create_ex.argtypes = [ctypes.c_char_p, ctypes.c_int] + [ctypes.c_float] * 4 + [ctypes.c_int] * 3
    L.orc_env_num_pairs.argtypes = [vp]
    L.orc_env_get_pairs.argtypes = [vp, ip]
    L.orc_env_get_pair_stats.argtypes = [vp, ip, fp]
    L.orc_env_get_pair_totals.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
    L.orc_gjk_query.argtypes = [fp, ctypes.c_int, fp, fp, fp, ctypes.c_int, fp, fp, ctypes.c_float, fp]
    L.orc_epa_query.argtypes = [fp, ctypes.c_int, fp, fp, fp, ctypes.c_int, fp, fp, fp]
    L.orc_set_penetration_solver.argtypes = [ctypes.c_int]
    L.orc_set_floor_as_hull.argtypes = [ctypes.c_int]
    L.orc_set_ccd_pretest.argtypes = [ctypes.c_int]
    L.orc_set_floor_hull_half.argtypes = [ctypes.c_float]
    L.orc_env_get_floor_stats.argtypes = [vp, ip]
    L.orc_env_destroy.argtypes = [vp]
    for f in ["orc_env_obs_dim", "orc_env_act_dim", "orc_env_num_bodies", "orc_env_num_members", "orc_env_state_size"]:
        getattr(L, f).argtypes = [vp]
    L.orc_env_reset.argtypes = [vp, fp, fp, ip]
    L.orc_env_step.argtypes = [vp, fp, fp, fp, ip]
    L.orc_env_reset_begin.argtypes = [vp]
    L.orc_env_apply_action.argtypes = [vp, fp]
    L.orc_env_physics_step.argtypes = [vp]
    L.orc_env_compute_step.argtypes = [vp, fp, fp, ip]
    L.orc_env_get_counters.argtypes = [vp, ip]
    L.orc_env_set_counters.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.orc_env_get_state.argtypes = [vp, fp]
    L.orc_env_set_state.argtypes = [vp, fp]
    L.orc_env_get_poses.argtypes = [vp, fp]
    L.orc_env_get_body_constants.argtypes = [vp, fp]
    L.orc_selftest_rng.argtypes = [ctypes.c_int, ctypes.c_int]
    L.orc_rng_draws.argtypes = [ctypes.c_int, ctypes.c_int, fp]
    L.orc_bench_env_steps.restype = ctypes.c_double
    L.orc_bench_env_steps.argtypes = [vp, ctypes.c_int, ctypes.c_uint, ip]
    return L


# collision mode of the oracle envs the tests create unless they say otherwise: 1 = member-vs-member contacts (the
# reference's behaviour), 0 = floor contacts only
SELF_COLLISION_DEFAULT = 1


def gjk_query(ptsA, scaleA, xfA, ptsB, scaleB, xfB, max_dist2=1e18, lib=None):
    """One narrowphase query between two convex hulls (unscaled points, scaling, (basis rows, origin))."""
    L = lib or load()
    a = np.ascontiguousarray(ptsA, np.float32).reshape(-1, 3)
    b = np.ascontiguousarray(ptsB, np.float32).reshape(-1, 3)
    sa = np.ascontiguousarray(scaleA, np.float32)
    sb = np.ascontiguousarray(scaleB, np.float32)
    xa = np.ascontiguousarray(np.concatenate([np.asarray(xfA[0], np.float32).ravel(), np.asarray(xfA[1], np.float32)]))
    xb = np.ascontiguousarray(np.concatenate([np.asarray(xfB[0], np.float32).ravel(), np.asarray(xfB[1], np.float32)]))
    out = np.zeros(14, np.float32)
    L.orc_gjk_query(a.ctypes.data_as(fp), len(a), sa.ctypes.data_as(fp), xa.ctypes.data_as(fp), b.ctypes.data_as(fp), len(b),
                    sb.ctypes.data_as(fp), xb.ctypes.data_as(fp), ctypes.c_float(max_dist2), out.ctypes.data_as(fp))
    return dict(has=bool(out[0]), normal=out[1:4].copy(), point_b=out[4:7].copy(), distance=float(out[7]), iterations=int(out[8]),
                degenerate=int(out[9]), method=int(out[10]), used_penetration=bool(out[11]), ccd_status=int(out[12]), ccd_iterations=int(out[13]))


def epa_query(ptsA, scaleA, xfA, ptsB, scaleB, xfB, lib=None):
    """btGjkEpaPenetrationDepthSolver::calcPenDepth on two hulls (oracle/orc_epa.cpp), with its diagnostics."""
    L = lib or load()
    a = np.ascontiguousarray(ptsA, np.float32).reshape(-1, 3)
    b = np.ascontiguousarray(ptsB, np.float32).reshape(-1, 3)
    sa = np.ascontiguousarray(scaleA, np.float32)
    sb = np.ascontiguousarray(scaleB, np.float32)
    xa = np.ascontiguousarray(np.concatenate([np.asarray(xfA[0], np.float32).ravel(), np.asarray(xfA[1], np.float32)]))
    xb = np.ascontiguousarray(np.concatenate([np.asarray(xfB[0], np.float32).ravel(), np.asarray(xfB[1], np.float32)]))
    out = np.zeros(15, np.float32)
    L.orc_epa_query(a.ctypes.data_as(fp), len(a), sa.ctypes.data_as(fp), xa.ctypes.data_as(fp), b.ctypes.data_as(fp), len(b),
                    sb.ctypes.data_as(fp), xb.ctypes.data_as(fp), out.ctypes.data_as(fp))
    return dict(penetrating=bool(out[0]), normal=out[1:4].copy(), witness_a=out[4:7].copy(), witness_b=out[7:10].copy(),
                distance=float(out[10]), gjk_iterations=int(out[11]), epa_status=int(out[12]), epa_iterations=int(out[13]),
                epa_vertices=int(out[14]))


def set_penetration_solver(which, lib=None):
    """0 = EPA (the reference's configuration, default), 1 = the sampled-direction solver of rounds 2-3."""
    (lib or load()).orc_set_penetration_solver(int(which))


def set_ccd_pretest(on, lib=None):
    """1 = run the libccd-derived intersection pre-test of btGjkPairDetector (bullet3 >= 2.88) in front of every query"""
    (lib or load()).orc_set_ccd_pretest(int(on))


def set_floor_as_hull(on, lib=None):
    """measurement switch: 1 = the floor as the reference builds it (cube hull scaled (1000, 1, 1000)) through GJK, 0 = plane (default)"""
    (lib or load()).orc_set_floor_as_hull(int(on))


class OracleEnv:
    """One scalar oracle environment (reference semantics: reset() / do_step(action))."""

    def __init__(self, seed=1234, skeleton=SKEL, initial_remaining_seconds=1.0, max_episode_seconds=30.0,
                 target_velocity=0.5, minimal_velocity=0.1, reset_frames=30, lib=None, env_kind=0, self_collision=SELF_COLLISION_DEFAULT):
        self.L = lib or load()
        self.self_collision = int(self_collision)
        self.h = self.L.orc_env_create_ex(skeleton.encode(), seed, initial_remaining_seconds, max_episode_seconds,
                                          target_velocity, minimal_velocity, reset_frames, env_kind, self.self_collision)
        if not self.h:
            raise RuntimeError(self.L.orc_last_error().decode())
        self.obs_dim = self.L.orc_env_obs_dim(self.h)
        self.act_dim = self.L.orc_env_act_dim(self.h)
        self.nb = self.L.orc_env_num_bodies(self.h)
        self.nm = self.L.orc_env_num_members(self.h)
        self._obs = np.zeros(self.obs_dim, np.float32)
        self._r = ctypes.c_float()
        self._d = ctypes.c_int()

    def __del__(self):
        try:
            if self.h:
                self.L.orc_env_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _out(self):
        return self._obs.copy(), float(self._r.value), bool(self._d.value)

    def reset(self):
        self.L.orc_env_reset(self.h, self._obs.ctypes.data_as(fp), ctypes.byref(self._r), ctypes.byref(self._d))
        return self._out()

    def do_step(self, action):
        a = np.ascontiguousarray(action, np.float32)
        self.L.orc_env_step(self.h, a.ctypes.data_as(fp), self._obs.ctypes.data_as(fp), ctypes.byref(self._r), ctypes.byref(self._d))
        return self._out()

    def reset_begin(self):
        self.L.orc_env_reset_begin(self.h)

    def apply_action(self, action):
        a = np.ascontiguousarray(action, np.float32)
        self.L.orc_env_apply_action(self.h, a.ctypes.data_as(fp))

    def physics_step(self, n=1):
        for _ in range(n):
            self.L.orc_env_physics_step(self.h)

    def compute_step(self):
        self.L.orc_env_compute_step(self.h, self._obs.ctypes.data_as(fp), ctypes.byref(self._r), ctypes.byref(self._d))
        return self._out()

    def counters(self):
        c = np.zeros(5, np.int32)
        self.L.orc_env_get_counters(self.h, c.ctypes.data_as(ip))
        return dict(curr_step=int(c[0]), remaining_steps=int(c[1]), max_steps=int(c[2]), contacts=int(c[3]), joint_rows=int(c[4]))

    def set_counters(self, curr_step, remaining):
        self.L.orc_env_set_counters(self.h, int(curr_step), int(remaining))

    def state_size(self):
        return self.L.orc_env_state_size(self.h)

    def get_state(self):
        s = np.zeros(self.state_size(), np.float32)
        self.L.orc_env_get_state(self.h, s.ctypes.data_as(fp))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, np.float32)
        self.L.orc_env_set_state(self.h, s.ctypes.data_as(fp))

    def poses(self):
        p = np.zeros((self.nb, 7), np.float32)
        self.L.orc_env_get_poses(self.h, p.ctypes.data_as(fp))
        return p

    def pairs(self):
        n = self.L.orc_env_num_pairs(self.h)
        p = np.zeros((n, 2), np.int32)
        if n:
            self.L.orc_env_get_pairs(self.h, p.ctypes.data_as(ip))
        return p

    def pair_stats(self):
        c = np.zeros(5, np.int32)
        f = np.zeros(1, np.float32)
        self.L.orc_env_get_pair_stats(self.h, c.ctypes.data_as(ip), f.ctypes.data_as(fp))
        return dict(pair_contacts=int(c[0]), pair_tests=int(c[1]), gjk_iterations=int(c[2]), penetration_calls=int(c[3]),
                    live_pairs=int(c[4]), deepest=float(f[0]))

    def floor_stats(self):
        c = np.zeros(4, np.int32)
        self.L.orc_env_get_floor_stats(self.h, c.ctypes.data_as(ip))
        return dict(queries=int(c[0]), gjk_iterations=int(c[1]), penetration_calls=int(c[2]), ccd_intersect=int(c[3]))

    def pair_totals(self):
        """narrowphase work since creation, every physics step counted (settle steps of reset() too)"""
        c = (ctypes.c_longlong * 3)()
        self.L.orc_env_get_pair_totals(self.h, c)
        return dict(queries=int(c[0]), penetration_calls=int(c[1]), physics_steps=int(c[2]))

    def body_constants(self):
        p = np.zeros((self.nb, 19), np.float32)
        self.L.orc_env_get_body_constants(self.h, p.ctypes.data_as(fp))
        return p
