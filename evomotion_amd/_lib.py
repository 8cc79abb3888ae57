"""ctypes binding of libevomotion_hip.so (the C ABI declared in include/evomotion.h).

The HIP library is the product; there is no CPU or PyTorch fallback.  Importing this module when the
library has not been built raises ImportError with the build instruction."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libevomotion_hip.so")
DEFAULT_SKELETON = os.path.join(_HERE, "data", "robot_walk_spider.skel")

EVM_OK = 0


class EvmEnvParams(ctypes.Structure):
    _fields_ = [
        ("initial_remaining_seconds", ctypes.c_float),
        ("max_episode_seconds", ctypes.c_float),
        ("target_velocity", ctypes.c_float),
        ("minimal_velocity", ctypes.c_float),
        ("reset_frames", ctypes.c_int),
        ("env_kind", ctypes.c_int),
        ("self_collision", ctypes.c_int),
    ]


class EvmError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C evomotion_amd/csrc`). evomotion_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    vp, cp, fp = ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int)
    lib.evm_last_error.restype = cp
    lib.evm_env_default_params.argtypes = [ctypes.POINTER(EvmEnvParams)]
    lib.evm_env_default_params_for.argtypes = [cp, ctypes.POINTER(EvmEnvParams)]
    lib.evm_env_create.argtypes = [cp, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(EvmEnvParams), ctypes.POINTER(vp)]
    lib.evm_env_destroy.argtypes = [vp]
    lib.evm_env_destroy.restype = None
    lib.evm_env_spaces.argtypes = [vp, ip, ip]
    lib.evm_env_counts.argtypes = [vp, ip, ip, ip, ip]
    lib.evm_env_pairs.argtypes = [vp, ip, ip]
    lib.evm_env_debug_pair_counts.argtypes = [vp, ip]
    lib.evm_env_reset.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.evm_env_step.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.evm_env_step_autoreset.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.evm_env_get_body_poses.argtypes = [vp, vp, vp]
    lib.evm_env_state_size.argtypes = [vp]
    lib.evm_env_get_state.argtypes = [vp, fp]
    lib.evm_env_set_state.argtypes = [vp, fp]
    lib.evm_env_debug_reset_begin.argtypes = [vp, vp]
    lib.evm_env_debug_physics_steps.argtypes = [vp, ctypes.c_int, vp]
    lib.evm_env_get_body_constants.argtypes = [vp, fp]
    lib.evm_env_get_diagnostics.argtypes = [vp, vp, vp]
    lib.evm_skeleton_probe.argtypes = [cp, ip, fp]
    lib.evm_skeleton_digest.argtypes = [cp, ctypes.POINTER(ctypes.c_ulonglong)]
    lib.evm_skeleton_schedule.argtypes = [cp, ip, ip, ip, ctypes.c_int]
    lib.evm_skeleton_group_schedule.argtypes = [cp, ctypes.c_int, ip, ip, ctypes.c_int]
    lib.evm_skeleton_group_schedule_ex.argtypes = [cp, ctypes.c_int, ctypes.c_int, ip, ip, ctypes.c_int]
    lib.evm_env_get_residual.argtypes = [vp, fp, ctypes.c_int, vp]
    lib.evm_env_get_errors.argtypes = [vp, ip, ctypes.c_int, vp]
    lib.evm_env_get_pair_counters.argtypes = [vp, ip, ctypes.c_int, vp]
    lib.evm_env_get_speculation_counters.argtypes = [vp, ip, ctypes.c_int, vp]
    lib.evm_env_get_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
    lib.evm_env_clear_stats.argtypes = [vp]
    lib.evm_env_get_stamps.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong)]
    lib.evm_policy_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    lib.evm_policy_destroy.argtypes = [vp]
    lib.evm_policy_destroy.restype = None
    lib.evm_policy_param_counts.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    lib.evm_policy_set_weights.argtypes = [vp, fp, ctypes.c_size_t, fp, ctypes.c_size_t]
    lib.evm_policy_set_weights_device.argtypes = [vp, vp, vp, vp]
    lib.evm_policy_forward.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_uint64, vp, vp, vp, vp, vp, vp]
    lib.evm_policy_set_tile_rows.argtypes = [vp, ctypes.c_int]
    lib.evm_policy_timing_begin.argtypes = [vp]
    lib.evm_policy_timing_end.argtypes = [vp, vp, fp, ip]
    lib.evm_replay_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    lib.evm_replay_destroy.argtypes = [vp]
    lib.evm_replay_destroy.restype = None
    lib.evm_replay_push.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.evm_replay_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong), vp]
    lib.evm_replay_sample.argtypes = [vp, ctypes.c_int, ctypes.c_uint64, vp, vp, vp, vp, vp, vp, vp]
    lib.evm_replay_timing_begin.argtypes = [vp]
    lib.evm_replay_timing_end.argtypes = [vp, vp, fp, ip, fp, ip]
    dp = ctypes.POINTER(ctypes.c_double)
    lib.evm_ppo_create.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.evm_ppo_destroy.argtypes = [vp]
    lib.evm_ppo_destroy.restype = None
    lib.evm_ppo_set_params.argtypes = [vp, vp, vp, ctypes.c_int, vp]
    lib.evm_ppo_copy.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.evm_ppo_adam_step.argtypes = [vp, ctypes.c_int, ctypes.c_int, ip]
    lib.evm_ppo_grad_buffer.argtypes = [vp, vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    lib.evm_ppo_gae_merge.argtypes = [vp, vp, ctypes.c_int, vp, vp]
    lib.evm_ppo_gae.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp]
    lib.evm_ppo_gae_normalize.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp]
    lib.evm_ppo_grads.argtypes = [vp, ctypes.c_size_t, vp, vp, vp, vp, vp, vp, ctypes.c_double, ctypes.c_float, ctypes.c_float,
                                  ctypes.c_float, ctypes.c_int, vp]
    pvp = ctypes.POINTER(vp)
    lib.evm_ppo_select_rows.argtypes = [vp, ctypes.c_size_t, vp, vp, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_size_t), pvp, pvp, pvp, pvp,
                                        pvp, pvp, vp]
    lib.evm_ppo_apply.argtypes = [vp, ctypes.c_float, ctypes.c_float, vp]
    lib.evm_ppo_losses.argtypes = [vp, dp, dp, vp]
    lib.evm_ppo_timing.argtypes = [vp, ctypes.c_int, fp, ip]
    lib.evm_q_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(vp)]
    lib.evm_q_destroy.argtypes = [vp]
    lib.evm_q_destroy.restype = None
    lib.evm_q_param_count.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t)]
    lib.evm_q_copy.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.evm_q_adam_step.argtypes = [vp, ctypes.c_int, ctypes.c_int, ip]
    lib.evm_q_forward.argtypes = [vp, ctypes.c_uint, ctypes.c_size_t, vp, vp, ctypes.POINTER(vp), vp]
    lib.evm_q_grads.argtypes = [vp, ctypes.c_size_t, vp, vp, vp, vp]
    lib.evm_q_apply.argtypes = [vp, ctypes.c_float, vp]
    lib.evm_q_soft_update.argtypes = [vp, ctypes.c_float, vp]
    lib.evm_q_losses.argtypes = [vp, vp, vp]
    lib.evm_q_action_grad.argtypes = [vp, ctypes.c_size_t, vp, vp, vp, vp, vp]
    lib.evm_sac_sample.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp]
    lib.evm_sac_actor_grad.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.evm_ppo_actor_forward.argtypes = [vp, ctypes.c_size_t, vp, vp, vp, vp]
    lib.evm_ppo_actor_backward.argtypes = [vp, ctypes.c_size_t, vp, vp, vp]
    lib.evm_ppo_actor_apply.argtypes = [vp, ctypes.c_float, vp]
    lib.evm_sac_target_q.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, ctypes.c_float, vp, vp]
    lib.evm_sac_entropy_step.argtypes = [ctypes.c_int, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp, vp]
    lib.evm_env_timing_begin.argtypes = [vp, vp]
    lib.evm_env_timing_end.argtypes = [vp, vp, fp, ip]
    lib.evm_env_timing_end_detail.argtypes = [vp, vp, fp, ip, fp]
    return lib


lib = _load()


def check(rc):
    if rc != EVM_OK:
        msg = lib.evm_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)  # std::invalid_argument in the reference
        raise EvmError(f"evomotion error {rc}: {msg}")
