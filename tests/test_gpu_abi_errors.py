"""Argument errors at the C-ABI boundary on the GPU box: they come back as error codes with a message (std::invalid_argument ->
ValueError in the Python mirror), never as a launch with a bad pointer."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def test_argument_errors_are_codes_with_messages():
    import torch
    from evomotion_amd import VecRobotWalk, FusedActorCritic
    from evomotion_amd._lib import lib, EvmError
    with pytest.raises(ValueError):
        VecRobotWalk(0)
    env = VecRobotWalk(5)                                   # a ragged batch: one partial tile
    env.reset()
    with pytest.raises(ValueError):
        env.do_step(torch.zeros(5, 11))                     # wrong action width
    with pytest.raises(ValueError):
        env.do_step(torch.zeros(4, 12))                     # wrong batch
    null = ctypes.c_void_p()
    for call in (lambda: lib.evm_env_step(env._h, null, null, null, null, null),
                 lambda: lib.evm_env_step_autoreset(env._h, null, null, null, null, null, null),
                 lambda: lib.evm_env_reset(env._h, null, null, null, null, null),
                 lambda: lib.evm_env_get_body_poses(env._h, null, null)):
        assert call() == -1 and b"null" in lib.evm_last_error()
    st = env.do_step(torch.zeros(5, 12))                    # the env is still usable afterwards
    assert torch.isfinite(st.state).all()
    with pytest.raises((ValueError, EvmError)):
        FusedActorCritic(371, 17, 256, 0)                   # 2 * 17 head outputs do not fit the 32 head columns
    with pytest.raises((ValueError, EvmError)):
        FusedActorCritic(385, 12, 256, 0)                   # wider than the padded 384-input first layer
    with pytest.raises((ValueError, EvmError)):
        FusedActorCritic(371, 12, 128, 0)                   # the fused kernels are built for hidden_size 256
