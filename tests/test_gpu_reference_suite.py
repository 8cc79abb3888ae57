"""The GPU half of the reference's own unit tests (evo_motion_networks/tests/src/), restated on the PRODUCT path — the fused HIP
kernels behind the C ABI — with the reference's parameter grids and assertions:

  test_agents.cpp:153-176          TestPpoGae: batch_size episodes of two act() calls and a done(), reward 1, epoch 1, train_every 1,
                                   replay_buffer_size 2 batch_size; every action has rank 1, action_space entries, no NaN, -1 <= a <= 1
  test_agents.cpp:45-68            TestSoftActorCritic: five episodes of 2 batch_size act() calls with random rewards, epoch 2
  test_linear_modules.cpp:14-50    the actor's (mu, sigma) and the critic's value from the fused forward kernel: shapes, mu in [-1,1],
                                   sigma > 0, one row and batches
  test_functions.cpp:82-139        truncated normal sample / log-pdf from the device kernel (bounds -1, 1: the ones this path uses) with
                                   sigma = softplus(U(-30, 30)) — 1e-13 .. 30

The grid is the reference's (state, action, batch_size, train_every) in {2, 3}^4; its fifth axis, hidden_size in {2, 3}, is fixed to 256:
the fused kernels are built for the hidden size of BASELINE's configuration and refuse others with a message (checked here)."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GRID = list(itertools.product((2, 3), (2, 3), (2, 3), (2, 3)))


def _check_action(action, action_space):
    import torch
    assert action.dim() == 1 and action.shape[0] == action_space
    assert not bool(torch.isnan(action).any())
    assert bool((action >= -1.0).all()) and bool((action <= 1.0).all())


@pytest.mark.parametrize("state_space,action_space,batch_size,train_every", GRID)
def test_ppo_gae(state_space, action_space, batch_size, train_every):
    import torch
    from evomotion_amd import PpoGaeAgent
    g = torch.Generator().manual_seed(1)
    agent = PpoGaeAgent(1234, [state_space], [action_space], 256, 0.99, 0.95, 0.2, 0.01, 0.5, 1, batch_size, 1, batch_size * 2, 1e-3, 0.5)
    for i in range(batch_size):
        for j in range(2):
            _check_action(agent.act(torch.randn(state_space, generator=g), 1.0), action_space)
        agent.done(torch.randn(state_space, generator=g), 1.0)
    # (commented out in the reference: `for m in get_metrics(): m.loss() != 0`)  here: it trained, the meters are filled and finite
    meters = agent.get_metrics()
    assert agent.curr_train_step >= 1 and all(np.isfinite(m.loss()) for m in meters) and meters[2].loss() == 2.0


@pytest.mark.parametrize("state_space,action_space,batch_size,train_every", GRID)
def test_soft_actor_critic(state_space, action_space, batch_size, train_every):
    import torch
    from evomotion_amd import SoftActorCriticAgent
    g = torch.Generator().manual_seed(2)
    agent = SoftActorCriticAgent(1234, [state_space], [action_space], 256, 256, batch_size, 2, 1e-3, 0.9, 0.005, 128, train_every)
    for i in range(5):
        for j in range(batch_size * 2):
            _check_action(agent.act(torch.randn(state_space, generator=g), float(torch.randn(1, generator=g))), action_space)
        agent.done(torch.randn(state_space, generator=g), float(torch.randn(1, generator=g)))
    meters = agent.get_metrics()
    assert agent.curr_train_step >= 2 and all(np.isfinite(m.loss()) for m in meters) and meters[4].loss() == 2.0 * batch_size


def test_other_hidden_sizes_are_refused_with_a_message():
    from evomotion_amd import FusedActorCritic
    from evomotion_amd._lib import EvmError
    for hidden in (2, 3):
        with pytest.raises(EvmError) as e:
            FusedActorCritic(2, 2, hidden, 0)
        assert "hidden_size = 256" in str(e.value)


@pytest.mark.parametrize("state_space,action_space,batch_size", list(itertools.product((1, 2), (1, 2, 16), (1, 2, 100))))
def test_fused_modules(state_space, action_space, batch_size):
    """TestActorModule / TestBatchedActorModule / TestCriticModule on the fused kernel (action_space 100 of the reference's grid
    exceeds the head tile of this kernel — 2 A <= 32 — and is refused, below)"""
    import torch
    from evomotion_amd import ActorModule, CriticModule, FusedActorCritic
    torch.manual_seed(5)
    pol = FusedActorCritic(state_space, action_space, 256, 0)
    pol.load_modules(ActorModule([state_space], [action_space], 256).cuda(), CriticModule([state_space], 256).cuda())
    state = torch.randn(batch_size, state_space, device="cuda")
    action, logp, value, mu, sigma = pol.forward(state, seed=3, want_dist=True)
    torch.cuda.synchronize()
    for t in (action, logp, mu, sigma):
        assert tuple(t.shape) == (batch_size, action_space) and bool(torch.isfinite(t).all())
    assert tuple(value.shape) == (batch_size,) and bool(torch.isfinite(value).all())
    assert bool((mu >= -1).all()) and bool((mu <= 1).all()) and bool((sigma > 0).all())
    assert bool((action >= -1).all()) and bool((action <= 1).all())


def test_action_space_beyond_the_head_tile_is_refused():
    from evomotion_amd import FusedActorCritic
    with pytest.raises(ValueError) as e:      # EVM_E_INVALID <-> std::invalid_argument
        FusedActorCritic(2, 100, 256, 0)
    assert "unsupported state/action size" in str(e.value)


@pytest.mark.parametrize("rows,A", [(6, 1), (1000, 12), (4096, 16)])
def test_truncated_normal_device_kernel(rows, A):
    """TestSample / TestPDF on evm_sac_sample (sample + summed log-pdf): samples inside [-1, 1], everything finite, pdf > 0, over the
    reference's sigma range softplus(U(-30, 30))"""
    import torch
    from evomotion_amd.qnet import sac_sample
    g = torch.Generator(device="cuda").manual_seed(11)
    mu = torch.rand(rows, A, device="cuda", generator=g) * 2.0 - 1.0
    sigma = torch.nn.functional.softplus(torch.rand(rows, A, device="cuda", generator=g) * 60.0 - 30.0)
    u = torch.rand(rows, A, device="cuda", generator=g)
    action, logp_sum = sac_sample(mu.contiguous(), sigma.contiguous(), u.contiguous())
    torch.cuda.synchronize()
    assert tuple(action.shape) == (rows, A) and tuple(logp_sum.shape) == (rows,)
    assert bool((action >= -1).all()) and bool((action <= 1).all()) and bool(torch.isfinite(action).all())
    assert bool(torch.isfinite(logp_sum).all()) and bool((torch.exp(logp_sum.double()) > 0).all())
