"""The reference's own unit tests (evo_motion_networks/tests/src/) restated on this repo's host-side mirrors — the same parameter
grids and the same assertions, so a maintainer of the reference reads their suite here:

  test_functions.cpp:82-149        truncated normal sample / pdf / log-pdf / entropy: shapes kept, samples inside the bounds, everything
                                   finite, pdf > 0 — sizes {[1,2,3], [1000,1000], [6]} x min {-2,-1,-0.1} x max {0.1,1,2}
  test_linear_modules.cpp:14-147   ActorModule / CriticModule / QNetworkModule output shapes, mu in [-1,1], sigma > 0 — state {1,2} x
                                   hidden {1,2} x action {1,2,100} x batch {1,2}
  test_metrics.cpp:20-25           -> tests/test_metrics.py

The GPU side of the same suite (agents through act / done at tiny sizes, the fused kernels' mu / sigma ranges) is
tests/test_gpu_reference_suite.py."""
import itertools

import pytest
import torch

SIZES = [(1, 2, 3), (1000, 1000), (6,)]
BOUNDS = list(itertools.product((-2.0, -1.0, -0.1), (0.1, 1.0, 2.0)))


@pytest.fixture(scope="module")
def agent_mod(hip_lib):
    from evomotion_amd import agent
    return agent


def _mu_sigma(sizes, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    mu = torch.rand(sizes, generator=g) * (hi - lo) + lo
    sigma = torch.nn.functional.softplus(torch.rand(sizes, generator=g) * 60.0 - 30.0)
    return mu, sigma


@pytest.mark.parametrize("sizes", SIZES)
@pytest.mark.parametrize("lo,hi", BOUNDS)
def test_truncated_normal_sample_pdf_entropy(agent_mod, sizes, lo, hi):
    mu, sigma = _mu_sigma(sizes, lo, hi, 7)
    out = agent_mod.truncated_normal_sample(mu, sigma, lo, hi)                        # TestSample
    assert tuple(out.shape) == tuple(sizes)
    assert bool((out >= lo).all()) and bool((out <= hi).all())
    assert bool(torch.isfinite(out).all())
    log_pdf = agent_mod.truncated_normal_log_pdf(out, mu, sigma, lo, hi)              # TestPDF (pdf = exp(log_pdf))
    pdf = torch.exp(log_pdf)
    assert tuple(log_pdf.shape) == tuple(sizes)
    assert bool((pdf > 0).all()) and bool(torch.isfinite(pdf).all()) and bool(torch.isfinite(log_pdf).all())
    ent = agent_mod.truncated_normal_entropy(mu, sigma, lo, hi)                        # TestEntropy
    assert tuple(ent.shape) == tuple(sizes) and bool(torch.isfinite(ent).all())


@pytest.mark.parametrize("state_space,hidden_size,action_space,batch_size", list(itertools.product((1, 2), (1, 2), (1, 2, 100), (1, 2))))
def test_linear_modules(agent_mod, state_space, hidden_size, action_space, batch_size):
    from evomotion_amd.sac import QNetworkModule
    torch.manual_seed(3)
    actor = agent_mod.ActorModule([state_space], [action_space], hidden_size)
    critic = agent_mod.CriticModule([state_space], hidden_size)
    q = QNetworkModule([state_space], [action_space], hidden_size)
    for shape in ((), (batch_size,)):                                                   # TestActorModule / TestBatchedActorModule ...
        state = torch.randn(*shape, state_space)
        mu, sigma = actor(state)
        assert tuple(mu.shape) == tuple(sigma.shape) == (*shape, action_space)
        assert bool((mu >= -1).all()) and bool((mu <= 1).all()) and bool((sigma > 0).all())
        value = critic(state)
        value = value[0] if isinstance(value, (tuple, list)) else value
        assert tuple(value.shape) == (*shape, 1)
        qv = q(state, torch.randn(*shape, action_space))
        qv = qv[0] if isinstance(qv, (tuple, list)) else qv
        assert tuple(qv.shape) == (*shape, 1) and bool(torch.isfinite(qv).all())
