"""No kernel stores outside the tensor it was given: every device allocation of the package is padded with sentinel
guard bands (tests/guard_alloc.py) while training steps, the flow's evaluation paths and a sampler step run."""
import pytest
import torch

from indm_amd import configs, hip, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.losses import flow_forward
from indm_amd.rng import DeviceRng

import guard_alloc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def guards():
    guard_alloc.install()
    try:
        yield guard_alloc
    finally:
        guard_alloc.uninstall()
        hip.set_compute_dtype("f32")


def _setup(dtype, B):
    dev = torch.device(DEV)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.device = dev
    cfg.training.batch_size = B
    cfg.compute_dtype = dtype
    torch.manual_seed(0)
    score = utils.create_model(cfg, dev)
    flow = create_flow_model(cfg).to(dev)
    state, fstate, _ = utils.build_states(cfg, score, flow, dev)
    return cfg, dev, score, flow, state, fstate


def test_guard_bands_catch_an_overrun(guards):
    """positive control: a store one element past a package allocation is reported, an in-bounds store is not"""
    t = hip.torch.empty(1000, dtype=torch.float32, device=DEV)
    t.fill_(1.0)
    assert guards.check() == []
    t = hip.torch.empty((10, 100), dtype=torch.float32, device=DEV)
    torch.as_strided(t, (1001,), (1,)).fill_(2.0)
    bad = guards.check()
    assert len(bad) == 1 and "PAST the end" in bad[0], bad
    y = hip.axpby(torch.ones(7, device=DEV), None, 2.0, 0.0)     # a real op's output is a guarded allocation
    assert guards.count() >= 1 and guards.check() == [] and float(y.sum()) == 14.0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dtype,B", [("f32", 4), ("bf16", 4), ("f32", 3), ("bf16", 5), ("f32", 16)])
def test_training_step_writes_stay_in_bounds(guards, dtype, B):
    cfg, dev, score, flow, state, fstate = _setup(dtype, B)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); flow.train()
    rng = DeviceRng(dev, seed=21, poisson_seed=22)
    g = torch.Generator().manual_seed(23)
    for s in range(3):
        x = (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).to(dev)
        out = step_fn(state, fstate, x, rng=rng)
        assert guards.count() > 1000      # the step's outputs and scratch really are guarded allocations
        bad = guards.check()
        assert not bad, f"step {s}: " + " | ".join(bad[:6])
        assert all(torch.isfinite(o).all() for o in out)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dtype,B", [("f32", 3), ("bf16", 4)])
def test_flow_eval_paths_write_in_bounds(guards, dtype, B):
    cfg, dev, score, flow, state, fstate = _setup(dtype, B)
    hip.set_compute_dtype(dtype)
    flow.eval()
    x = (torch.rand(B, 3, 32, 32, generator=torch.Generator().manual_seed(5)) * 2 - 1).to(dev)
    with torch.no_grad():
        z, ld = flow_forward(cfg, flow, x, reverse=False, rng=DeviceRng(dev, seed=3))
        bad = guards.check()
        assert not bad, "forward: " + " | ".join(bad[:6])
        xr, _ = flow_forward(cfg, flow, z, reverse=True, rng=DeviceRng(dev, seed=4))
        bad = guards.check()
        assert not bad, "inverse: " + " | ".join(bad[:6])
    assert torch.isfinite(xr).all()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,pred,corr", [("ve/CIFAR10/indm", "reverse_diffusion", "langevin"),
                                            ("vp/CIFAR10/indm_nll", "euler_maruyama", "none")])
def test_sampler_and_likelihood_updates_write_in_bounds(guards, name, pred, corr):
    """the block-level sampler entry points (noise drawn inside the update kernel) and the Hutchinson divergence under
    guard bands: three PC steps + the flow inverse, then one NELBO evaluation"""
    from indm_amd import likelihood, sampling
    dev = torch.device(DEV)
    cfg = configs.get_config(name)
    cfg.device = dev
    cfg.sampling.method, cfg.sampling.predictor, cfg.sampling.corrector = "pc", pred, corr
    cfg.sampling.num_scales = 3
    torch.manual_seed(0)
    score = utils.create_model(cfg, dev).eval()
    flow = create_flow_model(cfg).to(dev).eval()
    sde = sde_lib.get_sde(cfg)
    inv = utils.get_data_inverse_scaler(cfg)
    fn = sampling.get_sampling_fn(cfg, sde, (2, 3, 32, 32), inv, cfg.sampling.truncation_time, device=dev)
    before, after, _ = fn(score, flow, rng=DeviceRng(dev, seed=9), graph=False)
    bad = guards.check()
    assert not bad, "sampler: " + " | ".join(bad[:6])
    assert torch.isfinite(before).all() and torch.isfinite(after).all()
    x = (torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(5)) * 2 - 1).to(dev)
    nelbo = likelihood.get_elbo_fn(cfg, sde, inverse_scaler=inv)(score, flow, x, rng=DeviceRng(dev, seed=10))
    bad = guards.check()
    assert not bad, "nelbo: " + " | ".join(bad[:6])
    assert all(torch.isfinite(torch.as_tensor(v)).all() for v in nelbo)
