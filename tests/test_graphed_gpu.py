"""The score network's training call as captured hipGraphs (indm_amd/graphed.py) against the eager launch sequence: same
losses and same post-step parameters bit for bit (dropout 0), in fp32 and in bf16 (H16 storage) mode; with dropout the replays
draw a fresh mask per call from the device-resident seed."""
import numpy as np
import pytest
import torch

import golden_utils as gu
from indm_amd import graphed, hip, losses, ncsnpp, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mini(nf=64, dropout=0.0):
    """nf = 64: the last convolution (64 -> 3) takes the skinny-output path with its re-laid-out weight copy, and the attention
    block (128 channels, 64 tokens) the fused bf16 kernels -- both have to see the UPDATED weights on every replay"""
    cfg = gu.mini_config("mini_vp_nll")
    cfg.model.nf, cfg.model.dropout = nf, dropout
    cfg.device = torch.device(DEV)
    return cfg


def _run_steps(cfg, graph, dtype, nsteps=3, seed=7):
    torch.manual_seed(3)
    np.random.seed(3)
    cfg.compute_dtype = dtype
    graphed.enabled(graph)
    graphed.clear()
    score = utils.create_model(cfg, cfg.device)
    fl = create_flow_model(cfg).to(cfg.device)
    with torch.no_grad():
        for p in score.parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.05)
    state, fstate, _ = utils.build_states(cfg, score, fl, cfg.device)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); fl.train()
    outs = []
    try:
        for it in range(nsteps):
            g = torch.Generator().manual_seed(100 + it)
            x = (torch.rand(4, 3, cfg.data.image_size, cfg.data.image_size, generator=g) * 2 - 1).to(cfg.device)
            rng = DeviceRng(cfg.device, seed=seed + it, poisson_seed=11 + it)
            outs.append([v.clone() for v in step_fn(state, fstate, x, rng=rng)])
    finally:
        hip.set_compute_dtype("f32")
        graphed.enabled(True)
        graphed.clear()
    params = torch.cat([p.detach().flatten().float().cpu() for p in score.parameters()])
    return outs, params


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graphed_score_call_equals_the_eager_launch_sequence(dtype):
    cfg = _mini()
    eager, pe = _run_steps(cfg, False, dtype)
    graph, pg = _run_steps(cfg, True, dtype)
    for it, (a, b) in enumerate(zip(eager, graph)):
        for nm, u, v in zip(("loss", "score", "flow", "logp"), a, b):
            assert torch.equal(u, v), (dtype, it, nm, (u - v).abs().max())
    assert torch.equal(pe, pg), (pe - pg).abs().max()


def test_graphed_dropout_draws_a_fresh_mask_per_replay():
    cfg = _mini(dropout=0.3)
    cfg.compute_dtype = "f32"
    torch.manual_seed(5)
    score = utils.create_model(cfg, cfg.device)
    fl = create_flow_model(cfg).to(cfg.device)
    with torch.no_grad():
        for p in score.parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.05)
    state, _, _ = utils.build_states(cfg, score, fl, cfg.device)
    score.train()
    graphed.clear()
    x = torch.randn(4, cfg.data.image_size, cfg.data.image_size, 3, device=DEV, requires_grad=True)
    t = torch.rand(4, device=DEV) * 900 + 1

    class R:
        def __init__(self, s):
            self.s = s

        def dropout_seed(self):
            return self.s

    flat = score.__dict__["_flat"]
    o1 = graphed.score_call(score, flat, x, t, R(11)).clone()
    o2 = graphed.score_call(score, flat, x, t, R(12)).clone()
    o3 = graphed.score_call(score, flat, x, t, R(11)).clone()
    assert torch.equal(o1, o3) and not torch.equal(o1, o2)
    assert torch.isfinite(o1).all() and torch.isfinite(o2).all()
    # backward through the replay: gradients reach x and the flat gradient buffer
    state["optimizer"].zero_grad()
    graphed.score_call(score, flat, x, t, R(13)).square().sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and x.grad.abs().max() > 0
    assert flat.g.abs().max() > 0
    graphed.clear()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graphed_steps_equal_eager_on_the_full_size_networks(dtype):
    """two optimisation steps of the FULL-SIZE vp/CIFAR10/indm_nll networks (batch 2), eager vs graphed score call, same init and
    noise: identical losses and parameters -- the second step replays the graph on UPDATED weights (the skinny last convolution's
    re-laid-out weight copy and the bf16 packs must be refreshed inside the graph, not frozen at capture time)"""
    from indm_amd import configs
    res = {}
    try:
        for mode in ("eager", "graph"):
            graphed.enabled(mode == "graph")
            graphed.clear()
            cfg = configs.get_config("vp/CIFAR10/indm_nll")
            cfg.model.dropout = 0.0
            cfg.training.batch_size = 2
            cfg.device = torch.device(DEV)
            cfg.compute_dtype = dtype
            torch.manual_seed(0)
            score = utils.create_model(cfg, cfg.device)
            fl = create_flow_model(cfg).to(cfg.device)
            with torch.no_grad():
                for p in score.parameters():
                    if p.dim() > 1 and p.abs().max() < 1e-6:
                        p.copy_(torch.randn_like(p) * 0.02)
            state, fstate, _ = utils.build_states(cfg, score, fl, cfg.device)
            step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
            score.train(); fl.train()
            rec = []
            for it in range(2):
                g = torch.Generator().manual_seed(50 + it)
                x = (torch.rand(2, 3, 32, 32, generator=g) * 2 - 1).to(DEV)
                ls = step_fn(state, fstate, x, rng=DeviceRng(cfg.device, seed=9 + it, poisson_seed=3 + it))
                rec.append(([v.clone() for v in ls], state["optimizer"].flat.p.clone(), fstate["optimizer"].flat.p.clone()))
            res[mode] = rec
            del state, fstate, step_fn, score, fl
    finally:
        hip.set_compute_dtype("f32")
        graphed.enabled(True)
        graphed.clear()
    for it in range(2):
        e, g = res["eager"][it], res["graph"][it]
        for nm, a, b in zip(("loss", "score", "flow", "logp"), e[0], g[0]):
            assert torch.equal(a, b), (dtype, it, nm, a, b)
        assert torch.equal(e[1], g[1]) and torch.equal(e[2], g[2]), (dtype, it)
