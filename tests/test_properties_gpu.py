"""Size-independent properties at the BENCHMARK's full sizes (batch 128, 512-wide flow, production tile shapes), where
an element-by-element oracle comparison would take minutes on the CPU:
  * adjointness  <dy, conv(x, w)> == <dgrad(dy, w), x> == <wgrad(x, dy), w>   ties the forward, data-grad and
    weight-grad kernels (8-wave, big-tile, split-K paths) to each other;
  * linearity and cross-path consistency of the long GEMMs (big-tile vs 8-wave kernels, gated vs pre-multiplied);
  * flow forward -> fixed-point inverse round trip of the full 16-16 residual flow."""
import pytest
import torch

from indm_amd import configs, hip
from indm_amd.flow import create_flow_model, flow_forward
from indm_amd.rng import DeviceRng

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _r(*shape, seed=0, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.rand(*shape, generator=g, device=DEV) * 2 - 1) * scale


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _close(a, b, rel, name):
    assert abs(a - b) <= rel * max(abs(a), abs(b), 1e-30), (name, a, b)


@pytest.mark.parametrize("N,H,Ci,Co,k", [(128, 32, 128, 128, 3), (128, 16, 256, 256, 3), (128, 16, 512, 256, 3),
                                         (128, 32, 256, 128, 1), (128, 32, 3, 512, 3), (128, 32, 512, 3, 3)])
def test_conv_adjointness_full_size(N, H, Ci, Co, k):
    p = k // 2
    x, w, dy = _r(N, H, H, Ci, seed=1), _r(k, k, Ci, Co, seed=2, scale=0.1), _r(N, H, H, Co, seed=3)
    y = hip.conv2d(x, w, 1, p)
    dx = hip.conv2d_dgrad(dy, w, (H, H), 1, p)
    dw = hip.conv2d_wgrad(x, dy, k, k, 1, p)
    ref = _dot(dy, y)
    _close(_dot(dx, x), ref, 2e-5, "dgrad adjoint")
    _close(_dot(dw, w), ref, 2e-5, "wgrad adjoint")


def test_flow_gemm_paths_full_size():
    M = 128 * 1024
    a, g, w = _r(M, 512, seed=1), _r(M, 512, seed=2), _r(512, 512, seed=3, scale=0.05)
    wt = w.t().contiguous()
    nn = hip.gemm(a, w)                          # big-tile 256x128, B row-contiguous
    nt = hip.gemm(a, wt, trans_b=True)           # big-tile 256x128, B k-contiguous
    assert (nn - nt).abs().max().item() <= 2e-5 * nn.abs().max().item()
    gated = hip.gemm(a, wt, trans_b=True, gate=g)            # 8-wave kernel, gate in the loader
    pre = hip.gemm(hip.mul(a, g), wt, trans_b=True)          # big-tile kernel on the pre-multiplied operand
    assert (gated - pre).abs().max().item() <= 2e-5 * pre.abs().max().item()
    a2 = _r(M, 512, seed=4)
    lin = hip.gemm(hip.axpby(a, a2, 1.0, -2.0), w)
    assert (lin - (nn - 2.0 * hip.gemm(a2, w))).abs().max().item() <= 5e-5 * nn.abs().max().item()
    # weight-grad GEMM (big-tile split-K) against the adjoint identity  <dc, a w> = <a^T dc, w>
    dc = _r(M, 512, seed=5)
    dw = hip.gemm(a, dc, trans_a=True, splitk=True)
    _close(_dot(dw, w), _dot(dc, nn), 2e-5, "1x1 wgrad adjoint")
    dwg = hip.gemm(a, dc, trans_a=True, splitk=True, gate=g)
    _close(_dot(dwg, w), _dot(dc, hip.gemm(a, w, gate=g)), 2e-5, "gated 1x1 wgrad adjoint")


@pytest.mark.timeout(900)
def test_full_size_flow_round_trip():
    torch.manual_seed(3)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.device = torch.device(DEV)
    fl = create_flow_model(cfg).to(DEV).eval()
    x = _r(128, 3, 32, 32, seed=9)
    with torch.no_grad():
        z, _ = flow_forward(cfg, fl, x, log_det=None, reverse=False, rng=DeviceRng(DEV, 1))
        iters = []
        xr, _ = flow_forward(cfg, fl, z, log_det=None, reverse=True, rng=DeviceRng(DEV, 2), iters=iters)
    assert len(iters) == 32 and max(iters) < 1000
    # the reference's own stopping rule is (x - x_prev)^2 / (1e-5 + 1e-5 |y|) < 1 per element, per block
    assert (xr - x).abs().max().item() < 5e-2 and (xr - x).abs().mean().item() < 5e-3, \
        ((xr - x).abs().max().item(), (xr - x).abs().mean().item())


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_training_on_a_fixed_batch_reduces_the_objective(dtype):
    """End-to-end sanity of forward + hand-written backwards + fused optimiser at the full-size configuration: 30
    optimisation steps on ONE batch must lower the per-image objective (fresh noise, times and series lengths every
    step, so the comparison is between 6-step means) and keep everything finite."""
    from indm_amd import losses, sde_lib, utils
    dev = torch.device(DEV)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.device = dev
    cfg.training.batch_size = 8
    cfg.compute_dtype = dtype
    torch.manual_seed(0)
    score = utils.create_model(cfg, dev)
    flow = create_flow_model(cfg).to(dev)
    state, fstate, _ = utils.build_states(cfg, score, flow, dev)
    try:
        step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
        score.train(); flow.train()
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(8, 3, 32, 32, generator=g) * 2 - 1).to(dev)
        rng = DeviceRng(dev, seed=3, poisson_seed=5)
        hist = []
        for _ in range(30):
            out = step_fn(state, fstate, x, rng=rng)
            assert all(torch.isfinite(o).all() for o in out)
            hist.append(float(out[0].mean()))
        first, last = sum(hist[:6]) / 6, sum(hist[-6:]) / 6
        assert last < first - 0.05 * abs(first), (first, last, hist)
        assert state["step"] == 30 and fstate["step"] == 30
    finally:
        hip.set_compute_dtype("f32")


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_no_memory_growth_over_many_steps(dtype):
    """60 training steps (full-size networks, batch 4, varying Poisson series lengths): the memory held between steps
    and the sizes of the host-side caches (pointer tables, re-laid-out weights, packed bf16 weights) stay bounded."""
    import gc
    from indm_amd import losses, sde_lib, utils
    dev = torch.device(DEV)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.device = dev
    cfg.training.batch_size = 4
    cfg.compute_dtype = dtype
    torch.manual_seed(0)
    score = utils.create_model(cfg, dev)
    flow = create_flow_model(cfg).to(dev)
    state, fstate, _ = utils.build_states(cfg, score, flow, dev)
    try:
        step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
        score.train(); flow.train()
        rng = DeviceRng(dev, seed=11)
        g = torch.Generator().manual_seed(2)
        marks = {}
        for it in range(60):
            x = (torch.rand(4, 3, 32, 32, generator=g) * 2 - 1).to(dev)
            out = step_fn(state, fstate, x, rng=rng)
            del out, x
            if it in (19, 39, 59):
                torch.cuda.synchronize()
                gc.collect()
                marks[it] = torch.cuda.memory_allocated()
    finally:
        hip.set_compute_dtype("f32")
    assert abs(marks[59] - marks[19]) <= 8 << 20, marks      # a few MB of cached tables at most
    assert abs(marks[59] - marks[39]) <= 8 << 20, marks
    assert len(hip._tables) <= 256 and len(hip._packed) <= 1100 and len(hip._w2t_packed) <= 1100


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,B", [("vp/CIFAR10/indm_nll", 1), ("vp/CIFAR10/indm_nll", 3), ("ve/CIFAR10/indm", 3),
                                    ("vp/CELEBA/indm_fid", 1)])
def test_every_entry_point_runs_at_small_odd_batches(name, B):
    """Smoke sweep of the whole function-level surface at batch sizes the goldens do not use, full-size networks: a
    training step in fp32 and in bf16 compute, the flow round trip, nelbo_fn and the configured sampler (8 steps) --
    everything finite, the round trip within the fixed-point rule."""
    from indm_amd import likelihood, losses, sampling, sde_lib, utils
    dev = torch.device(DEV)
    cfg = configs.get_config(name)
    cfg.device = dev
    cfg.training.batch_size = B
    cfg.sampling.num_scales = 8
    torch.manual_seed(4)
    score = utils.create_model(cfg, dev)
    flow = create_flow_model(cfg).to(dev)
    state, fstate, _ = utils.build_states(cfg, score, flow, dev)
    sde = sde_lib.get_sde(cfg)
    S = cfg.data.image_size
    x = _r(B, 3, S, S, seed=5)
    rng = DeviceRng(dev, seed=6)
    try:
        for dtype in ("f32", "bf16"):
            cfg.compute_dtype = dtype
            step_fn = losses.get_step_fn(cfg, sde, train=True)
            score.train(); flow.train()
            out = step_fn(state, fstate, x, rng=rng)
            assert all(o.shape == (B,) and torch.isfinite(o).all() for o in out), (dtype, out)
        hip.set_compute_dtype("f32")
        score.eval(); flow.eval()
        inv = utils.get_data_inverse_scaler(cfg)
        with torch.no_grad():
            z, _ = flow_forward(cfg, flow, x, log_det=None, reverse=False, rng=rng)
            xr, _ = flow_forward(cfg, flow, z, log_det=None, reverse=True, rng=rng)
        # the reference's stopping rule is (x - x_prev)^2 / (1e-5 + 1e-5 |y|) < 1 per element and block: ~3e-3 per
        # block, 32 blocks in a row (after two lr = 1e-3 steps the blocks sit closer to their Lipschitz bound than at init)
        err = (xr - x).abs()
        assert err.max().item() < 0.25 and err.mean().item() < 2e-2, (err.max().item(), err.mean().item())
        nb = likelihood.get_elbo_fn(cfg, sde, inv)(score, flow, x, rng=rng)
        assert nb[0].shape == (B,) and torch.isfinite(nb[0]).all() and torch.isfinite(nb[1]).all()
        fn = sampling.get_sampling_fn(cfg, sde, (B, 3, S, S), inv, cfg.sampling.truncation_time, device=DEV)
        smp = fn(score, flow, rng=rng)
        assert smp[1].shape == (B, 3, S, S) and torch.isfinite(smp[1]).all()
    finally:
        hip.set_compute_dtype("f32")


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_training_is_bit_reproducible(dtype):
    """Two runs from the same seeds give bit-identical losses AND parameters after three full-size steps: every
    reduction in the library (split-K, column / segment sums, the stage-2 cross-wave sum, grad-norm) has a fixed order;
    nothing uses atomics."""
    from indm_amd import losses, sde_lib, utils

    def run():
        dev = torch.device(DEV)
        cfg = configs.get_config("vp/CIFAR10/indm_nll")
        cfg.device = dev
        cfg.training.batch_size = 4
        cfg.compute_dtype = dtype
        torch.manual_seed(0)
        score = utils.create_model(cfg, dev)
        flow = create_flow_model(cfg).to(dev)
        state, fstate, _ = utils.build_states(cfg, score, flow, dev)
        step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
        score.train(); flow.train()
        rng = DeviceRng(dev, seed=21, poisson_seed=22)
        g = torch.Generator().manual_seed(23)
        # per-module checksums of every module's forward output and of the gradient arriving at it (EXACT: integer sums
        # over the bit patterns, so a one-ulp change in one element shows), as device scalars in execution order: a
        # failure then names the first MODULE whose output or cotangent differed
        trace, order = [], []
        states.append(hip.debug_state())

        def cksum(t):
            v = t.detach().float().contiguous().view(torch.int32).to(torch.int64)
            return torch.stack([v.sum(), (v * v).sum()])

        def fwd_hook(name):
            def hook(_m, _inp, out):
                outs = [o for o in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(o) and o.is_floating_point()]
                for j, o in enumerate(outs):
                    trace.append(cksum(o)); order.append(f"fwd {name}[{j}]")
                    if o.requires_grad:
                        o.register_hook(lambda gr, nm=f"bwd {name}[{j}]": (trace.append(cksum(gr)), order.append(nm)) and None)
            return hook

        for pre, mod in (("score", score), ("flow", flow)):
            for n, m in mod.named_modules():
                if n:
                    m.register_forward_hook(fwd_hook(f"{pre}.{n}"))
        rec = []
        for _ in range(3):
            x = (torch.rand(4, 3, 32, 32, generator=g) * 2 - 1).to(dev)
            del trace[:], order[:]
            out = torch.stack(step_fn(state, fstate, x, rng=rng))
            torch.cuda.synchronize()
            step_trace, step_order = (torch.stack(trace) if trace else torch.zeros(0, 2, dtype=torch.int64)), list(order)
            # the flat gradient buffers still hold this step's gradients: kept so that a failure names the first
            # parameter whose gradient differed, not just "the losses differ"
            rec.append({"module trace": step_trace, "_order": step_order,
                        "loss": out, "score grad": state["optimizer"].flat.g.clone(),
                        "flow grad": fstate["optimizer"].flat.g.clone(),
                        "score params": state["optimizer"].flat.p.clone(),
                        "flow params": fstate["optimizer"].flat.p.clone()})
        names = {}
        for key, st in (("score grad", state), ("flow grad", fstate)):
            o, lst = 0, []
            for n, p in st["optimizer"].flat.named:
                lst.append((n, o, p.numel()))
                o += (p.numel() + 3) // 4 * 4
            names[key] = lst
        return rec, names

    states = []

    def first_difference(a, b, names):
        for s in range(3):
            for key in a[s]:
                if key == "_order":
                    continue
                x, y = a[s][key], b[s][key]
                if key == "module trace":
                    if a[s]["_order"] != b[s]["_order"]:
                        return f"step {s}: the two runs executed different module sequences"
                    bad = (x != y).any(dim=1).nonzero().flatten().tolist()
                    if bad:
                        od = a[s]["_order"]
                        return (f"step {s}: {len(bad)} of {len(od)} module checksums differ; first (execution order): "
                                f"{[(od[i], x[i].tolist(), y[i].tolist()) for i in bad[:4]]}; last identical before it: "
                                f"{od[bad[0] - 1] if bad[0] else None}; kinds of the differing ones: "
                                f"{sorted(set(od[i].split()[0] for i in bad))}")
                    continue
                if not torch.equal(x, y):
                    msg = f"step {s}: {key} differs, max |d| {(x - y).abs().max().item():.3e}"
                    if key == "loss":
                        msg += f", rows {sorted(set((x != y).nonzero()[:, 0].tolist()))} of (total, score, flow, logp)"
                    if key in names:
                        d = x != y
                        hit = [(n, int(d[o:o + k].sum()), k) for n, o, k in names[key] if d[o:o + k].any()]
                        same = [n for n, o, k in names[key] if not d[o:o + k].any()]
                        msg += (f"; {len(hit)} of {len(names[key])} parameters differ (name, differing, numel): {hit[:8]} ... "
                                f"{hit[-8:]}; identical: {same[:40]}")
                    return msg
        return None

    try:
        (a, names), (b, _) = run(), run()
    finally:
        hip.set_compute_dtype("f32")
    diff = first_difference(a, b, names)
    if diff is not None:   # which run is the odd one out?  (a third run tells a first-run effect from a standing race)
        try:
            c, _ = run()
        finally:
            hip.set_compute_dtype("f32")
        diff += (f"  ||  run 2 vs run 3: {first_difference(b, c, names)}  ||  run 1 vs run 3: {first_difference(a, c, names)}"
                 f"  ||  process state at the start of each run: {states}")
    assert diff is None, diff
