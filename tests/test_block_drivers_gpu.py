"""Block-level C-ABI drivers (indm_iresblock_fwd_logdet / _bwd / _inverse, csrc/iresblock.hip) against the op-level
composition they replace (indm_amd/flow.py with hip.block_drivers(False)): same kernels, same order -> the results must be
IDENTICAL (training forward + every gradient, eval power-series log-det, fixed-point inverse), on the full-size
16-16 / idim 512 flow at a batch where both scales are covered (M = 4096 / 1024 rows).  The op-level composition itself
is pinned to the reference's goldens and the CPU oracle by the parity suites."""
import pytest
import torch

from indm_amd import configs, hip
from indm_amd.flow import create_flow_model, flow_forward
from indm_amd.rng import DeviceRng

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _flow(seed=3, name="vp/CIFAR10/indm_nll"):
    cfg = configs.get_config(name)
    cfg.device = torch.device(DEV)
    torch.manual_seed(seed)
    return cfg, create_flow_model(cfg).to(DEV)


@pytest.fixture
def drivers():
    was = hip.block_drivers()
    yield
    hip.block_drivers(was)


def _train_pass(cfg, fl, x, on):
    hip.block_drivers(on)
    fl.train()
    for p in fl.parameters():
        p.grad = None
    z, ld = flow_forward(cfg, fl, x, reverse=False, rng=DeviceRng(DEV, seed=11, poisson_seed=5))
    gz = torch.sin(torch.arange(z.numel(), device=DEV, dtype=torch.float32)).view_as(z)
    (z * gz).sum().add(ld.sum() * 0.37).backward()
    grads = {n: p.grad.clone() for n, p in fl.named_parameters() if p.grad is not None}
    return z.detach().clone(), ld.detach().clone(), grads


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_training_block_driver_is_bit_identical_to_op_level(drivers, dtype):
    cfg, fl = _flow()
    x = torch.rand(4, 3, 32, 32, device=DEV) * 2 - 1
    hip.set_compute_dtype(dtype)
    was = hip.f16_storage(False)   # like with like: bf16 MFMA inputs on fp32 tensors (the F16 storage form has its own test)
    try:
        z1, ld1, g1 = _train_pass(cfg, fl, x, True)
        z0, ld0, g0 = _train_pass(cfg, fl, x, False)
    finally:
        hip.set_compute_dtype("f32")
        hip.f16_storage(True)
    assert torch.equal(z1, z0) and torch.equal(ld1, ld0)
    assert set(g1) == set(g0) and len(g1) > 200
    for n in g0:
        assert torch.equal(g1[n], g0[n]), (n, (g1[n] - g0[n]).abs().max().item())


def test_driver_is_actually_used_and_timed(drivers):
    """the driver path runs (chain launches are visible to the C-side launch timer) and declines uncovered geometry"""
    cfg, fl = _flow()
    x = torch.rand(4, 3, 32, 32, device=DEV) * 2 - 1
    t = hip.ChainTimer(1024)
    assert t.reset("ires_chain64x512x32,gated")
    hip.set_chain_timer(t)
    try:
        _train_pass(cfg, fl, x, True)
    finally:
        hip.set_chain_timer(None)
    r = t.read()
    # 16 scale-0 blocks x K = n + 2 >= 2 no-grad series iterations each
    assert r["launches"] >= 32 and r["ms"] > 0 and r["flops"] == r["launches"] * 2.0 * 4096 * 512 * (512 + 27)
    # batch 1: scale 1 has M = 256 rows -> not covered -> None (the caller composes the op-level calls)
    e = torch.zeros(1, 16, 16, 12, device=DEV)
    w = torch.zeros(512, 512, device=DEV)
    assert hip.iresblock_fwd(e, torch.zeros(1, 512, device=DEV), torch.zeros(3, 3, 12, 512, device=DEV), w,
                             torch.zeros(3, 3, 512, 12, device=DEV), torch.zeros(512, 128, device=DEV),
                             torch.zeros(512, device=DEV), torch.zeros(512, device=DEV), torch.zeros(12, device=DEV), e,
                             [1.0, 1.0], True, True) is None


def test_eval_logdet_and_inverse_drivers_match_op_level(drivers):
    cfg, fl = _flow(5)
    fl.eval()
    x = torch.rand(4, 3, 32, 32, device=DEV) * 2 - 1
    outs = {}
    for on in (True, False):
        hip.block_drivers(on)
        with torch.no_grad():
            z, ld = flow_forward(cfg, fl, x, reverse=False, rng=DeviceRng(DEV, seed=21, poisson_seed=9))
            it = []
            xr, _ = flow_forward(cfg, fl, z, reverse=True, rng=DeviceRng(DEV, seed=22), iters=it)
        outs[on] = (z, ld, xr, it)
    (z1, ld1, x1, it1), (z0, ld0, x0, it0) = outs[True], outs[False]
    # the drivers multiply by W2 through its transposed copy in every mode (the op-level eval path uses W2 itself):
    # same products, possibly another summation order -> a few ulp
    assert (z1 - z0).abs().max().item() <= 2e-6 * z0.abs().max().item()
    assert (ld1 - ld0).abs().max().item() <= 1e-5 * ld0.abs().max().item() + 1e-5
    assert it1 == it0
    assert (x1 - x0).abs().max().item() <= 1e-5


@pytest.mark.parametrize("B", [4, 32])
def test_fused_f16_launches_are_bit_identical_at_the_flow_level(drivers, B):
    """Round 6: with the thin-order weight packs the F16 drivers run a series iteration (indm_f16_chain_conv), the g-net
    (indm_f16_gnet, batch >= 32: 64-row tiles) and the backward's conv + gate + GEMM (indm_f16_conv_gemm) as one launch each.  Each
    kernel is bit-identical to the launches it replaces (tests/test_f16_gpu.py); here the whole flow: output, log-dets and every
    parameter gradient must not change by a bit when the packs are withheld (hip.CHAIN_CONV off)."""
    cfg, fl = _flow()
    x = torch.rand(B, 3, 32, 32, device=DEV) * 2 - 1
    was = hip.CHAIN_CONV["on"]
    try:
        hip.set_compute_dtype("bf16")
        hip.f16_storage(True)
        hip.CHAIN_CONV["on"] = True
        z1, ld1, g1 = _train_pass(cfg, fl, x, True)
        hip.CHAIN_CONV["on"] = False
        z0, ld0, g0 = _train_pass(cfg, fl, x, True)
    finally:
        hip.CHAIN_CONV["on"] = was
        hip.set_compute_dtype("f32")
    assert torch.equal(z1, z0) and torch.equal(ld1, ld0) and set(g1) == set(g0) and len(g0) > 200
    for n in g0:
        assert torch.equal(g1[n], g0[n]), (n, (g1[n] - g0[n]).abs().max().item())


@pytest.mark.parametrize("name", ["vp/CIFAR10/indm_nll", "vp/CELEBA/indm_fid"])
def test_two_stream_f16_backward_is_bit_identical(drivers, name):
    """the bf16-storage backward driver with its weight / bias gradients on the side stream (hip.BWD_STREAMS_F16): same bits, three
    times in a row (CelebA: the wide 48-channel form has a spread operand per stream)."""
    cfg, fl = _flow(name=name)
    S = cfg.data.image_size
    x = torch.rand(8, 3, S, S, device=DEV) * 2 - 1
    was = hip.BWD_STREAMS_F16["on"]
    try:
        hip.set_compute_dtype("bf16")
        hip.f16_storage(True)
        hip.BWD_STREAMS_F16["on"] = False
        z0, ld0, g0 = _train_pass(cfg, fl, x, True)
        hip.BWD_STREAMS_F16["on"] = True
        for _ in range(3):
            z1, ld1, g1 = _train_pass(cfg, fl, x, True)
            assert torch.equal(z1, z0) and torch.equal(ld1, ld0) and set(g1) == set(g0)
            for n in g0:
                assert torch.equal(g1[n], g0[n]), (n, (g1[n] - g0[n]).abs().max().item())
    finally:
        hip.BWD_STREAMS_F16["on"] = was
        hip.set_compute_dtype("f32")


def test_two_stream_backward_is_bit_identical(drivers):
    """indm_iresblock_bwd_args.side_stream (weight / bias gradients on a second stream, ordered by events): same kernels and
    arguments -> the same bits as the one-stream order, repeatedly (a missing event would show up as a run-to-run difference)."""
    cfg, fl = _flow()
    x = torch.rand(16, 3, 32, 32, device=DEV) * 2 - 1
    was = hip.bwd_streams()
    try:
        hip.bwd_streams(False)
        z0, ld0, g0 = _train_pass(cfg, fl, x, True)
        hip.bwd_streams(True)
        for _ in range(3):
            z1, ld1, g1 = _train_pass(cfg, fl, x, True)
            assert torch.equal(z1, z0) and torch.equal(ld1, ld0) and set(g1) == set(g0)
            for n in g0:
                assert torch.equal(g1[n], g0[n]), (n, (g1[n] - g0[n]).abs().max().item())
    finally:
        hip.bwd_streams(was)


def _rel(a, b):
    return (a - b).norm().item() / (b.norm().item() + 1e-30)


@pytest.mark.parametrize("B,name", [(4, "vp/CIFAR10/indm_nll"), (16, "vp/CIFAR10/indm_nll"), (4, "vp/CELEBA/indm_fid")])
def test_bf16_storage_block_tracks_bf16_compute_on_fp32_tensors(drivers, B, name):
    """bf16 STORAGE of the flow's 512-wide tensors (F16 fragment order, csrc/flow_f16.hip: activations rounded to bf16
    once more than the compute-only mode, hardware sine / cosine, gates applied at other points of the chain) against the
    bf16-compute / fp32-tensor drivers and the fp32 drivers on the same weights and noise: outputs, log-dets and every
    parameter gradient.  CelebA: the 12-channel scale at 32 x 32 and the 48-channel scale at 16 x 16 (round 6: the wide form of
    the F16 drivers -- im2col operand + 64-row GEMMs; before, that scale ran the op-level composition on fp32 tensors)."""
    cfg, fl = _flow(name=name)
    S = cfg.data.image_size
    x = torch.rand(B, 3, S, S, device=DEV) * 2 - 1
    try:
        hip.set_compute_dtype("bf16")
        hip.f16_storage(True)
        assert hip.storage_dtype() == "bf16"
        hip.f16_stats(reset=True)
        z2, ld2, g2 = _train_pass(cfg, fl, x, True)
        stats = hip.f16_stats(reset=True)
        want = {(3, 32, 32), (12, 16, 16)} if S == 32 else {(12, 32, 32), (48, 16, 16)}
        assert set(stats) == want and all(v[0] > 0 and v[1] == 0 for v in stats.values()), stats
        hip.f16_storage(False)
        z1, ld1, g1 = _train_pass(cfg, fl, x, True)
        hip.set_compute_dtype("f32")
        z0, ld0, g0 = _train_pass(cfg, fl, x, True)
    finally:
        hip.set_compute_dtype("f32")
        hip.f16_storage(True)
    assert torch.isfinite(z2).all() and torch.isfinite(ld2).all()
    # the flow's output is x + sum of 32 small residuals: absolute agreement far below a bf16 ulp of the values
    assert (z2 - z0).abs().max().item() <= 2e-3, (z2 - z0).abs().max().item()
    e1, e2 = (ld1 - ld0).abs().max().item(), (ld2 - ld0).abs().max().item()
    scale = ld0.abs().max().item()
    assert e2 <= max(3.0 * e1, 2e-2 * scale) + 1e-3, (e1, e2, scale)
    assert set(g2) == set(g0)
    worst = []
    for n in g0:
        r1, r2 = _rel(g1[n], g0[n]), _rel(g2[n], g0[n])
        worst.append((r2, r1, n))
        # bias gradients are sums of signed per-pixel terms over every pixel of the batch (heavy cancellation): their
        # relative error is the terms' rounding noise over a small sum -- twice the slack of the weight gradients
        assert r2 <= max((8.0 if n.endswith(".bias") else 4.0) * r1, 3e-2), (n, r1, r2)
    print("largest relative gradient errors (storage, compute-only, name):", sorted(worst, reverse=True)[:5])


def test_bf16_storage_eval_logdet_and_inverse_tolerances(drivers):
    """with compute_dtype = 'bf16' the EVAL forward (the NELBO's power-series log-det, n + 20 terms) also runs on
    bf16-stored activations with hardware sine / cosine, while the fixed-point inverse runs the bf16-compute kernels on fp32
    tensors -- the two directions of one flow use different numerics.  Stated tolerances, against the fp32 drivers on the same
    weights and noise: z within 2e-3 absolute, the log-det within 2 % of its largest entry (+1e-3), and the inverse of the SAME
    latent (same prior sample h: the reverse pass draws its conditioning from the prior, so forward -> inverse is not a
    round trip in either mode) within 5e-3 absolute, same iteration counts +- 1 per block."""
    cfg, fl = _flow(7)
    fl.eval()
    hip.block_drivers(True)
    x = torch.rand(4, 3, 32, 32, device=DEV) * 2 - 1
    outs = {}
    try:
        for mode in ("f32", "bf16"):
            hip.set_compute_dtype(mode)
            hip.f16_storage(True)
            assert hip.storage_dtype() == ("bf16" if mode == "bf16" else "f32")
            hip.f16_stats(reset=True)
            with torch.no_grad():
                z, ld = flow_forward(cfg, fl, x, reverse=False, rng=DeviceRng(DEV, seed=31, poisson_seed=9))
                stats = hip.f16_stats(reset=True)
                if mode == "f32":
                    z_ref = z
                it = []
                xr, _ = flow_forward(cfg, fl, z_ref, reverse=True, rng=DeviceRng(DEV, seed=32), iters=it)
            outs[mode] = (z, ld, xr, stats, it)
    finally:
        hip.set_compute_dtype("f32")
    z0, ld0, x0, s0, it0 = outs["f32"]
    z1, ld1, x1, s1, it1 = outs["bf16"]
    assert s0 == {(3, 32, 32): [0, 16], (12, 16, 16): [0, 16]} and s1 == {(3, 32, 32): [16, 0], (12, 16, 16): [16, 0]}, (s0, s1)
    assert torch.isfinite(z1).all() and torch.isfinite(ld1).all() and torch.isfinite(x1).all()
    assert (z1 - z0).abs().max().item() <= 2e-3
    assert (ld1 - ld0).abs().max().item() <= 2e-2 * ld0.abs().max().item() + 1e-3, (ld1, ld0)
    assert (x1 - x0).abs().max().item() <= 5e-3, (x1 - x0).abs().max().item()
    assert len(it0) == len(it1) and max(abs(a - b) for a, b in zip(it0, it1)) <= 1, (it0, it1)
