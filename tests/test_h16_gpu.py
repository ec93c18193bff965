"""bf16 STORAGE of the score network (csrc/unet_h16.hip) -- per-kernel checks through the C ABI against fp64 PyTorch
statements of the same op on the SAME bf16-rounded inputs (products of bf16 values are exact in fp32: what is left is the
fp32 accumulation order and the final rounding of the output to bf16, 2^-9 relative)."""
import math

import pytest
import torch
import torch.nn.functional as F

from indm_amd import hip

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def bf(x):
    return x.to(torch.bfloat16)


def close_bf16(got, ref, K, name=""):
    """got: bf16 / fp32 tensor of the kernel; ref: fp64.  |err| <= bf16 half-ulp of the value + fp32 accumulation noise"""
    g, r = got.detach().cpu().double(), ref.detach().cpu().double()
    assert g.shape == r.shape, (name, g.shape, r.shape)
    scale = r.abs().max().item() + 1e-30
    tol = (2.0 ** -8) * r.abs() + 1e-6 * math.sqrt(K) * scale + 1e-30
    if got.dtype == torch.float32:
        tol = 1e-6 * math.sqrt(K) * scale + 1e-5 * r.abs()
    bad = ((g - r).abs() > tol)
    assert not bad.any(), f"{name}: {int(bad.sum())} of {bad.numel()} off; max err {(g - r).abs().max().item():.3e} (scale {scale:.3e})"


def test_convert_roundtrip():
    x = rnd(3, 16, 16, 96, scale=3.0).to(DEV)
    y = hip.to_bf16(x)
    assert y.dtype == torch.bfloat16 and torch.equal(y, x.to(torch.bfloat16))
    z = hip.to_f32(y, 0.5)
    assert torch.equal(z, y.float() * 0.5)
    x2 = rnd(1030, scale=2.0).to(DEV)       # tail path (not a multiple of 8)
    assert torch.equal(hip.to_bf16(x2), x2.to(torch.bfloat16))
    assert torch.equal(hip.to_f32(hip.to_bf16(x2)), x2.to(torch.bfloat16).float())


def test_pack_weights():
    ws = [rnd(3, 3, 64, 128, seed=1), rnd(3, 3, 96, 40, seed=2), rnd(256, 128, seed=3), rnd(1, 1, 33, 17, seed=4)]
    packs = hip.h16_pack_weights([w.to(DEV) for w in ws])
    for w, (fwd, dg) in zip(ws, packs):
        if w.dim() == 2:
            w = w[None, None]
        KH, KW, Ci, Co = w.shape
        T = KH * KW
        w3 = w.reshape(T, Ci, Co)
        want_fwd = w3.permute(2, 0, 1).reshape(Co, T * Ci).to(torch.bfloat16)
        want_dg = torch.flip(w3, [0]).permute(1, 0, 2).reshape(Ci, T * Co).to(torch.bfloat16)
        assert torch.equal(fwd.cpu(), want_fwd)
        assert torch.equal(dg.cpu(), want_dg)


CONV_CASES = [
    # N, H, W, Cin, Cout, K, stride, pad
    (4, 16, 16, 64, 128, 3, 1, 1),
    (2, 32, 32, 128, 256, 3, 1, 1),       # 2048 rows: 128-row tiles
    (40, 32, 32, 128, 128, 3, 1, 1),      # 40960 rows: 160 tiles of 256 rows -> still the 128-row kernel; see the big case below
    (2, 8, 8, 512, 256, 3, 1, 1),         # split-K
    (3, 4, 4, 256, 256, 3, 1, 1),         # 48 rows: ragged M
    (2, 16, 16, 96, 40, 3, 1, 1),         # ragged N (40 columns)
    (4, 16, 16, 256, 128, 1, 1, 0),       # 1x1 / NIN
    (2, 33, 33, 64, 64, 3, 2, 0),         # stride 2 on a padded map (conv_downsample_2d)
    (1, 64, 64, 32, 32, 3, 1, 1),
]


def _conv_ref(x, w, stride, pad, alpha, bias, bias2, res, beta):
    xd = bf(x).double().permute(0, 3, 1, 2)
    wd = bf(w).double().permute(3, 2, 0, 1)
    y = F.conv2d(xd, wd, stride=stride, padding=pad).permute(0, 2, 3, 1)
    if bias is not None:
        y = y + bias.double()
    if bias2 is not None:
        y = y + bias2.double()[:, None, None, :]
    y = alpha * y
    if res is not None:
        y = y + beta * bf(res).double()
    return y


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("epi", ["plain", "full", "f32"])
def test_h16_conv2d(case, epi):
    N, H, W, Ci, Co, K, stride, pad = case
    x, w = rnd(N, H, W, Ci, seed=1), rnd(K, K, Ci, Co, seed=2, scale=0.2)
    OH = (H + 2 * pad - K) // stride + 1
    (fwd, _), = hip.h16_pack_weights([w.to(DEV)])
    bias = bias2 = res = None
    alpha, beta = 1.0, 1.0
    if epi == "full":
        bias, bias2, res = rnd(Co, seed=3), rnd(N, Co, seed=4), rnd(N, OH, OH, Co, seed=5)
        alpha, beta = 0.70710678, -0.5
    y = hip.h16_conv2d(bf(x).to(DEV), fwd, K, K, stride, pad, alpha, None if bias is None else bias.to(DEV),
                       None if bias2 is None else bias2.to(DEV), None if res is None else bf(res).to(DEV), beta,
                       out_f32=(epi == "f32"))
    assert y is not None, "geometry should be covered"
    assert y.dtype == (torch.float32 if epi == "f32" else torch.bfloat16)
    close_bf16(y, _conv_ref(x, w, stride, pad, alpha, bias, bias2, res, beta), K * K * Ci, name=f"conv {case} {epi}")


def test_h16_conv2d_big_tile():
    """enough rows for the 256 x 128 kernel (>= 512 tiles): 16 x 32 x 32 x 8 images ..."""
    N, H, W, Ci, Co = 132, 32, 32, 32, 128      # 135168 rows = 528 tiles of 256
    x, w = rnd(N, H, W, Ci, seed=1), rnd(3, 3, Ci, Co, seed=2, scale=0.2)
    bias2, res = rnd(N, Co, seed=4), rnd(N, H, W, Co, seed=5)
    (fwd, _), = hip.h16_pack_weights([w.to(DEV)])
    y = hip.h16_conv2d(bf(x).to(DEV), fwd, 3, 3, 1, 1, 0.5, None, bias2.to(DEV), bf(res).to(DEV), 2.0)
    close_bf16(y, _conv_ref(x, w, 1, 1, 0.5, None, bias2, res, 2.0), 9 * Ci, name="conv big tile")


@pytest.mark.parametrize("case", [(4, 16, 16, 64, 128, 3), (2, 8, 8, 256, 512, 3), (4, 16, 16, 256, 128, 1), (3, 4, 4, 256, 256, 3)])
def test_h16_dgrad_is_the_conv_on_the_flipped_pack(case):
    N, H, W, Ci, Co, K = case
    pad = (K - 1) // 2
    dy, w = rnd(N, H, W, Co, seed=1), rnd(K, K, Ci, Co, seed=2, scale=0.2)
    (_, dg), = hip.h16_pack_weights([w.to(DEV)])
    dx = hip.h16_conv2d(bf(dy).to(DEV), dg, K, K, 1, K - 1 - pad)
    assert dx is not None
    xd = torch.zeros(N, Ci, H, W, dtype=torch.float64, requires_grad=True)
    yd = F.conv2d(xd, bf(w).double().permute(3, 2, 0, 1), padding=pad)
    (ref,) = torch.autograd.grad(yd, xd, bf(dy).double().permute(0, 3, 1, 2))
    close_bf16(dx, ref.permute(0, 2, 3, 1), K * K * Co, name=f"dgrad {case}")


WGRAD_CASES = [
    (4, 16, 16, 64, 128, 3, 1, 1),
    (2, 32, 32, 128, 256, 3, 1, 1),
    (8, 8, 8, 384, 128, 3, 1, 1),       # Cin = 384: a 256-row tile straddles taps
    (2, 16, 16, 96, 40, 3, 1, 1),       # ragged M (864 rows) and N
    (4, 16, 16, 256, 128, 1, 1, 0),     # 1x1 / NIN
    (2, 33, 33, 64, 64, 3, 2, 0),       # stride 2
    (32, 4, 4, 256, 256, 3, 1, 1),
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_h16_wgrad(case):
    N, H, W, Ci, Co, K, stride, pad = case
    OH = (H + 2 * pad - K) // stride + 1
    x, dy = rnd(N, H, W, Ci, seed=1), rnd(N, OH, OH, Co, seed=2)
    dw = hip.h16_conv2d_wgrad(bf(x).to(DEV), bf(dy).to(DEV), K, K, stride, pad, 0.5)
    assert dw is not None, "geometry should be covered"
    wd = torch.zeros(Co, Ci, K, K, dtype=torch.float64, requires_grad=True)
    yd = F.conv2d(bf(x).double().permute(0, 3, 1, 2), wd, stride=stride, padding=pad)
    (ref,) = torch.autograd.grad(yd, wd, bf(dy).double().permute(0, 3, 1, 2))
    close_bf16(dw, 0.5 * ref.permute(2, 3, 1, 0), N * OH * OH, name=f"wgrad {case}")


def test_h16_declines_what_it_does_not_cover():
    x = bf(rnd(2, 8, 8, 16)).to(DEV)             # Cin % 32 != 0
    (fwd, _), = hip.h16_pack_weights([rnd(3, 3, 16, 32).to(DEV)])
    assert hip.h16_conv2d(x, fwd, 3, 3) is None
    x = bf(rnd(2, 6, 6, 32)).to(DEV)             # 6 x 6 maps: not a power of two
    (fwd, _), = hip.h16_pack_weights([rnd(3, 3, 32, 32).to(DEV)])
    assert hip.h16_conv2d(x, fwd, 3, 3) is None
    assert hip.h16_conv2d_wgrad(x, bf(rnd(2, 6, 6, 32)).to(DEV), 3, 3) is None
