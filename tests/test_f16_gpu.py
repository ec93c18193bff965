"""bf16-storage kernels of the residual flow (csrc/flow_f16.hip) against fp64 statements of the same contractions on
bf16-rounded inputs: the F16 fragment layout itself, the skinny 3x3 convolution with its three epilogues, and the four
modes of the 64-row kernel (GEMM, GEMM + Sin activation, the back-to-back chain link, the thin-N panel product)."""
import math

import pytest
import torch

import _emu
from indm_amd import hip

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def r16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def f16_order(x):
    """fp32 [M, N] -> the flat element order of the F16 layout (include/indm_hip.h), on the host"""
    M, N = x.shape
    v = x.view(M // 32, 32, N // 32, 2, 2, 2, 4)          # [mt, l31, nb, s, a, lhi, b]: n = 32 nb + 16 s + 8 a + 4 lhi + b
    return v.permute(0, 2, 3, 5, 1, 4, 6).reshape(-1)     # [mt, nb, s, lhi, l31, a, b]


def close16(got, ref, name, ulps=1.0, atol=1e-6):
    """got: fp32 values of a bf16 result; ref: fp64.  One bf16 ulp = 2^-8 relative (round-to-nearest: half of it, plus
    the accumulation-order noise of an fp32 MFMA chain)."""
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs()
    bound = ulps * 2.0 ** -8 * ref.abs() + atol + 1e-3 * 2.0 ** -8 * ref.abs().max()
    bad = (err > bound)
    assert not bad.any(), f"{name}: {int(bad.sum())} of {bad.numel()} beyond {ulps} bf16 ulp; max err {err.max():.3e} (ref max {ref.abs().max():.3e})"


@pytest.mark.parametrize("M,N", [(64, 512), (4096, 512), (96, 64)])
def test_pack_unpack_layout(M, N):
    x = rnd(M, N)
    y = hip.f16_pack(x.to(DEV))
    assert y.dtype == torch.bfloat16 and y.numel() == M * N
    assert torch.equal(y.cpu().float(), f16_order(r16(x)))
    assert torch.equal(hip.f16_unpack(y, M, N).cpu(), r16(x))


def test_packed_weight_order():
    w = rnd(512, 27, seed=3)
    p = hip.pack_weight_bf16(w.to(DEV)).cpu().float().view(32, 1, 64, 8)
    wp = torch.nn.functional.pad(r16(w), (0, 5))
    for ks, lane, e in ((0, 0, 0), (3, 37, 5), (31, 63, 7), (17, 31, 2)):
        k = 16 * ks + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3)
        assert p[ks, 0, lane, e] == wp[k, lane & 31]


@pytest.mark.parametrize("N,H,W,C", [(8, 32, 32, 3), (4, 16, 16, 12), (2, 64, 64, 3), (32, 16, 16, 1), (3 * 4, 32, 32, 3)])
@pytest.mark.parametrize("sign", [-1, 1])
def test_thin_conv(N, H, W, C, sign):
    Co, M = 512, N * H * W
    x, w, b = rnd(N, H, W, C), rnd(9 * C, Co, seed=1, scale=0.3), rnd(Co, seed=2)
    ref = _emu.conv3x3_thin(r16(x).double(), r16(w).double(), sign) .reshape(M, Co) + b.double()
    out = hip.f16_thin_conv(x.to(DEV), w.to(DEV), sign, hip.F16_PLAIN, bias=b.to(DEV))
    assert out is not None
    close16(hip.f16_unpack(out, M, Co), ref, "thin conv")
    a, d = hip.f16_thin_conv(x.to(DEV), w.to(DEV), sign, hip.F16_ACT, bias=b.to(DEV))
    close16(hip.f16_unpack(a, M, Co), torch.sin(2 * math.pi * ref) / (2 * math.pi), "thin conv sin", ulps=1.0, atol=3e-5)
    close16(hip.f16_unpack(d, M, Co), torch.cos(2 * math.pi * ref), "thin conv cos", ulps=1.0, atol=2e-4)
    g = rnd(M, Co, seed=5)
    u, raw = hip.f16_thin_conv(x.to(DEV), w.to(DEV), sign, hip.F16_GATE, bias=b.to(DEV), gate=hip.f16_pack(g.to(DEV)),
                               want_raw=True)
    close16(hip.f16_unpack(raw, M, Co), ref, "thin conv raw")
    close16(hip.f16_unpack(u, M, Co), ref * r16(g).double(), "thin conv gated", ulps=1.5)
    u2, none = hip.f16_thin_conv(x.to(DEV), w.to(DEV), sign, hip.F16_GATE, bias=None, gate=hip.f16_pack(g.to(DEV)))
    assert none is None
    close16(hip.f16_unpack(u2, M, Co), (ref - b.double()) * r16(g).double(), "thin conv gated, no bias", ulps=1.5)


@pytest.mark.parametrize("N,H,W,C", [(4, 16, 16, 48), (32, 16, 16, 48), (2, 32, 32, 16)])
@pytest.mark.parametrize("sign", [-1, 1])
def test_wide_conv_on_the_im2col_operand(N, H, W, C, sign):
    """12 < C <= 48 (CelebA's 48-channel flow scale, 9 C = 432 taps): indm_f16_tap_spread + the 64-row GEMM with its three
    epilogues = the convolution the thin kernel computes at C <= 12 (same fp64 statement, same tolerances)."""
    Co, M, K = 512, N * H * W, 9 * C
    x, w, b = rnd(N, H, W, C), rnd(K, Co, seed=1, scale=0.1), rnd(Co, seed=2)
    col = hip.f16_tap_spread(x.to(DEV), sign)
    assert col is not None
    # the operand is exactly the fp32 spread, zero-padded to 512 columns and rounded to bf16
    want = hip.f16_pack(torch.nn.functional.pad(hip.tap_spread(x.to(DEV), sign).view(M, K), (0, 512 - K)))
    assert torch.equal(col.view(torch.int16), want.view(torch.int16))
    Wp = hip.pack_weight_bf16(torch.nn.functional.pad(w, (0, 0, 0, 512 - K)).to(DEV))
    ref = _emu.conv3x3_thin(r16(x).double(), r16(w).double(), sign).reshape(M, Co) + b.double()
    out = hip.f16_rows(hip.F16_GEMM, col, M, Wp=Wp, bias=b.to(DEV))
    close16(hip.f16_unpack(out, M, Co), ref, "wide conv", atol=1e-5)
    a, d = hip.f16_rows(hip.F16_GEMM_ACT, col, M, Wp=Wp, bias=b.to(DEV))
    close16(hip.f16_unpack(a, M, Co), torch.sin(2 * math.pi * ref) / (2 * math.pi), "wide conv sin", atol=3e-5)
    close16(hip.f16_unpack(d, M, Co), torch.cos(2 * math.pi * ref), "wide conv cos", atol=2e-4)
    g = rnd(M, Co, seed=5)
    u, raw = hip.f16_rows(hip.F16_GEMM_GATE, col, M, Wp=Wp, D1=hip.f16_pack(g.to(DEV)), bias=b.to(DEV))
    close16(hip.f16_unpack(raw, M, Co), ref, "wide conv raw", atol=1e-5)
    close16(hip.f16_unpack(u, M, Co), ref * r16(g).double(), "wide conv gated", ulps=1.5, atol=1e-5)


@pytest.mark.parametrize("N,H,W,C", [(8, 32, 32, 3), (128, 32, 32, 3), (4, 16, 16, 12), (128, 16, 16, 12), (2, 64, 64, 3), (32, 32, 32, 12)])
def test_chain_conv_is_the_thin_convolution_and_the_chain_link_in_one_launch(N, H, W, C):
    """indm_f16_chain_conv (one series iteration of the log-det estimators as one launch) builds the A tile the thin
    convolution's gate epilogue would have stored, in the same k order and summation order: P is BIT-identical to
    indm_f16_thin_conv(epilogue 2) + indm_f16_rows(mode 2), which have their own fp64 tests above."""
    M, J = N * H * W, 9 * C
    x, w3t = rnd(N, H, W, C), rnd(J, 512, seed=1, scale=0.2)
    d2, d1 = rnd(M, 512, seed=2), rnd(M, 512, seed=3)
    W2T, panel = rnd(512, 512, seed=4, scale=0.05), rnd(512, J, seed=5, scale=0.1)
    d2_16, d1_16 = hip.f16_pack(d2.to(DEV)), hip.f16_pack(d1.to(DEV))
    Wp = hip.pack_weight_bf16(W2T.to(DEV))
    Pp = hip.pack_weight_bf16(torch.nn.functional.pad(panel, (0, (32 if J <= 32 else 128) - J)).to(DEV))
    u2, none = hip.f16_thin_conv(x.to(DEV), w3t.to(DEV), 1, hip.F16_GATE, gate=d2_16)
    assert none is None
    want, _ = hip.f16_rows(hip.F16_CHAIN, u2, M, Wp=Wp, D1=d1_16, Pp=Pp, J=J)
    got = hip.f16_chain_conv(x.to(DEV), hip.pack_thin_conv_bf16(w3t.to(DEV)), d2_16, Wp, d1_16, Pp, sign=1)
    assert got is not None and torch.equal(got, want), (got - want).abs().max().item()
    # the training step's last link: t3 (ungated) and t2 stored as well, all three bit-identical to the two launches
    u2b, t3w = hip.f16_thin_conv(x.to(DEV), w3t.to(DEV), 1, hip.F16_GATE, gate=d2_16, want_raw=True)
    Pw, t2w = hip.f16_rows(hip.F16_CHAIN, u2b, M, Wp=Wp, D1=d1_16, Pp=Pp, J=J, want_t2=True)
    Pg, t3g, t2g = hip.f16_chain_conv(x.to(DEV), hip.pack_thin_conv_bf16(w3t.to(DEV)), d2_16, Wp, d1_16, Pp, sign=1, want_t=True)
    assert torch.equal(Pg, Pw) and torch.equal(t3g.view(torch.int16), t3w.view(torch.int16))
    assert torch.equal(t2g.view(torch.int16), t2w.view(torch.int16))
    # and against the fp64 statement (bf16-rounded operands, the gated intermediates rounded to bf16 where the kernels round them)
    idx = torch.arange(0, M, max(1, M // 128))
    t3 = _emu.conv3x3_thin(r16(x).double(), r16(w3t).double(), 1).reshape(M, 512)[idx]
    u = r16((t3 * r16(d2)[idx].double()).float()).double()
    t2 = u @ r16(W2T).double()
    ref = r16((t2 * r16(d1)[idx].double()).float()).double() @ r16(panel).double()
    assert (got[idx.to(DEV)].cpu().double() - ref).abs().max().item() <= 4e-3 * ref.abs().max().item()


@pytest.mark.parametrize("N,H,W,C", [(8, 32, 32, 3), (128, 32, 32, 3), (4, 16, 16, 12), (128, 16, 16, 12)])
def test_conv_gemm_is_the_thin_convolution_and_the_gemm_in_one_launch(N, H, W, C):
    """indm_f16_conv_gemm (the backward's q1 = conv(q0, W1), q1 .* d1, du2 = (q1 .* d1) W2) against the two launches it replaces:
    all three outputs bit-identical."""
    M = N * H * W
    x, w1 = rnd(N, H, W, C), rnd(9 * C, 512, seed=1, scale=0.2)
    d1, W2 = rnd(M, 512, seed=2), rnd(512, 512, seed=4, scale=0.05)
    d1_16, Wp = hip.f16_pack(d1.to(DEV)), hip.pack_weight_bf16(W2.to(DEV))
    gated, raw = hip.f16_thin_conv(x.to(DEV), w1.to(DEV), -1, hip.F16_GATE, gate=d1_16, want_raw=True)
    want = hip.f16_rows(hip.F16_GEMM, gated, M, Wp=Wp)
    o, r, g = hip.f16_conv_gemm(x.to(DEV), hip.pack_thin_conv_bf16(w1.to(DEV)), d1_16, Wp, sign=-1)
    for got, ref, name in ((o, want, "out"), (r, raw, "raw"), (g, gated, "gated")):
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), name


@pytest.mark.parametrize("N,H,W,C", [(32, 32, 32, 3), (128, 32, 32, 3), (128, 16, 16, 12)])
def test_gnet_in_one_launch(N, H, W, C):
    """indm_f16_gnet (conv + Sin, GEMM + Sin, thin-N) against indm_f16_thin_conv(epilogue 1) + indm_f16_rows mode 5: five outputs,
    bit-identical."""
    M, J = N * H * W, 9 * C
    x, w1, b1 = rnd(N, H, W, C), rnd(J, 512, seed=1, scale=0.2), rnd(512, seed=7)
    W2, b2, cb = rnd(512, 512, seed=4, scale=0.05), rnd(512, seed=8), rnd(N, 512, seed=9)
    panel = rnd(512, J, seed=6, scale=0.1)
    Wp = hip.pack_weight_bf16(W2.to(DEV))
    Pp = hip.pack_weight_bf16(torch.nn.functional.pad(panel, (0, (32 if J <= 32 else 128) - J)).to(DEV))
    a1, d1 = hip.f16_thin_conv(x.to(DEV), w1.to(DEV), -1, hip.F16_ACT, bias=b1.to(DEV))
    a2, d2, P = hip.f16_rows(hip.F16_GEMM_ACT_THINN, a1, M, Wp=Wp, Pp=Pp, J=J, bias=b2.to(DEV), bias2=cb.to(DEV), rows_per_group=H * W)
    got = hip.f16_gnet(x.to(DEV), hip.pack_thin_conv_bf16(w1.to(DEV)), b1.to(DEV), Wp, b2.to(DEV), cb.to(DEV), H * W, Pp)
    assert got is not None
    for g, r, name in zip(got[:4], (a1, d1, a2, d2), ("a1", "d1", "a2", "d2")):
        assert torch.equal(g.view(torch.int16), r.view(torch.int16)), name
    assert torch.equal(got[4], P)


def test_chain_conv_declines_uncovered_geometry():
    z16 = torch.zeros(2048 * 512, dtype=torch.bfloat16, device=DEV)
    Wp = hip.pack_weight_bf16(rnd(512, 512).to(DEV))
    assert hip.f16_chain_conv(rnd(2, 32, 32, 5).to(DEV), hip.pack_thin_conv_bf16(rnd(48, 512).to(DEV)), z16, Wp, z16,
                              hip.pack_weight_bf16(rnd(512, 128).to(DEV))) is None


@pytest.mark.parametrize("M,J", [(64, 27), (32768, 27), (32768, 108), (131072, 27)])
def test_rows_activation_with_thin_n_in_one_launch(M, J):
    """mode 5 = mode 1 (GEMM + Sin activation) + mode 3 (thin-N panel product of the activation) in one launch: all three outputs
    bit-identical to the two launches (which have their own fp64 tests above)."""
    A, Wm, b, b2 = rnd(M, 512), rnd(512, 512, seed=1, scale=0.05), rnd(512, seed=2), rnd(M // 64, 512, seed=3)
    panel = rnd(512, J, seed=6, scale=0.1)
    A16, Wp = hip.f16_pack(A.to(DEV)), hip.pack_weight_bf16(Wm.to(DEV))
    Pp = hip.pack_weight_bf16(torch.nn.functional.pad(panel, (0, (32 if J <= 32 else 128) - J)).to(DEV))
    a, d = hip.f16_rows(hip.F16_GEMM_ACT, A16, M, Wp=Wp, bias=b.to(DEV), bias2=b2.to(DEV), rows_per_group=64)
    T = hip.f16_rows(hip.F16_THINN, a, M, Pp=Pp, J=J)
    a5, d5, T5 = hip.f16_rows(hip.F16_GEMM_ACT_THINN, A16, M, Wp=Wp, Pp=Pp, J=J, bias=b.to(DEV), bias2=b2.to(DEV), rows_per_group=64)
    assert torch.equal(a5.view(torch.int16), a.view(torch.int16)) and torch.equal(d5.view(torch.int16), d.view(torch.int16))
    # (the thin-N launch sums its 32-row tiles' eight wave partials in the same fixed order; its tile is 32 rows, this one's 32 or 64)
    assert torch.equal(T5, T), (T5 - T).abs().max().item()


def test_tap_spread_declines_uncovered_geometry():
    assert hip.f16_tap_spread(rnd(2, 16, 16, 6).to(DEV), -1) is None       # C % 4
    assert hip.f16_tap_spread(rnd(2, 16, 16, 64).to(DEV), -1) is None      # 9 C > 512


def test_thin_conv_declines_uncovered_geometry():
    assert hip.f16_thin_conv(rnd(2, 32, 32, 5).to(DEV), rnd(45, 512).to(DEV), -1) is None
    assert hip.f16_thin_conv(rnd(1, 8, 8, 3).to(DEV), rnd(27, 512).to(DEV), -1) is None      # 64 pixels: M % 128


@pytest.mark.parametrize("M", [64, 4096, 131072])
def test_rows_gemm_and_activation(M):
    K = N = 512
    A, Wm, b, b2 = rnd(M, K), rnd(K, N, seed=1, scale=0.05), rnd(N, seed=2), rnd(M // 64, N, seed=3)
    A16, Wp = hip.f16_pack(A.to(DEV)), hip.pack_weight_bf16(Wm.to(DEV))
    idx = torch.arange(0, M, max(1, M // 512))
    ref = r16(A)[idx].double() @ r16(Wm).double()
    out = hip.f16_rows(hip.F16_GEMM, A16, M, Wp=Wp)
    close16(hip.f16_unpack(out, M, N)[idx.to(DEV)], ref, "rows gemm", atol=1e-5)
    refb = ref + b.double() + b2.double()[idx // 64]
    out = hip.f16_rows(hip.F16_GEMM, A16, M, Wp=Wp, bias=b.to(DEV), bias2=b2.to(DEV), rows_per_group=64)
    close16(hip.f16_unpack(out, M, N)[idx.to(DEV)], refb, "rows gemm + biases", atol=1e-5)
    a, d = hip.f16_rows(hip.F16_GEMM_ACT, A16, M, Wp=Wp, bias=b.to(DEV), bias2=b2.to(DEV), rows_per_group=64)
    close16(hip.f16_unpack(a, M, N)[idx.to(DEV)], torch.sin(2 * math.pi * refb) / (2 * math.pi), "rows act sin", atol=3e-5)
    close16(hip.f16_unpack(d, M, N)[idx.to(DEV)], torch.cos(2 * math.pi * refb), "rows act cos", atol=2e-4)


@pytest.mark.parametrize("M,J", [(64, 27), (4096, 27), (8192, 108), (131072, 27), (32768, 108), (1024, 432), (8192, 432), (65536, 144)])
def test_rows_chain_and_thin_n(M, J):
    K = N = 512
    A, Wm, D1 = rnd(M, K), rnd(K, N, seed=1, scale=0.05), rnd(M, N, seed=4)
    panel = rnd(N, J, seed=6, scale=0.1)
    A16, D16 = hip.f16_pack(A.to(DEV)), hip.f16_pack(D1.to(DEV))
    Wp, Pp = hip.pack_weight_bf16(Wm.to(DEV)), hip.pack_weight_bf16(torch.nn.functional.pad(panel, (0, (32 if J <= 32 else 128 if J <= 128 else 448) - J)).to(DEV))
    idx = torch.cat([torch.arange(0, M, max(1, M // 256)), torch.arange(max(0, M - 70), M)]).unique()
    t2 = r16(A)[idx].double() @ r16(Wm).double()
    u1 = r16((t2 * r16(D1)[idx].double()).float()).double()       # the gated intermediate is rounded to bf16 for the second MFMA
    refP = u1 @ r16(panel).double()
    P, t2o = hip.f16_rows(hip.F16_CHAIN, A16, M, Wp=Wp, D1=D16, Pp=Pp, J=J, want_t2=True)
    scale = refP.abs().max().item()
    assert (P[idx.to(DEV)].cpu().double() - refP).abs().max().item() <= 4e-3 * scale, "chain P"
    close16(hip.f16_unpack(t2o, M, N)[idx.to(DEV)], t2, "chain t2", atol=1e-5)
    P2, none = hip.f16_rows(hip.F16_CHAIN, A16, M, Wp=Wp, D1=D16, Pp=Pp, J=J)
    assert none is None and torch.equal(P2, P)
    refT = r16(A)[idx].double() @ r16(panel).double()
    T = hip.f16_rows(hip.F16_THINN, A16, M, Pp=Pp, J=J)
    assert (T[idx.to(DEV)].cpu().double() - refT).abs().max().item() <= 1e-4 * refT.abs().max().item() + 1e-5, "thin-N"


@pytest.mark.parametrize("R,rps", [(4096, 1024), (1024, 256), (131072, 1024), (8192, 4096)])
def test_bwd_act(R, rps):
    N = 512
    ts = [rnd(R, N, seed=i) for i in range(5)]
    da, du, d, a, t = (hip.f16_pack(x.to(DEV)) for x in ts)
    rda, rdu, rd, ra, rt = (r16(x).double() for x in ts)
    ref = rda * rd - 4 * math.pi ** 2 * ra * (rdu * rt)
    dp, seg = hip.f16_bwd_act(da, du, d, a, t, R, rps)
    idx = torch.arange(0, R, max(1, R // 1024))
    close16(hip.f16_unpack(dp, R, N)[idx.to(DEV)], ref[idx], "bwd_act dp", ulps=1.0, atol=1e-4)
    # the column sums are taken of the fp32 values BEFORE the bf16 rounding of dp
    sref = ref.view(R // rps, rps, N).sum(1)
    err = (seg.cpu().double() - sref).abs().max().item()
    assert err <= 2e-4 * sref.abs().max().item() + 1e-3, err
    dp2, none = hip.f16_bwd_act(da, du, d, a, t, R, None) if R % 128 == 0 else (dp, None)
    assert none is None and torch.equal(dp2, dp)
    assert hip.f16_bwd_act(da, du, d, a, t, R, 64) is None


@pytest.mark.parametrize("M,T", [(4096, 27), (1024, 108), (131072, 27), (32768, 108), (96, 5)])
def test_thin_wgrad(M, T):
    N = 512
    thin, wide, gate = rnd(M, T, seed=1), rnd(M, N, seed=2), rnd(M, N, seed=3)
    w16, g16 = hip.f16_pack(wide.to(DEV)), hip.f16_pack(gate.to(DEV))
    ref = r16(thin).double().t() @ r16(wide).double()
    out = hip.f16_thin_wgrad(thin.to(DEV), w16, M, alpha=0.5)
    assert (out.cpu().double() - 0.5 * ref).abs().max().item() <= 2e-5 * math.sqrt(M) + 1e-4 * ref.abs().max().item()
    out = hip.f16_thin_wgrad(thin.to(DEV), w16, M, transpose=True)
    assert out.shape == (N, T)
    assert (out.cpu().double() - ref.t()).abs().max().item() <= 4e-5 * math.sqrt(M) + 1e-4 * ref.abs().max().item()
    refg = r16(thin).double().t() @ r16((r16(wide) * r16(gate))).double()      # the gated product is rounded to bf16
    out = hip.f16_thin_wgrad(thin.to(DEV), w16, M, gate=g16)
    assert (out.cpu().double() - refg).abs().max().item() <= 4e-5 * math.sqrt(M) + 1e-4 * refg.abs().max().item()


@pytest.mark.parametrize("M", [64, 4096, 131072])
def test_wgrad_gemm(M):
    N = 512
    X, Y, G = rnd(M, N, seed=1), rnd(M, N, seed=2), rnd(M, N, seed=3)
    x16, y16, g16 = (hip.f16_pack(v.to(DEV)) for v in (X, Y, G))
    ref = r16(X).double().t() @ r16(Y).double()
    out = hip.f16_wgrad_gemm(x16, y16, M)
    tol = 4e-5 * math.sqrt(M) + 1e-4 * ref.abs().max().item()
    assert (out.cpu().double() - ref).abs().max().item() <= tol
    out = hip.f16_wgrad_gemm(x16, y16, M, alpha=2.0, transpose=True)
    assert (out.cpu().double() - 2.0 * ref.t()).abs().max().item() <= 2 * tol
    refg = r16(r16(X) * r16(G)).double().t() @ r16(Y).double()
    out = hip.f16_wgrad_gemm(x16, y16, M, gate=g16)
    assert (out.cpu().double() - refg).abs().max().item() <= tol
