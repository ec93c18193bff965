"""Production-dispatch parity: the kernel variants that own the batch-128 benchmark are checked
ELEMENTWISE against an fp64 evaluation on the CPU -- not against another kernel of this repo.  Each case runs the exact
shape the step launches at batch 128, asserts through `indm_igemm_last_variant()` that the benchmarked variant was
the one dispatched, and compares >= 4096 sampled output rows (all columns) with torch fp64 on the host.
Tolerance: fp32 products summed in fp32 over K terms vs fp64 -> |err| <= 2e-6 * sqrt(K) * |a||b| scale (stated per case)."""
import pytest
import torch
import torch.nn.functional as F

from indm_amd import hip

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def _variant():
    return hip.lib().indm_igemm_last_variant().decode()


def _rows(M, n=4096, seed=1):
    g = torch.Generator().manual_seed(seed)
    r = torch.randint(0, M, (n - 512,), generator=g)
    # both ends of the range and tile edges of the 256-row blocks are always included
    edge = torch.cat([torch.arange(0, 128), torch.arange(M - 128, M), torch.arange(255, 255 + 256),
                      ])
    return torch.unique(torch.cat([r, edge]))


def _check(got, want, K, name):
    got, want = got.double().cpu(), want.double()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    tol = 2e-6 * (K ** 0.5) * ref / 4 + 1e-6 * ref
    assert err <= tol, f"{name}: max|err| {err:.3e} > {tol:.3e} (ref max {ref:.3e})"


def test_gated_gemm_glds256x128():
    """the dominant class: (t .* cos2) @ W2^T on the row-major [K,N] weight, [131072,512] x [512,512]"""
    M, K, N = 131072, 512, 512
    a, g, w = _rnd(M, K, seed=1), torch.cos(_rnd(M, K, seed=2)), _rnd(K, N, seed=3, scale=K ** -0.5)
    out = hip.gemm(a.to(DEV), w.to(DEV), gate=g.to(DEV))
    assert _variant() == "glds256x128,gated", _variant()
    r = _rows(M)
    _check(out[r.to(DEV)], (a[r].double() * g[r].double()) @ w.double(), K, "gated glds GEMM")
    # ungated, transposed-B form (the forward 1x1 / NIN GEMMs of both nets: glds256x128 on B_COL)
    out = hip.gemm(a.to(DEV), w.t().contiguous().to(DEV), trans_b=True)
    assert _variant() == "glds256x128", _variant()
    _check(out[r.to(DEV)], a[r].double() @ w.double(), K, "ungated glds GEMM (B_COL)")
    out = hip.gemm(a.to(DEV), w.to(DEV))
    assert _variant() == "glds256x128", _variant()
    _check(out[r.to(DEV)], a[r].double() @ w.double(), K, "ungated glds GEMM (B_ROW)")


@pytest.mark.parametrize("hw,ci,co", [(32, 128, 128), (16, 256, 256)])
def test_conv3x3_fwd_and_dgrad_glds256x128(hw, ci, co):
    """3x3 convolutions of the score net at batch 128 (A_CONV on the direct-to-LDS kernel): forward and data-grad;
    fp64 conv on 4 whole images (first, last, two interior) = 4 * hw^2 >= 1024 rows x all channels"""
    Nb = 128
    x, w = _rnd(Nb, hw, hw, ci, seed=4), _rnd(3, 3, ci, co, seed=5, scale=(9 * ci) ** -0.5)
    imgs = [0, 41, 86, Nb - 1]
    y = hip.conv2d(x.to(DEV), w.to(DEV))
    assert _variant().startswith("glds256x128"), _variant()
    w_oihw = w.permute(3, 2, 0, 1).double()
    want = F.conv2d(x[imgs].permute(0, 3, 1, 2).double(), w_oihw, padding=1).permute(0, 2, 3, 1)
    _check(y[imgs], want, 9 * ci, f"conv fwd {hw}x{hw} {ci}->{co}")
    dy = _rnd(Nb, hw, hw, co, seed=6)
    dx = hip.conv2d_dgrad(dy.to(DEV), w.to(DEV), (hw, hw))
    v = _variant()
    assert v.startswith("glds256x128") or v == "big256x256", v
    want = F.conv_transpose2d(dy[imgs].permute(0, 3, 1, 2).double(), w_oihw, padding=1).permute(0, 2, 3, 1)
    _check(dx[imgs], want, 9 * co, f"conv dgrad {hw}x{hw} {co}->{ci} ({v})")


def test_wgrad_gemm_big256x256():
    """1x1 weight-grad GEMM of the flow: a^T [512 x 131072] x dc [131072 x 512], split-K on 256x256 big tiles; gated
    form ((a .* g)^T dc) as the VJP chain's backward uses it.  64 sampled output rows x all 512 columns in fp64."""
    M, K = 131072, 512
    a, dc, g = _rnd(M, K, seed=7), _rnd(M, K, seed=8), torch.cos(_rnd(M, K, seed=9))
    rows = torch.tensor(sorted(set(list(range(0, 512, 9)) + [255, 256, 511])))
    for gate in (None, g):
        out = hip.gemm(a.to(DEV), dc.to(DEV), trans_a=True, splitk=True, gate=None if gate is None else gate.to(DEV))
        assert _variant() == "big256x256", _variant()
        ag = a[:, rows].double() * (1.0 if gate is None else gate[:, rows].double())
        _check(out[rows.to(DEV)], ag.t() @ dc.double(), M, "wgrad GEMM big256x256" + (" gated" if gate is not None else ""))


def test_conv_wgrad_production_variant():
    """3x3 weight grad of a 16x16 256->256 score-net conv at batch 128 (A_WGRAD; round 3: the direct-to-LDS kernel's
    [16 k][256 m] weight-gradient image + split-K, before that big tiles + split-K): 8 input channels x all taps x all
    output channels in fp64"""
    Nb, hw, ci, co = 128, 16, 256, 256
    x, dy = _rnd(Nb, hw, hw, ci, seed=10), _rnd(Nb, hw, hw, co, seed=11)
    dw = hip.conv2d_wgrad(x.to(DEV), dy.to(DEV), 3, 3)
    assert _variant() in ("glds256x128,splitk", "big256x256", "big256x128"), _variant()
    sub = [0, 3, 64, 127, 128, 200, 254, 255]
    want = torch.nn.grad.conv2d_weight(x[..., sub].permute(0, 3, 1, 2).double(), (co, len(sub), 3, 3),
                                       dy.permute(0, 3, 1, 2).double(), padding=1)       # [co, sub, 3, 3]
    _check(dw[:, :, sub, :], want.permute(2, 3, 1, 0), Nb * hw * hw, "conv wgrad big tile")


@pytest.mark.parametrize("hw,c", [(32, 3), (16, 12)])
def test_thin_conv3x3_at_full_M(hw, c):
    """the flow's skinny-input 3x3 convolutions at batch 128 (thin-K kernels, M = 131072 / 32768 rows): forward with
    bias + per-image conditioning bias, and the transposed form; fp64 on 4 whole images"""
    Nb, co = 128, 512
    x, w = _rnd(Nb, hw, hw, c, seed=12), _rnd(3, 3, c, co, seed=13, scale=(9 * c) ** -0.5)
    b, b2 = _rnd(co, seed=14), _rnd(Nb, co, seed=15)
    imgs = [0, 17, 100, Nb - 1]
    y = hip.conv2d(x.to(DEV), w.to(DEV), bias=b.to(DEV), bias2=b2.to(DEV))
    assert _variant() == ("thin_conv3x3_flat" if c <= 3 else "thin_conv3x3_k128"), _variant()
    want = F.conv2d(x[imgs].permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), b.double(), padding=1)
    want = want.permute(0, 2, 3, 1) + b2[imgs].double()[:, None, None, :]
    _check(y[imgs], want, 9 * c, f"thin conv {c}->512")
    # skinny OUTPUT (512 -> c): one GEMM with N = 9c + tap gather (the generic / thin-N tiles at M = 131072)
    xw, w2 = _rnd(Nb, hw, hw, co, seed=16), _rnd(3, 3, co, c, seed=17, scale=(9 * co) ** -0.5)
    y2 = hip.conv2d(xw.to(DEV), w2.to(DEV))
    want = F.conv2d(xw[imgs].permute(0, 3, 1, 2).double(), w2.permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1)
    _check(y2[imgs], want, 9 * co, f"skinny-output conv 512->{c}")
    # gated transposed conv with a skinny output: (v .* gate) through W1^T (the VJP chain's last link)
    gate = torch.cos(_rnd(Nb, hw, hw, co, seed=18))
    t = hip.conv2d_dgrad(xw.to(DEV), w.to(DEV), (hw, hw), gate=gate.to(DEV))
    want = F.conv_transpose2d((xw[imgs] * gate[imgs]).permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(),
                              padding=1).permute(0, 2, 3, 1)
    _check(t[imgs], want, 9 * co, f"gated transposed conv 512->{c}")


@pytest.mark.parametrize("M,J", [(131072, 27), (32768, 108)])
@pytest.mark.parametrize("want_t2", [False, True])
def test_ires_chain_fp32_at_benchmark_size(M, J, want_t2):
    """The benchmark's dominant kernel at its benchmark size: the fp32 back-to-back chain kernel
    `ires_chain_kernel<1|4,false,want_t2>` at M = 131072 / J = 27 (scale 0 at batch 128, 2048 workgroups) and
    M = 32768 / J = 108 (scale 1), with and without the stored 512-wide intermediate.  >= 4096 sampled rows (both ends,
    workgroup edges) x all columns against ((t3 .* d2) W2T .* d1) W1p evaluated in fp64 on the host."""
    K = N = 512
    JP = 32 if J <= 32 else 128
    t3, d2, d1 = _rnd(M, K, seed=21), torch.cos(_rnd(M, K, seed=22)), torch.cos(_rnd(M, N, seed=23))
    w2t = _rnd(K, N, seed=24, scale=K ** -0.5)
    w1p = torch.zeros(N, JP)
    w1p[:, :J] = _rnd(N, J, seed=25, scale=N ** -0.5)
    assert hip.compute_dtype() == "f32"
    out = hip.ires_chain(t3.to(DEV), d2.to(DEV), w2t.to(DEV), d1.to(DEV), w1p.to(DEV), J, want_t2=want_t2)
    assert out is not None, "the chain kernel declined its own benchmark geometry"
    P, t2 = out if want_t2 else (out, None)
    assert P.shape == (M, J)
    r = torch.unique(torch.cat([_rows(M), torch.arange(63, M, M // 512), torch.arange(64, M, M // 512)]))
    assert r.numel() >= 4096
    t2_ref = (t3[r].double() * d2[r].double()) @ w2t.double()
    if want_t2:
        _check(t2[r.to(DEV)], t2_ref, K, f"chain t2 M={M}")
    _check(P[r.to(DEV)], (t2_ref * d1[r].double()) @ w1p[:, :J].double(), K + N, f"chain panel M={M} J={J}")
