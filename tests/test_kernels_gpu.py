"""Per-kernel parity: every HIP kernel (called through the C ABI via indm_amd.hip) against the
plain-PyTorch statement of the same op in tests/_emu.py, evaluated on CPU."""
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


def close(a, b, rtol=2e-5, atol=2e-5, name=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"{name}: max err {err:.3e} (ref max {ref:.3e})"


def run_both(fn, *args, **kw):
    gargs = [a.to(DEV) if torch.is_tensor(a) else a for a in args]
    gkw = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    out = getattr(hip, fn)(*gargs, **gkw)
    ref = getattr(_emu, fn)(*[a.double() if torch.is_tensor(a) else a for a in args],
                            **{k: (v.double() if torch.is_tensor(v) else v) for k, v in kw.items()})
    return out, ref


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 512, 512), (100, 70, 33), (1, 512, 64), (300, 3, 512),
                                   (27, 512, 1000), (129, 130, 17), (4096, 512, 512)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm(M, N, K, ta, tb):
    a = rnd(K, M) if ta else rnd(M, K)
    b = rnd(N, K, seed=1) if tb else rnd(K, N, seed=1)
    out, ref = run_both("gemm", a, b, trans_a=ta, trans_b=tb)
    close(out, ref, rtol=1e-5, atol=1e-5 * math.sqrt(K), name="gemm")


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_gated(ta, tb):
    M, N, K = 300, 70, 136
    a = rnd(K, M) if ta else rnd(M, K)
    g = rnd(*a.shape, seed=5)
    b = rnd(N, K, seed=1) if tb else rnd(K, N, seed=1)
    out, ref = run_both("gemm", a, b, trans_a=ta, trans_b=tb, gate=g)
    close(out, ref, rtol=1e-5, atol=1e-4, name="gemm gated")


@pytest.mark.parametrize("N,H,W,C,Co", [(8, 32, 32, 3, 512), (4, 16, 16, 12, 512), (2, 64, 64, 3, 128), (16, 16, 16, 1, 256),
                                         (3, 32, 32, 3, 260), (2, 32, 32, 8, 128), (1, 64, 16, 2, 128),
                                         (5, 32, 32, 12, 512), (5, 16, 16, 3, 128), (33, 32, 32, 3, 256), (7, 16, 16, 12, 384)])
@pytest.mark.parametrize("sign", [-1, 1])
def test_conv3x3_thin(N, H, W, C, Co, sign):
    """skinny-input 3x3 conv with the tap gather inside the thin-K GEMM loader (flat C <= 3 form and C % 4 == 0 form)"""
    x, w = rnd(N, H, W, C), rnd(9 * C, Co, seed=1, scale=0.3)
    out = hip.conv3x3_thin(x.to(DEV), w.to(DEV), sign)
    assert out is not None, "geometry should be covered"
    ref = _emu.conv3x3_thin(x.double(), w.double(), sign)
    close(out, ref, rtol=1e-5, atol=1e-4, name="conv3x3_thin")
    bias, bias2, res = rnd(Co, seed=2), rnd(N, Co, seed=3), rnd(N, H, W, Co, seed=4)
    out = hip.conv3x3_thin(x.to(DEV), w.to(DEV), sign, 0.5, bias.to(DEV), bias2.to(DEV), res.to(DEV), -1.5)
    ref = _emu.conv3x3_thin(x.double(), w.double(), sign, 0.5, bias.double(), bias2.double(), res.double(), -1.5)
    close(out, ref, rtol=1e-5, atol=1e-4, name="conv3x3_thin + epilogue")
    # the epilogue WITHOUT a residual (the row-fragment kernel of round 5 when the geometry is its own: C = 3 | 12, W = 16 | 32,
    # Co % 128 == 0; a residual sends the call to the staged kernel): bias + per-image bias, alpha != 1 and alpha == 1 forms
    for alpha in (0.5, 1.0):
        out = hip.conv3x3_thin(x.to(DEV), w.to(DEV), sign, alpha, bias.to(DEV), bias2.to(DEV))
        ref = _emu.conv3x3_thin(x.double(), w.double(), sign, alpha, bias.double(), bias2.double())
        close(out, ref, rtol=1e-5, atol=1e-4, name=f"conv3x3_thin + bias + bias2, alpha {alpha}")


def test_conv_dgrad_gated():
    dy, g = rnd(2, 16, 16, 64), rnd(2, 16, 16, 64, seed=3)
    for ci in (3, 12, 32):
        w = rnd(3, 3, ci, 64, seed=1, scale=0.2)
        out, ref = run_both("conv2d_dgrad", dy, w, (16, 16), gate=g)
        close(out, ref, rtol=1e-5, atol=1e-4, name="dgrad gated")


def test_gemm_epilogue_batched_splitk():
    a, b = rnd(6, 64, 48), rnd(6, 48, 40, seed=2)
    bias, res = rnd(40, seed=3), rnd(6, 64, 40, seed=4)
    out, ref = run_both("gemm", a, b, alpha=0.5, bias=bias, res=res, beta=-2.0)  # alpha*(ab+bias)+beta*res
    close(out, ref, name="gemm batched")
    a, b = rnd(512, 96), rnd(96, 72, seed=5)
    bias2 = rnd(4, 72, seed=6)
    out, ref = run_both("gemm", a, b, bias2=bias2, rows_per_group=128)
    close(out, ref, name="gemm bias2")
    # weight-grad shape: tiny output, long reduction -> split-K path
    a, b = rnd(20000, 64), rnd(20000, 96, seed=7)
    out, ref = run_both("gemm", a, b, trans_a=True, splitk=True)
    close(out, ref, rtol=1e-5, atol=2e-3, name="gemm splitk")
    # attention shapes: batched Q K^T and P V
    q, k = rnd(8, 256, 64), rnd(8, 256, 64, seed=8)
    out, ref = run_both("gemm", q, k, trans_b=True)
    close(out, ref, name="qk^t")


# ---- big-tile kernel (igemm_big_kernel: 256x256 / 256x128 block tiles, one wave per SIMD) ----------------------
@pytest.mark.parametrize("M,N,K,tb", [(28672 + 37, 512, 80, False), (28672 + 37, 512, 80, True),
                                      (57344 + 5, 128, 48, False), (57344, 192, 16, True),
                                      (30000, 500, 100, True), (29000, 260, 36, False)])
def test_gemm_big_tile(M, N, K, tb):
    a, g = rnd(M, K), rnd(M, K, seed=5)
    b = rnd(N, K, seed=1) if tb else rnd(K, N, seed=1)
    out, ref = run_both("gemm", a, b, trans_b=tb)
    close(out, ref, rtol=1e-5, atol=1e-5 * math.sqrt(K), name="big gemm")
    rpg = M // 7 if M % 7 == 0 else M
    bias, bias2, res = rnd(N, seed=2), rnd(M // rpg, N, seed=3), rnd(M, N, seed=4)
    out, ref = run_both("gemm", a, b, trans_b=tb, gate=g, alpha=0.5, bias=bias, bias2=bias2, rows_per_group=rpg,
                        res=res, beta=-1.5)
    close(out, ref, rtol=1e-5, atol=1e-4, name="big gemm gated + epilogue")


# ---- thin kernels (igemm_thin.hip): HBM-streaming GEMMs with one dimension <= 128 ---------------------------------
@pytest.mark.parametrize("M,N,K", [(4096, 512, 27), (2048 + 77, 512, 27), (1500, 132, 5), (3000, 256, 32),
                                   (4096 + 3, 512, 108), (1024, 128, 128), (2000, 640, 36), (1111, 512, 1)])
def test_gemm_thin_k(M, N, K):
    """thin-K form (3x3 conv with a skinny input after tap_spread): flat A tiles (lda == K <= 32) and chunked K"""
    a, b = rnd(M, K), rnd(K, N, seed=1)
    out, ref = run_both("gemm", a, b)
    close(out, ref, rtol=1e-5, atol=1e-5 * math.sqrt(K), name="thin-K gemm")
    for rpg in (M, 7, 100, 8, 1024):
        if M % rpg:
            continue
        bias, bias2, res = rnd(N, seed=2), rnd(M // rpg, N, seed=3), rnd(M, N, seed=4)
        out, ref = run_both("gemm", a, b, alpha=0.5, bias=bias, bias2=bias2, rows_per_group=rpg, res=res, beta=-1.5)
        close(out, ref, rtol=1e-5, atol=1e-4, name=f"thin-K gemm + epilogue rpg={rpg}")


@pytest.mark.parametrize("M,N,K", [(131072, 512, 27), (131072 + 128 + 5, 512, 4), (65536, 256, 1), (98304, 512, 32)])
def test_gemm_thin_k_persistent_walk(M, N, K):
    """the flat thin-K kernel with MORE row tiles than workgroups: every workgroup walks several tiles and prefetches
    the next one while it multiplies (the small cases above stop after one tile per workgroup).  Round 4 found the
    prefetch -- inline-asm loads awaited by a later asm wait -- copied out of its registers before the wait by the
    compiler; with K <= 4 the multiply loop between the two is shorter than any memory latency."""
    a, b = rnd(M, K), rnd(K, N, seed=1)
    for rep in range(3):
        out = hip.gemm(a.to(DEV), b.to(DEV))
        assert "thin_k" in hip.lib().indm_igemm_last_variant().decode()
        # every row against fp64 on the GPU side of the comparison budget: rows are independent, check a strided sample
        idx = torch.arange(0, M, 37)
        ref = a[idx].double() @ b.double()
        close(out[idx.to(DEV)], ref, rtol=1e-5, atol=1e-5 * math.sqrt(K), name=f"thin-K persistent rep {rep}")
        idx = torch.arange(M - 300, M)
        close(out[idx.to(DEV)], a[idx].double() @ b.double(), rtol=1e-5, atol=1e-5 * math.sqrt(K), name="thin-K tail")


@pytest.mark.parametrize("T,N,P", [(27, 512, 8192), (108, 512, 4096), (5, 256, 4096 + 32), (128, 640, 6400), (12, 260, 4128)])
def test_gemm_thin_wgrad(T, N, P):
    """weight-grad forms with a skinny side: thin^T wide -> [T, N] and (wide .* gate)^T thin -> [N, T]"""
    thin, wide, gate = rnd(P, T), rnd(P, N, seed=1), rnd(P, N, seed=2)
    out, ref = run_both("gemm", thin, wide, trans_a=True, splitk=True, alpha=0.5)
    close(out, ref, rtol=1e-5, atol=3e-3, name="thin^T wide")
    out, ref = run_both("gemm", wide, thin, trans_a=True, splitk=True)
    close(out, ref, rtol=1e-5, atol=3e-3, name="wide^T thin")
    out, ref = run_both("gemm", wide, thin, trans_a=True, splitk=True, gate=gate, alpha=2.0)
    close(out, ref, rtol=1e-5, atol=3e-3, name="(wide.gate)^T thin")


@pytest.mark.parametrize("N,H,W,Ci,Co,k,s,p", [(64, 32, 32, 16, 128, 3, 1, 1), (32, 32, 32, 32, 512, 3, 1, 1),
                                               (57, 32, 32, 16, 256, 1, 1, 0), (64, 32, 32, 32, 128, 3, 2, 1)])
def test_conv_big_tile(N, H, W, Ci, Co, k, s, p):
    x, w = rnd(N, H, W, Ci), rnd(k, k, Ci, Co, seed=1, scale=0.2)
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    bias, bias2 = rnd(Co, seed=2), rnd(N, Co, seed=3)
    out, ref = run_both("conv2d", x, w, stride=s, pad=p, bias=bias, bias2=bias2, alpha=0.7,
                        res=rnd(N, OH, OW, Co, seed=9), beta=0.3)
    close(out, ref, rtol=1e-5, atol=1e-4, name="big conv fwd")
    # data gradient of a conv whose INPUT is the wide tensor: dy has Ci channels here, dx has Co
    w2 = rnd(k, k, Co, Ci, seed=6, scale=0.2)
    dy = rnd(N, OH, OW, Ci, seed=4)
    out, r2 = run_both("conv2d_dgrad", dy, w2, (H, W), stride=s, pad=p, alpha=2.0, res=rnd(N, H, W, Co, seed=5),
                       beta=0.5)
    close(out, r2, rtol=1e-5, atol=1e-4, name="big conv dgrad")


@pytest.mark.parametrize("M,N,K", [(512, 512, 24576), (256, 768, 16384), (512, 260, 40000 // 16 * 16)])
def test_wgrad_gemm_big_tile_splitk(M, N, K):
    """weight-grad GEMM (A stored [K,M]) on the 256x256 big-tile split-K path, with and without the gate"""
    a, b, g = rnd(K, M), rnd(K, N, seed=1), rnd(K, M, seed=2)
    out, ref = run_both("gemm", a, b, trans_a=True, splitk=True, alpha=0.5)
    close(out, ref, rtol=1e-5, atol=3e-3, name="big wgrad")
    out, ref = run_both("gemm", a, b, trans_a=True, splitk=True, gate=g)
    close(out, ref, rtol=1e-5, atol=3e-3, name="big wgrad gated")


@pytest.mark.parametrize("N,H,W,C,Co", [(5, 32, 32, 3, 512), (4, 16, 16, 12, 512), (33, 32, 32, 3, 256), (7, 16, 16, 12, 384),
                                        (2, 32, 32, 12, 128)])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_conv3x3_thin_act_equals_the_two_steps(N, H, W, C, Co, mode):
    """conv + Sin in one launch (the g-net's first level inside the block drivers): the pre-activation is never written and
    (a, d) are BIT-identical to conv3x3_thin followed by sincos_fwd, with a bias, a per-image bias, both, neither; fp32 and
    bf16-rounded operands; against the fp64 reference within the convolution's tolerance"""
    hip.set_compute_dtype(mode)
    try:
        x, w = rnd(N, H, W, C, seed=3), rnd(9 * C, Co, seed=4, scale=0.3)
        bias, bias2 = rnd(Co, seed=5), rnd(N, Co, seed=6)
        for b, b2 in ((None, None), (bias, None), (bias, bias2), (None, bias2)):
            bd = b.to(DEV) if b is not None else None
            b2d = b2.to(DEV) if b2 is not None else None
            out = hip.conv3x3_thin_act(x.to(DEV), w.to(DEV), -1, bd, b2d)
            assert out is not None, "geometry should be covered"
            p = hip.conv3x3_thin(x.to(DEV), w.to(DEV), -1, 1.0, bd, b2d)
            a, d = hip.sincos_fwd(p)
            assert torch.equal(out[0], a) and torch.equal(out[1], d)
        if mode == "f32":
            ra, rd = _emu.conv3x3_thin_act(x.double(), w.double(), -1, bias.double(), bias2.double())
            fa, fd = hip.conv3x3_thin_act(x.to(DEV), w.to(DEV), -1, bias.to(DEV), bias2.to(DEV))
            close(fa, ra, rtol=1e-5, atol=1e-4, name="conv + Sin: a")
            close(fd, rd, rtol=1e-5, atol=2e-3, name="conv + Sin: d")   # (d/dv cos(2 pi v) = 2 pi: the conv's 1e-4 x 2 pi, rounded up)
    finally:
        hip.set_compute_dtype("f32")
    # not covered: 64-pixel-wide images, a residual-free call with 48 channels -> None, the caller runs the two steps
    assert hip.conv3x3_thin_act(rnd(2, 64, 64, 3).to(DEV), rnd(27, 512).to(DEV)) is None


@pytest.mark.parametrize("M,N,K,rpg", [(57344, 512, 512, 1024), (57344 + 72, 128, 64, 8), (65536, 256, 48, 256), (57344, 192, 16, 64),
                                        (114688 + 8, 64, 32, 1)])
def test_gemm_sin_equals_the_two_steps(M, N, K, rpg):
    """GEMM + Sin in one launch (the g-net's second level inside the block drivers): (a, d) BIT-identical to gemm(trans_b) followed
    by sincos_fwd -- bias, per-image bias, both, neither; a ragged last tile; a short K -- and within the GEMM's tolerance of the
    fp64 reference; problems the direct-to-LDS kernel does not take return None"""
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2)
    bias, bias2 = rnd(N, seed=3), rnd((M + rpg - 1) // rpg, N, seed=4)
    for bb, b2 in ((None, None), (bias, None), (bias, bias2), (None, bias2)):
        bd = bb.to(DEV) if bb is not None else None
        b2d = b2.to(DEV) if b2 is not None else None
        out = hip.gemm_sin(a.to(DEV), b.to(DEV), bd, b2d, rpg)
        assert out is not None, "problem should be covered"
        assert hip.lib().indm_igemm_last_variant().decode() == "glds256x128,sin"
        p = hip.gemm(a.to(DEV), b.to(DEV), trans_b=True, bias=bd, bias2=b2d, rows_per_group=rpg)
        s, c = hip.sincos_fwd(p)
        assert torch.equal(out[0], s) and torch.equal(out[1], c)
    rows = torch.cat([torch.arange(0, 200), torch.arange(M - 200, M)])
    rs, rc = _emu.sincos_fwd(_emu.gemm(a.double(), b.double(), trans_b=True, bias=bias.double(), bias2=bias2.double(),
                                       rows_per_group=rpg))
    fs, fc = hip.gemm_sin(a.to(DEV), b.to(DEV), bias.to(DEV), bias2.to(DEV), rpg)
    close(fs[rows], rs[rows], rtol=1e-5, atol=1e-5 * math.sqrt(K), name="gemm + Sin: a")
    close(fc[rows], rc[rows], rtol=1e-5, atol=1e-4 * math.sqrt(K), name="gemm + Sin: d")
    assert hip.gemm_sin(rnd(512, 64).to(DEV), rnd(128, 64).to(DEV)) is None          # too few tiles
    assert hip.gemm_sin(rnd(57344, 24).to(DEV), rnd(128, 24).to(DEV)) is None        # K % 16


@pytest.mark.parametrize("M,N,K,rps", [(57344, 512, 512, 1024), (65536, 128, 64, 256), (114688, 256, 32, 64), (57344 + 64, 192, 48, 64),
                                        (114688 + 128, 64, 16, 128)])
def test_gemm_ires_bwd_act_equals_the_two_steps(M, N, K, rps):
    """GEMM + Sin-level backward in one launch (the block drivers' da1 = dp2 W2^T -> dp1, dt2, per-image sums): dp, dt and the
    segment sums BIT-identical to gemm(trans_b) followed by ires_bwd_act (the epilogue adds a chunk's rows in the elementwise
    kernel's order, across the two lane halves), with and without dt / segment sums; fp64 reference within tolerance"""
    x, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2)
    du, d, a, t = (rnd(M, N, seed=10 + i) for i in range(4))
    xd, bd, dud, dd, ad, td = (z.to(DEV) for z in (x, b, du, d, a, t))
    da = hip.gemm(xd, bd, trans_b=True)
    for want_dt, seg in ((True, rps), (False, rps), (True, None), (False, None)):
        out = hip.gemm_ires_bwd_act(xd, bd, dud, dd, ad, td, want_dt, seg)
        assert out is not None, "problem should be covered"
        assert hip.lib().indm_igemm_last_variant().decode() == "glds256x128,ires_bwd_act"
        ref = hip.ires_bwd_act(da, dud, dd, ad, td, want_dt, seg)
        for o, r, name in zip(out, ref, ("dp", "dt", "seg")):
            assert (o is None) == (r is None), name
            if o is not None:
                assert torch.equal(o, r), f"{name}: max |diff| {float((o - r).abs().max())}"
    rows = torch.cat([torch.arange(0, 128), torch.arange(M - 128, M)])
    rdp, rdt, _ = _emu.ires_bwd_act(_emu.gemm(x[rows].double(), b.double(), trans_b=True), du[rows].double(), d[rows].double(),
                                    a[rows].double(), t[rows].double(), True)
    dp, dt, _ = hip.gemm_ires_bwd_act(xd, bd, dud, dd, ad, td, True, None)
    close(dp[rows], rdp, rtol=1e-5, atol=2e-4 * math.sqrt(K), name="gemm + ires_bwd_act: dp")
    close(dt[rows], rdt, rtol=1e-6, atol=1e-6, name="gemm + ires_bwd_act: dt")
    assert hip.gemm_ires_bwd_act(xd[:512], bd, dud[:512], dd[:512], ad[:512], td[:512]) is None   # too few tiles


@pytest.mark.parametrize("M,N,K,rps", [(57344, 512, 512, 1024), (65536, 128, 64, 256), (57344 + 192, 192, 48, 64)])
def test_gemm_gated_ires_bwd_act_equals_the_two_steps(M, N, K, rps):
    """the level whose GEMM produces du (gated NN): du, dp and the segment sums BIT-identical to gemm(gate=) followed by
    ires_bwd_act, with and without segment sums"""
    x, g, b = rnd(M, K, seed=1), rnd(M, K, seed=5), rnd(K, N, seed=2, scale=0.2)
    da, d, a, t = (rnd(M, N, seed=10 + i) for i in range(4))
    xd, gd, bd, dad, dd, ad, td = (z.to(DEV) for z in (x, g, b, da, d, a, t))
    du = hip.gemm(xd, bd, gate=gd)
    for seg in (rps, None):
        out = hip.gemm_gated_ires_bwd_act(xd, gd, bd, dad, dd, ad, td, seg)
        assert out is not None, "problem should be covered"
        assert hip.lib().indm_igemm_last_variant().decode() == "glds256x128,gated,ires_bwd_act"
        dp, _, sg = hip.ires_bwd_act(dad, du, dd, ad, td, False, seg)
        assert torch.equal(out[0], du) and torch.equal(out[1], dp)
        assert (sg is None) == (out[2] is None) and (sg is None or torch.equal(out[2], sg))
    assert hip.gemm_gated_ires_bwd_act(xd[:512], gd[:512], bd, dad[:512], dd[:512], ad[:512], td[:512]) is None


# ---- "bf16 compute / fp32 accumulate" mode (BASELINE configs #3-#5) -----------------------------------------------
def _r16(t):
    return t.bfloat16().float()


@pytest.fixture
def bf16_mode():
    hip.set_compute_dtype("bf16")
    yield
    hip.set_compute_dtype("f32")


@pytest.mark.parametrize("M,N,K,ta,tb", [(300, 70, 136, False, False), (513, 130, 48, False, True),
                                         (28672 + 37, 512, 80, False, True), (96, 512, 20000, True, False),
                                         (512, 512, 24576, True, False), (27, 512, 1000, False, False)])
def test_bf16_gemm(bf16_mode, M, N, K, ta, tb):
    """exact products of bf16-rounded inputs, fp32 accumulation: compare with the fp64 product of the rounded inputs"""
    a = rnd(K, M) if ta else rnd(M, K)
    b = rnd(N, K, seed=1) if tb else rnd(K, N, seed=1)
    out = hip.gemm(a.to(DEV), b.to(DEV), trans_a=ta, trans_b=tb, splitk=ta)
    ref = _emu.gemm(_r16(a).double(), _r16(b).double(), trans_a=ta, trans_b=tb)
    close(out, ref, rtol=1e-5, atol=1e-5 * math.sqrt(K), name="bf16 gemm")
    g = rnd(*a.shape, seed=5)
    bias, res = rnd(N, seed=2), rnd(M, N, seed=4)
    out = hip.gemm(a.to(DEV), b.to(DEV), trans_a=ta, trans_b=tb, gate=g.to(DEV), alpha=0.5, bias=bias.to(DEV),
                   res=res.to(DEV), beta=-1.5, splitk=ta)
    ref = _emu.gemm(_r16(a * g).double(), _r16(b).double(), trans_a=ta, trans_b=tb, alpha=0.5, bias=bias.double(),
                    res=res.double(), beta=-1.5)
    close(out, ref, rtol=1e-5, atol=1e-4 * math.sqrt(K / 64 + 1), name="bf16 gemm gated + epilogue")


@pytest.mark.parametrize("N,H,W,Ci,Co,k,s,p", [(2, 16, 16, 32, 64, 3, 1, 1), (32, 32, 32, 32, 128, 3, 1, 1),
                                               (2, 16, 16, 48, 96, 3, 2, 1), (2, 16, 16, 3, 64, 3, 1, 1),
                                               (2, 16, 16, 64, 12, 3, 1, 1)])
def test_bf16_conv(bf16_mode, N, H, W, Ci, Co, k, s, p):
    x, w = rnd(N, H, W, Ci), rnd(k, k, Ci, Co, seed=1, scale=0.2)
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    out = hip.conv2d(x.to(DEV), w.to(DEV), stride=s, pad=p)
    close(out, _emu.conv2d(_r16(x).double(), _r16(w).double(), stride=s, pad=p), rtol=1e-5, atol=1e-4, name="bf16 conv")
    dy = rnd(N, OH, OW, Co, seed=4)
    out = hip.conv2d_dgrad(dy.to(DEV), w.to(DEV), (H, W), stride=s, pad=p)
    close(out, _emu.conv2d_dgrad(_r16(dy).double(), _r16(w).double(), (H, W), stride=s, pad=p), rtol=1e-5, atol=1e-4,
          name="bf16 dgrad")
    out = hip.conv2d_wgrad(x.to(DEV), dy.to(DEV), k, k, stride=s, pad=p)
    close(out, _emu.conv2d_wgrad(_r16(x).double(), _r16(dy).double(), k, k, stride=s, pad=p), rtol=1e-5, atol=3e-3,
          name="bf16 wgrad")


CONV_CASES = [
    # N, H, W, Ci, Co, k, stride, pad
    (2, 16, 16, 32, 64, 3, 1, 1),
    (3, 8, 8, 128, 128, 3, 1, 1),
    (2, 32, 32, 3, 64, 3, 1, 1),      # first conv / flow 3->512 style (scalar A path)
    (2, 16, 16, 12, 48, 3, 1, 1),
    (2, 16, 16, 64, 3, 3, 1, 1),      # 512->3 style: GEMM(N=9*Co) + tap-gather path
    (2, 16, 16, 64, 12, 3, 1, 1),
    (2, 16, 16, 3, 64, 3, 1, 1),      # 3->512 style: its data-grad takes the skinny transposed path
    (2, 16, 16, 12, 64, 3, 1, 1),
    (2, 16, 16, 32, 3, 3, 1, 1),      # narrow wide-side: stays on the direct implicit GEMM (128x32 tiles)
    (2, 16, 16, 48, 96, 3, 2, 1),     # encoder stride-2
    (2, 16, 16, 48, 96, 1, 2, 0),     # encoder 1x1 stride-2 shortcut
    (2, 17, 17, 16, 32, 3, 2, 0),     # VE conv_downsample_2d: FIR-padded 17x17 -> 8x8
    (1, 4, 4, 256, 256, 3, 1, 1),
]


@pytest.mark.parametrize("N,H,W,Ci,Co,k,s,p", CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(N, H, W, Ci, Co, k, s, p):
    x, w = rnd(N, H, W, Ci), rnd(k, k, Ci, Co, seed=1, scale=0.2)
    bias, bias2 = rnd(Co, seed=2), rnd(N, Co, seed=3)
    out, ref = run_both("conv2d", x, w, stride=s, pad=p, bias=bias, bias2=bias2, alpha=0.7, res=rnd(N, (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1, Co, seed=9), beta=0.3)
    close(out, ref, rtol=1e-5, atol=1e-4, name="conv fwd")
    dy = rnd(*ref.shape, seed=4)
    res = rnd(N, H, W, Ci, seed=5)
    out, r2 = run_both("conv2d_dgrad", dy, w, (H, W), stride=s, pad=p, alpha=2.0, res=res, beta=0.5)
    close(out, r2, rtol=1e-5, atol=1e-4, name="conv dgrad")
    out, r3 = run_both("conv2d_wgrad", x, dy, k, k, stride=s, pad=p)
    close(out, r3, rtol=1e-5, atol=2e-4, name="conv wgrad")


@pytest.mark.parametrize("Ci,Co", [(3, 64), (12, 64), (32, 48)])
def test_conv2d_wgrad_gated(Ci, Co):
    x, dy, g = rnd(2, 16, 16, Ci), rnd(2, 16, 16, Co, seed=1), rnd(2, 16, 16, Co, seed=2)
    out, ref = run_both("conv2d_wgrad", x, dy, 3, 3, alpha=0.5, dy_gate=g)
    close(out, ref, rtol=1e-5, atol=2e-4, name="wgrad gated")


def test_skinny_conv_sees_out_of_band_parameter_updates():
    """The fused optimiser rewrites parameter memory through the flat buffer (raw kernel, no autograd version bump):
    the packed-weight cache of the skinny conv forms must never serve a stale copy of a Parameter."""
    x = rnd(2, 16, 16, 64).to(DEV)
    w = torch.nn.Parameter(rnd(3, 3, 64, 3, seed=1, scale=0.2).to(DEV))
    y1 = hip.conv2d(x, w)
    alias = w.data.view(-1)
    hip.ema_update(alias, torch.zeros_like(alias), 1.0)      # shadow <- shadow - 1.0*(shadow - 0) = 0, in place
    y2 = hip.conv2d(x, w)
    assert float(y1.abs().max()) > 0 and float(y2.abs().max()) == 0.0
    # the transposed form (skinny INPUT) packs the weight the other way round: same guarantee
    dy = rnd(2, 16, 16, 3, seed=2).to(DEV)
    w2 = torch.nn.Parameter(rnd(3, 3, 64, 3, seed=3, scale=0.2).to(DEV))
    d1 = hip.conv2d_dgrad(dy, w2, (16, 16))
    a2 = w2.data.view(-1)
    hip.ema_update(a2, torch.zeros_like(a2), 1.0)
    d2 = hip.conv2d_dgrad(dy, w2, (16, 16))
    assert float(d1.abs().max()) > 0 and float(d2.abs().max()) == 0.0


def test_tap_gather_spread():
    P = rnd(2, 6, 5, 27)
    bias = rnd(3, seed=1)
    for sign in (1, -1):
        o, r = run_both("tap_gather", P, bias, sign, 0.5)
        close(o, r, name="tap_gather")
        y = rnd(2, 6, 5, 3, seed=2)
        o, r = run_both("tap_spread", y, sign, 2.0)
        close(o, r, name="tap_spread")
        # adjointness <gather(P), y> == <P, spread(y)>
        a = (_emu.tap_gather(P.double(), None, sign) * y.double()).sum()
        b = (P.double() * _emu.tap_spread(y.double(), sign)).sum()
        assert abs(float(a - b)) < 1e-9


def test_tap_gather_acc():
    P, m, a = rnd(2, 16, 16, 27), rnd(2, 16, 16, 3, seed=1), rnd(2, 16, 16, 3, seed=2)
    for sign in (-1, 1):
        (y, acc), (ry, racc) = run_both("tap_gather_acc", P, sign, mulp=m, acc_in=a, coeff=-0.75, alpha=1.0)
        close(y, ry, name="gather*mul")
        close(acc, racc, name="acc")
        (y, acc), (ry, racc) = run_both("tap_gather_acc", P, sign)
        close(y, ry, name="gather")
        assert acc is None and racc is None


def test_conv2d_big_wgrad_splitk():
    x, dy = rnd(16, 32, 32, 16), rnd(16, 32, 32, 32, seed=1)
    out, ref = run_both("conv2d_wgrad", x, dy, 3, 3, stride=1, pad=1)
    close(out, ref, rtol=1e-5, atol=2e-3, name="wgrad splitk")


@pytest.mark.parametrize("N,H,Ci,Co,k,s,p", [(32, 32, 32, 128, 3, 1, 1), (32, 32, 32, 256, 3, 1, 1),
                                             (64, 16, 64, 256, 3, 1, 1), (40, 32, 256, 128, 1, 1, 0),
                                             (32, 34, 32, 128, 3, 2, 0)])
def test_conv2d_wgrad_big_tile(N, H, Ci, Co, k, s, p):
    """convolution weight grads on the big-tile split-K path (256x128 and 256x256 tiles, strided, 1x1)"""
    OH = (H + 2 * p - k) // s + 1
    x, dy = rnd(N, H, H, Ci), rnd(N, OH, OH, Co, seed=1)
    out, ref = run_both("conv2d_wgrad", x, dy, k, k, stride=s, pad=p, alpha=0.25)
    close(out, ref, rtol=1e-5, atol=3e-3, name="big conv wgrad")


@pytest.mark.parametrize("N,HW,C", [(2, (32, 32), 128), (3, (16, 16), 256), (2, (16, 16), 384), (2, (4, 4), 512),
                                    (4, (8, 8), 16), (2, (16, 16), 64),
                                    # fused single-pass geometries and their edges: 12-channel groups at 32^2 (192 threads,
                                    # 16 chunks per thread), 8-channel groups, a 6-group net (group block of 2), a slab that
                                    # does not fill the workgroup, and 64^2 (too large for the fused path: three-kernel form)
                                    (2, (32, 32), 384), (2, (32, 32), 256), (3, (8, 8), 24), (2, (2, 2), 128),
                                    (1, (64, 64), 128), (5, (32, 32), 512)])
@pytest.mark.parametrize("act", [0, 1])
def test_groupnorm(N, HW, C, act):
    G = min(C // 4, 32)
    x = rnd(N, *HW, C) * 2 + 0.3
    gamma, beta = rnd(C, seed=1) + 1.5, rnd(C, seed=2)
    (y, mean, rstd), (yr, mr, rr) = run_both("groupnorm_fwd", x, gamma, beta, G, 1e-6, act=act)
    close(y, yr, name="gn y")
    close(mean, mr, name="gn mean")
    close(rstd, rr, rtol=1e-4, name="gn rstd")
    dy = rnd(N, *HW, C, seed=3)
    (dx, dg, db), (dxr, dgr, dbr) = run_both("groupnorm_bwd", dy, x, gamma, beta, mr.float(), rr.float(), G, act=act)
    close(dx, dxr, rtol=1e-4, atol=1e-4, name="gn dx")
    close(dg, dgr, rtol=1e-4, atol=1e-3, name="gn dgamma")
    close(db, dbr, rtol=1e-4, atol=1e-3, name="gn dbeta")


def test_groupnorm_dropout_statistics():
    x = rnd(4, 16, 16, 128).to(DEV)
    gamma, beta = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    y0, mean, rstd = hip.groupnorm_fwd(x, gamma, beta, 32, 1e-6, act=1)
    y1, _, _ = hip.groupnorm_fwd(x, gamma, beta, 32, 1e-6, act=1, drop_p=0.1, seed=1234)
    y2, _, _ = hip.groupnorm_fwd(x, gamma, beta, 32, 1e-6, act=1, drop_p=0.1, seed=1234)
    assert torch.equal(y1, y2)
    keep = (y1 != 0) | (y0 == 0)
    frac = keep.float().mean().item()
    assert abs(frac - 0.9) < 0.01, frac
    close(y1[keep], (y0 / 0.9)[keep], name="dropout scale")
    # backward regenerates the same mask: dx must vanish where the output was dropped w.r.t. dy path
    dy = torch.ones_like(x)
    dxa, _, _ = hip.groupnorm_bwd(dy * keep.float() / 0.9, x, gamma, beta, mean, rstd, 32, act=1)
    dxb, _, _ = hip.groupnorm_bwd(dy, x, gamma, beta, mean, rstd, 32, act=1, drop_p=0.1, seed=1234)
    close(dxb, dxa, rtol=1e-4, atol=1e-5, name="dropout bwd")


@pytest.mark.parametrize("rows,T", [(512, 256), (64, 16), (7, 100), (3, 1000)])
def test_softmax(rows, T):
    x = rnd(rows, T) * 4
    y, yr = run_both("softmax_fwd", x, scale=0.25)
    close(y, yr, name="softmax")
    dy = rnd(rows, T, seed=1)
    dx, dxr = run_both("softmax_bwd", dy, yr.float(), scale=0.25)
    close(dx, dxr, name="softmax bwd")


def test_reductions_elementwise():
    x, y = rnd(1000, 48), rnd(1000, 48, seed=1)
    for mode in (0, 1, 2):
        o, r = run_both("colreduce", x, y, mode=mode, alpha=0.5)
        close(o, r, rtol=1e-5, atol=1e-4, name=f"colreduce{mode}")
    x, y = rnd(5, 3, 32, 32), rnd(5, 3, 32, 32, seed=1)
    for mode in (0, 1, 2):
        o, r = run_both("rowreduce", x, y, mode=mode, alpha=-0.5)
        close(o, r, rtol=1e-5, atol=1e-4, name=f"rowreduce{mode}")
    p = rnd(3, 16, 16, 64) * 3
    (a, d), (ar, dr) = run_both("sincos_fwd", p)
    close(a, ar, name="sin")
    close(d, dr, name="cos")
    da, dd = rnd(3, 16, 16, 64, seed=2), rnd(3, 16, 16, 64, seed=3)
    o, r = run_both("sincos_bwd", da, dd, ar.float(), dr.float())
    close(o, r, name="sincos bwd")
    for act in (1, 2):
        o, r = run_both("act_fwd", p, act)
        close(o, r, name="act")
        o, r = run_both("act_bwd", da, p, act)
        close(o, r, name="act bwd")
    o, r = run_both("axpby", da, dd, 0.7, -1.3, 0.25)
    close(o, r, name="axpby")
    o, r = run_both("axpby", da.reshape(-1)[:1001].clone(), dd.reshape(-1)[:1001].clone(), 0.7, -1.3)
    close(o, r, name="axpby odd")
    o, r = run_both("mul", da, dd, alpha=2.0)
    close(o, r, name="mul")
    o, r = run_both("scale_rows", da, rnd(3, seed=9))
    close(o, r, name="scale_rows")


@pytest.mark.parametrize("nseg,rps,C", [(128, 1024, 512), (8, 256, 512), (5, 17, 130), (3, 64, 3), (1, 1000, 256)])
def test_segreduce(nseg, rps, C):
    x = rnd(nseg * rps, C)
    out, ref = run_both("segreduce", x, rps, alpha=0.5)
    close(out, ref, rtol=1e-5, atol=1e-5 * math.sqrt(rps), name="segreduce")


@pytest.mark.parametrize("n", [4096, 1000003, 7])
def test_mul_mul2(n):
    x, y1, y2 = rnd(n), rnd(n, seed=1), rnd(n, seed=2)
    out, ref = run_both("mul", x, y1, alpha=0.5)
    close(out, ref, name="mul")
    (o1, o2), (r1, r2) = run_both("mul2", x, y1, y2)
    close(o1, r1, name="mul2[0]")
    close(o2, r2, name="mul2[1]")


def test_resample_permutes():
    x = rnd(2, 8, 8, 12)
    for fn in ("upsample2x", "downsample2x", "squeeze2x"):
        o, r = run_both(fn, x)
        close(o, r, name=fn)
    x4 = rnd(2, 4, 4, 48)
    o, r = run_both("unsqueeze2x", x4)
    close(o, r, name="unsqueeze2x")
    assert torch.equal(hip.unsqueeze2x(hip.squeeze2x(x.to(DEV))).cpu(), x)
    xc = rnd(2, 5, 6, 7)
    o, r = run_both("nchw_to_nhwc", xc)
    close(o, r, name="nchw->nhwc")
    o, r = run_both("nhwc_to_nchw", xc)
    close(o, r, name="nhwc->nchw")
    # squeeze must agree with the reference's NCHW squeeze (squeeze.py:27-45) through the layout change
    xn = rnd(2, 3, 8, 8)
    ref = xn.reshape(2, 3, 4, 2, 4, 2).permute(0, 1, 3, 5, 2, 4).reshape(2, 12, 4, 4)
    got = hip.nhwc_to_nchw(hip.squeeze2x(hip.nchw_to_nhwc(xn.to(DEV)))).cpu()
    assert torch.equal(got, ref)
    assert torch.equal(hip.nhwc_to_nchw(hip.unsqueeze2x(hip.nchw_to_nhwc(ref.to(DEV)))).cpu(),
                       torch.pixel_shuffle(ref, 2))


def test_lop_normalize_multi():
    shapes = [(3, 3, 3, 512), (512, 512), (3, 3, 512, 3), (3, 3, 12, 64), (64, 12), (3, 3, 512, 12)]
    ws = [rnd(*sh, seed=i, scale=0.02 if i % 2 else 0.2) for i, sh in enumerate(shapes)]
    gw = [w.to(DEV) for w in ws]
    wns, l1s = hip.lop_normalize_multi_fwd(gw, 0.98)
    rwn, rl1 = _emu.lop_normalize_multi_fwd([w.double() for w in ws], 0.98)
    for a, b, c, d in zip(wns, rwn, l1s, rl1):
        close(a, b, rtol=1e-6, atol=1e-7, name="lop multi wn")
        close(c, d, rtol=2e-6, atol=1e-6, name="lop multi l1")
    dwn = [rnd(*sh, seed=10 + i) for i, sh in enumerate(shapes)]
    dws = hip.lop_normalize_multi_bwd([d.to(DEV) for d in dwn], gw, l1s, 0.98)
    rdw = _emu.lop_normalize_multi_bwd([d.double() for d in dwn], [w.double() for w in ws], rl1, 0.98)
    for a, b in zip(dws, rdw):
        close(a, b, rtol=1e-5, atol=1e-5, name="lop multi dw")


def test_lop_normalize():
    for shape, scale in [((3, 3, 3, 64), 0.5), ((1, 1, 64, 64), 0.01), ((3, 3, 32, 3), 0.2)]:
        w = rnd(*shape, scale=scale)
        (wn, l1), (wnr, l1r) = run_both("lop_normalize_fwd", w, 0.98)
        close(wn, wnr, name="lop wn")
        close(l1, l1r, name="lop l1")
        dwn = rnd(*shape, seed=1)
        dw, dwr = run_both("lop_normalize_bwd", dwn, w, l1r.float(), 0.98)
        close(dw, dwr, name="lop bwd")
        # and against autograd of the reference formula (lipschitz.py:350-359)
        wt = w.double().requires_grad_()
        sc = wt.abs().reshape(-1, shape[-1]).sum(0)
        (wt / torch.max(torch.ones(1, dtype=torch.double), sc / 0.98) * dwn.double()).sum().backward()
        close(dw, wt.grad, name="lop bwd autograd")


def test_sde_loss_sampler_kernels():
    B, shape = 5, (5, 32, 32, 3)
    out, z = rnd(*shape), rnd(*shape, seed=1)
    std, w = rnd(B, seed=2).abs() + 0.1, rnd(B, seed=3).abs() + 0.5
    for kind in (0, 1):
        for mode in (0, 1):
            for rm in (True, False):
                o, r = run_both("dsm_loss_fwd", out, z, std, w, kind, mode, rm)
                close(o, r, rtol=1e-5, atol=1e-4, name="dsm fwd")
                gl = rnd(B, seed=4)
                o, r = run_both("dsm_loss_bwd", gl, out, z, std, w, kind, mode, rm)
                close(o, r, rtol=1e-5, atol=1e-5, name="dsm bwd")
    o, r = run_both("sde_perturb", out, z, std, w)
    close(o, r, name="perturb")
    (xn, xm), (xnr, xmr) = run_both("affine_update", out, z, rnd(*shape, seed=5), std, w, rnd(B, seed=6))
    close(xn, xnr, name="affine new")
    close(xm, xmr, name="affine mean")
    (st, nc), (str_, ncr) = run_both("langevin_step", std, w, rnd(B, seed=7).abs(), 0.16)
    close(st, str_, name="langevin step")
    close(nc, ncr, name="langevin coef")


@pytest.mark.parametrize("shape", [(5, 32, 32, 3), (3, 64, 64, 3), (1, 8, 8, 4)])
def test_pc_step_entry_points_match_the_reference_formulas(shape):
    """indm_pc_step_{em,rd,langevin} / indm_hutch_div against the fp64 statement of the reference's update_fn
    (tests/_emu.py: sampling.py:191-197, 205-210, 279-290; likelihood.py:27-38) with explicit noise"""
    B = shape[0]
    x, out, z = rnd(*shape), rnd(*shape, seed=1), rnd(*shape, seed=2)
    f_lin, g, k = -rnd(B, seed=3).abs() * 5, rnd(B, seed=4).abs() + 0.3, -1.0 / (rnd(B, seed=5).abs() + 0.2)
    for pf in (False, True):
        (xn, xm), (xnr, xmr) = run_both("pc_step_em", x, out, f_lin, g, k, -1e-3, pf, z=z)
        close(xn, xnr, rtol=1e-5, atol=1e-5, name=f"em new pf={pf}")
        close(xm, xmr, rtol=1e-5, atol=1e-5, name=f"em mean pf={pf}")
        (xn, xm), (xnr, xmr) = run_both("pc_step_rd", x, out, f_lin * 1e-3, g, k, pf, z=z)
        close(xn, xnr, rtol=1e-5, atol=1e-5, name=f"rd new pf={pf}")
        close(xm, xmr, rtol=1e-5, atol=1e-5, name=f"rd mean pf={pf}")
    alpha = rnd(B, seed=6).abs() + 0.5
    (xn, xm), (xnr, xmr) = run_both("pc_step_langevin", x, out, k, alpha, 0.16, z=z)
    close(xn, xnr, rtol=2e-5, atol=1e-5, name="langevin new")
    close(xm, xmr, rtol=2e-5, atol=1e-5, name="langevin mean")
    d, dr = run_both("hutch_div", out, z)
    close(d, dr, rtol=1e-5, atol=1e-3, name="hutch div")
    # the composed op-level path (scale_rows + rowreduce + langevin_step + affine_update) gives the same update
    xg, og, zg, kg = x.to(DEV), out.to(DEV), z.to(DEV), k.to(DEV)
    grad = hip.scale_rows(og, kg)
    step, ncoef = hip.langevin_step(hip.rowreduce(grad, None, 1), hip.rowreduce(zg, None, 1), alpha.to(DEV), 0.16)
    cn, cm = hip.affine_update(xg, grad, zg, torch.ones_like(step), step, ncoef)
    close(xn, cn.cpu(), rtol=1e-5, atol=1e-6, name="langevin vs composed")
    close(xm, cm.cpu(), rtol=1e-5, atol=1e-6, name="langevin mean vs composed")


def test_pc_step_draws_the_same_noise_as_philox_randn():
    """noise drawn inside the update kernel == indm_philox_randn_f32 followed by the explicit-noise update, bit for
    bit, and the device state advances by the same amount (probability-flow steps included)"""
    shape, B = (6, 32, 32, 3), 6
    x, out = rnd(*shape).to(DEV), rnd(*shape, seed=1).to(DEV)
    f_lin, g, k = (-rnd(B, seed=3).abs()).to(DEV), (rnd(B, seed=4).abs() + 0.3).to(DEV), (-1.0 / (rnd(B, seed=5).abs() + 0.2)).to(DEV)
    alpha = (rnd(B, seed=6).abs() + 0.5).to(DEV)

    def state():
        return hip.to_device(torch.tensor([12345, 7], dtype=torch.int64), torch.device(DEV))

    calls = {
        "em": lambda z, ph: hip.pc_step_em(x, out, f_lin, g, k, -1e-3, False, z, ph),
        "rd": lambda z, ph: hip.pc_step_rd(x, out, f_lin * 1e-3, g, k, False, z, ph),
        "rd_pf": lambda z, ph: hip.pc_step_rd(x, out, f_lin * 1e-3, g, k, True, z, ph),
        "langevin": lambda z, ph: hip.pc_step_langevin(x, out, k, alpha, 0.16, z, ph),
    }
    for name, fn in calls.items():
        s1, s2 = state(), state()
        z = hip.philox_randn(shape, s1)
        a_new, a_mean = fn(z, None)
        b_new, b_mean = fn(None, s2)
        assert torch.equal(a_new, b_new) and torch.equal(a_mean, b_mean), name
        assert torch.equal(s1, s2) and int(s2[1]) == 7 + x.numel() // 4, (name, s1, s2)
    with pytest.raises(RuntimeError):
        hip.pc_step_em(x, out, f_lin, g, k, -1e-3, False, None, None)           # a stochastic step needs a noise source
    with pytest.raises(RuntimeError):
        hip.pc_step_em(x, out, f_lin, g, k, 1e-3, False, hip.philox_randn(shape, state()), None)   # dt must be < 0
    with pytest.raises(RuntimeError):
        hip.pc_step_langevin(x, out, k, alpha, 0.16, None, None)


def test_optimizer_kernels():
    n = 100003
    p, g = rnd(n), rnd(n, seed=1) * 3
    m, v, ema = rnd(n, seed=2) * 0.1, rnd(n, seed=3).abs() * 0.01, rnd(n, seed=4)
    gq = hip.sqnorm(g.to(DEV))
    close(gq, _emu.sqnorm(g), rtol=1e-5, name="sqnorm")
    P, M, V, E = p.clone().to(DEV), m.clone().to(DEV), v.clone().to(DEV), ema.clone().to(DEV)
    hip.adamw_ema(P, g.to(DEV), M, V, E, 2e-4, 0.9, 0.99, 1e-8, 0.01, 7, 1.0, gq, 1.0, 1 - 0.999)
    pr, mr, vr, er = p.double(), m.double(), v.double(), ema.double()
    _emu.adamw_ema(pr, g.double(), mr, vr, er, 2e-4, 0.9, 0.99, 1e-8, 0.01, 7, 1.0, _emu.sqnorm(g).double(), 1.0,
                   1 - 0.999)
    close(P, pr, rtol=1e-6, atol=1e-6, name="adam p")
    close(M, mr, name="adam m")
    close(V, vr, name="adam v")
    close(E, er, rtol=1e-6, atol=1e-6, name="ema")
    # against torch.optim.AdamW + clip_grad_norm_ (losses.py:30-62)
    pt = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([pt], lr=2e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    P2 = p.clone().to(DEV)
    M2, V2 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in (1, 2, 3):
        gs = rnd(n, seed=10 + step) * 2
        pt.grad = gs.clone()
        torch.nn.utils.clip_grad_norm_([pt], 1.0)
        opt.step()
        hip.adamw_ema(P2, gs.to(DEV), M2, V2, None, 2e-4, 0.9, 0.99, 1e-8, 0.01, step, 1.0, hip.sqnorm(gs.to(DEV)),
                      1.0, 0.0)
    close(P2, pt.data, rtol=1e-6, atol=1e-6, name="adamw vs torch")


def test_batchnorm_slogdet_ema():
    x = rnd(4, 8, 8, 48) * 2 + 0.5
    (m, v, r), (mr, vr, rr) = run_both("bn_stats", x, 1e-5)
    close(m, mr, name="bn mean"); close(v, vr, rtol=1e-4, name="bn var"); close(r, rr, rtol=1e-4, name="bn rstd")
    gamma, beta = rnd(48, seed=1) + 1.2, rnd(48, seed=2)
    for act in (0, 2):
        y, yr = run_both("bn_apply", x, mr.float(), rr.float(), gamma, beta, act=act)
        close(y, yr, name="bn apply")
        dy = rnd(4, 8, 8, 48, seed=3)
        outs, refs = run_both("bn_bwd", dy, x, mr.float(), rr.float(), gamma, beta, act=act)
        for o, r_, nm in zip(outs, refs, ("dx", "dgamma", "dbeta")):
            close(o, r_, rtol=1e-4, atol=1e-4, name="bn " + nm)
    for n in (64, 17):
        w = torch.linalg.qr(rnd(n, n))[0] + 0.05 * rnd(n, n, seed=4)
        (ld, inv), (ldr, invr) = run_both("slogdet_inv", w.contiguous())
        close(ld, ldr, rtol=1e-4, atol=1e-4, name="slogdet")
        close(inv, invr, rtol=1e-4, atol=1e-4, name="inverse")
    xa, xb, yy = rnd(3000), rnd(3000) + rnd(3000, seed=2) * 4e-3, rnd(3000, seed=3)
    c, cr = run_both("fixed_point_check", xa, xb, yy, 1e-5, 1e-5)
    assert float(c) == float(cr) and float(cr) > 0
    c, cr = run_both("fixed_point_check", xa, xa.clone(), yy, 1e-5, 1e-5)
    assert float(c) == 0.0
    s_, p_ = rnd(1001), rnd(1001, seed=1)
    S = s_.clone().to(DEV)
    hip.ema_update(S, p_.to(DEV), 0.25)
    sr = s_.double()
    _emu.ema_update(sr, p_.double(), 0.25)
    close(S, sr, name="ema")


UFD_CASES = [  # C, H, W, up, down, pad0, pad1  (the 9 VE call signatures, SURVEY App. B, + crops)
    (128, 32, 32, 1, 2, 1, 1), (3, 32, 32, 1, 1, 2, 2), (256, 16, 16, 1, 2, 1, 1), (128, 16, 16, 1, 1, 2, 2),
    (256, 8, 8, 1, 2, 1, 1), (256, 8, 8, 1, 1, 2, 2), (256, 4, 4, 2, 1, 2, 1), (256, 8, 8, 2, 1, 2, 1),
    (64, 16, 16, 2, 1, 2, 1), (5, 9, 7, 2, 3, -1, 2), (4, 8, 8, 1, 1, -1, -1), (2, 6, 6, 3, 2, 0, -2),
]


@pytest.mark.parametrize("C,H,W,up,down,p0,p1", UFD_CASES)
@pytest.mark.parametrize("layout", [0, 1])
def test_upfirdn2d(C, H, W, up, down, p0, p1, layout):
    x = rnd(2, C, H, W) if layout == 0 else rnd(2, H, W, C)
    k = torch.tensor([1., 3., 3., 1.])
    k = torch.outer(k, k)
    k = k / k.sum() + rnd(4, 4, seed=3) * 0.05  # asymmetric: catches flips / transposes
    o, r = run_both("upfirdn2d", x, k, up, up, down, down, p0, p1, p0, p1, layout)
    close(o, r, name="upfirdn2d")
    k3 = rnd(3, 2, seed=5)
    o, r = run_both("upfirdn2d", x, k3, up, 1, 1, down, p0, p1, p1, p0, layout)
    close(o, r, name="upfirdn2d aniso")


def test_fused_bias_act():
    x, b = rnd(3, 8, 5, 5), rnd(8, seed=1)
    ref = rnd(3, 8, 5, 5, seed=2)
    for act in (1, 3):
        for grad in (0, 1, 2):
            o, r = run_both("fused_bias_act", x, b, ref, act, grad, 0.2, math.sqrt(2))
            close(o, r, name=f"fba {act}{grad}")
    o, r = run_both("fused_bias_act", x, None, ref, 3, 1, 0.1, 1.0)
    close(o, r, name="fba nobias")


def test_empty_inputs_are_no_ops():
    """zero-sized batches go through the C ABI without a launch and without an error (status INDM_OK)"""
    e = torch.empty
    assert hip.gemm(e(0, 64, device=DEV), rnd(64, 32).to(DEV)).shape == (0, 32)
    assert hip.conv2d(e(0, 8, 8, 16, device=DEV), rnd(3, 3, 16, 32).to(DEV)).shape == (0, 8, 8, 32)
    assert hip.conv2d_dgrad(e(0, 8, 8, 32, device=DEV), rnd(3, 3, 16, 32).to(DEV), (8, 8)).shape == (0, 8, 8, 16)
    assert hip.mul(e(0, 4, device=DEV), e(0, 4, device=DEV)).numel() == 0
    assert hip.axpby(e(0, device=DEV), None, 1.0, 0.0).numel() == 0
    a, d = hip.sincos_fwd(e(0, 5, device=DEV), True)
    assert a.numel() == 0 and d.numel() == 0
    assert hip.tap_spread(e(0, 4, 4, 3, device=DEV), 1).shape == (0, 4, 4, 27)


def test_gemm_conv_randomised_shapes():
    """randomised sweep across the dispatch boundaries of the igemm engine (8-wave tiles, big tiles, split-K, skinny
    forms): every case against the fp64 statement"""
    g = torch.Generator().manual_seed(1234)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))

    for case in range(36):
        kind = case % 4
        if kind == 0:      # long-M GEMM around the big-tile thresholds
            M = ri(200, 300) * 256 + ri(0, 255) if case % 8 == 0 else ri(1, 40) * 128 + ri(0, 127)
            N, K = [64, 96, 128, 192, 256, 320, 512][ri(0, 6)], 16 * ri(1, 40) + (ri(0, 15) if case % 3 == 0 else 0)
            ta, tb = False, bool(ri(0, 1))
        elif kind == 1:    # weight-grad GEMM (A stored [K, M]) around the split-K / big-tile thresholds
            M, N, K = [27, 108, 256, 512, 260][ri(0, 4)], [27, 128, 256, 512][ri(0, 3)], 16 * ri(100, 2500)
            ta, tb = True, False
        elif kind == 2:    # small / ragged
            M, N, K, ta, tb = ri(1, 300), ri(1, 300), ri(1, 300), bool(ri(0, 1)), bool(ri(0, 1))
        else:
            M = None
        if M is not None:
            a = rnd(K, M, seed=case) if ta else rnd(M, K, seed=case)
            b = rnd(N, K, seed=case + 1) if tb else rnd(K, N, seed=case + 1)
            gate = rnd(*a.shape, seed=case + 2) if case % 5 == 0 else None
            out, ref = run_both("gemm", a, b, trans_a=ta, trans_b=tb, splitk=ta, gate=gate)
            close(out, ref, rtol=1e-5, atol=2e-5 * math.sqrt(K) + (2e-3 if ta else 0), name=f"gemm case {case} {M}x{N}x{K} {ta}{tb}")
        else:              # convolution: fwd, dgrad, wgrad
            Nimg, H = ri(1, 40), [8, 16, 32][ri(0, 2)]
            Ci, Co = [3, 12, 16, 32, 64, 128][ri(0, 5)], [3, 12, 32, 64, 128, 256][ri(0, 5)]
            k = [1, 3][ri(0, 1)]
            s = [1, 2][ri(0, 1)] if k == 3 else 1
            p = k // 2
            OH = (H + 2 * p - k) // s + 1
            x, w = rnd(Nimg, H, H, Ci, seed=case), rnd(k, k, Ci, Co, seed=case + 1, scale=0.2)
            dy = rnd(Nimg, OH, OH, Co, seed=case + 2)
            out, ref = run_both("conv2d", x, w, stride=s, pad=p)
            close(out, ref, rtol=1e-5, atol=1e-4, name=f"conv case {case}")
            out, ref = run_both("conv2d_dgrad", dy, w, (H, H), stride=s, pad=p)
            close(out, ref, rtol=1e-5, atol=1e-4, name=f"dgrad case {case}")
            out, ref = run_both("conv2d_wgrad", x, dy, k, k, stride=s, pad=p)
            close(out, ref, rtol=1e-5, atol=3e-3, name=f"wgrad case {case}")


@pytest.mark.parametrize("R,C,rps", [(256, 32, 64), (2048, 512, 1024), (1024, 512, 256), (192, 48, None), (100, 7, None)])
@pytest.mark.parametrize("want_dt", [False, True])
def test_ires_bwd_act(R, C, rps, want_dt):
    """one Sin level of the hand-written iResBlock backward: dp, the chain cotangent dt and the per-image column sums"""
    ts = [rnd(R, C, seed=i) for i in range(5)]
    out, ref = run_both("ires_bwd_act", *ts, want_dt=want_dt, rows_per_seg=rps)
    close(out[0], ref[0], name="ires dp")
    if want_dt:
        close(out[1], ref[1], name="ires dt")
    else:
        assert out[1] is None
    if rps:
        close(out[2], ref[2], rtol=2e-5, atol=2e-5 * math.sqrt(rps), name="ires seg")
    else:
        assert out[2] is None


@pytest.mark.parametrize("C,Cw", [(3, 512), (12, 512), (3, 32)])
def test_conv2d_wgrad_x_gate(C, Cw):
    """weight grad of a conv with a wide x and skinny dy where x carries a gate (the VJP chain's dW3)"""
    x, g, dy = rnd(4, 16, 16, Cw, seed=1), rnd(4, 16, 16, Cw, seed=2), rnd(4, 16, 16, C, seed=3)
    out, ref = run_both("conv2d_wgrad", x, dy, 3, 3, x_gate=g)
    close(out, ref, rtol=2e-5, atol=2e-5 * 32, name="wgrad x_gate")


@pytest.mark.parametrize("M,K,J", [(64, 512, 27), (256, 512, 27), (128, 512, 108), (192, 32, 27), (8192, 512, 108)])
def test_ires_chain_back_to_back(M, K, J):
    """((t3 .* d2) W2T .* d1) W1p in one launch (the 512-wide intermediate never leaves the chip) vs fp64"""
    JP = 32 if J <= 32 else 128
    t3, d2, d1 = rnd(M, K, seed=1), rnd(M, K, seed=2), rnd(M, 512, seed=3)
    w2t = rnd(K, 512, seed=4, scale=K ** -0.5)
    w1p = torch.zeros(512, JP)
    w1p[:, :J] = rnd(512, J, seed=5, scale=512 ** -0.5)
    out, ref = run_both("ires_chain", t3, d2, w2t, d1, w1p, J)
    assert out is not None and out.shape == (M, J)
    close(out, ref, rtol=2e-5, atol=2e-5, name="ires_chain")
    (out, t2), (ref, t2ref) = run_both("ires_chain", t3, d2, w2t, d1, w1p, J, want_t2=True)   # variant that also stores t2
    close(out, ref, rtol=2e-5, atol=2e-5, name="ires_chain (t2 variant) P")
    close(t2, t2ref, rtol=2e-5, atol=2e-5, name="ires_chain t2")
    # geometry the kernel does not cover -> None (the caller falls back to two GEMMs)
    assert hip.ires_chain(t3[:M - 1].contiguous().to(DEV), d2[:M - 1].contiguous().to(DEV), w2t.to(DEV),
                          d1[:M - 1].contiguous().to(DEV), w1p.to(DEV), J) is None


@pytest.mark.parametrize("B,T,C", [(3, 256, 256), (2, 64, 256), (2, 128, 128), (5, 256, 128)])
def test_fused_attention(B, T, C):
    """AttnBlockpp core (layerspp.py:88-104): softmax(q k^T / sqrt(C)) v forward and the full backward, logits kept in
    registers, vs the fp64 composition"""
    q, k, v, dh = (rnd(B, T, C, seed=i, scale=1.5) for i in range(4))
    scale = float(C) ** -0.5
    (h, lse), (href, lref) = run_both("attn_fwd", q, k, v, scale)
    close(h, href, rtol=2e-5, atol=2e-6, name="attn h")
    close(lse, lref, rtol=1e-5, atol=2e-5, name="attn lse")
    got = hip.attn_bwd(q.to(DEV), k.to(DEV), v.to(DEV), h, dh.to(DEV), lse, scale)
    ref = _emu.attn_bwd(q.double(), k.double(), v.double(), href, dh.double(), lref, scale)
    for g, r, nm in zip(got, ref, ("dq", "dk", "dv")):
        close(g, r, rtol=3e-5, atol=3e-6, name="attn " + nm)
    # through autograd (ops.attention) with a sharp softmax (large logits)
    from indm_amd import ops
    qs = (q * 6).to(DEV).requires_grad_(True)
    ks, vs = k.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)
    qd, kd, vd = (t.detach().cpu().double().requires_grad_(True) for t in (qs, ks, vs))
    oref = torch.softmax(qd @ kd.transpose(1, 2) * scale, -1) @ vd
    rq, rk, rv = torch.autograd.grad(oref, [qd, kd, vd], dh.double())
    was = ops.FUSED_ATTN_BWD["on"]
    try:
        for fused_bwd in (False, True):   # the default composed backward (recomputed probabilities + 5 GEMMs) and the fused one
            ops.FUSED_ATTN_BWD["on"] = fused_bwd
            out = ops.attention(qs, ks, vs, scale)
            gq, gk, gv = torch.autograd.grad(out, [qs, ks, vs], dh.to(DEV))
            close(out, oref, rtol=2e-5, atol=2e-6, name="sharp attn h")
            for g, r, nm in zip((gq, gk, gv), (rq, rk, rv), ("dq", "dk", "dv")):
                close(g, r, rtol=5e-5, atol=5e-6, name=f"sharp attn {nm} (fused bwd {fused_bwd})")
    finally:
        ops.FUSED_ATTN_BWD["on"] = was


def test_attention_fallback_geometry():
    """geometries the fused kernel does not cover go through the batched GEMM + softmax path with the same result"""
    from indm_amd import ops
    q, k, v = (rnd(2, 16, 32, seed=i) for i in range(3))
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), 32 ** -0.5)
    ref = torch.softmax(q.double() @ k.double().transpose(1, 2) * 32 ** -0.5, -1) @ v.double()
    close(out, ref, name="attention fallback")
    assert hip.attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), 1.0) is None


@pytest.mark.parametrize("M,N,K,tb", [(16384, 512, 512, False), (16384, 512, 512, True), (4096, 256, 1024, True)])
def test_bf16_glds_gemm(bf16_mode, M, N, K, tb):
    """bf16 mode on the direct-to-LDS kernel (long GEMMs, gated and plain, split-K variant included): exact products of
    the bf16-rounded inputs (the gate is multiplied in fp32 and the product rounded), fp32 accumulation"""
    a, g = rnd(M, K), rnd(M, K, seed=5)
    b = rnd(N, K, seed=1) if tb else rnd(K, N, seed=1)
    out = hip.gemm(a.to(DEV), b.to(DEV), trans_b=tb)
    assert "glds" in hip.lib().indm_igemm_last_variant().decode() and "bf16" in hip.lib().indm_igemm_last_variant().decode()
    close(out, _emu.gemm(_r16(a).double(), _r16(b).double(), trans_b=tb), rtol=1e-5, atol=1e-5 * math.sqrt(K), name="bf16 glds")
    out = hip.gemm(a.to(DEV), b.to(DEV), trans_b=tb, gate=g.to(DEV))
    close(out, _emu.gemm(_r16(a * g).double(), _r16(b).double(), trans_b=tb), rtol=1e-5, atol=1e-5 * math.sqrt(K),
          name="bf16 glds gated")


@pytest.mark.parametrize("K", [16, 32, 48, 64])
@pytest.mark.parametrize("tb", [False, True])
def test_bf16_glds_short_k(bf16_mode, K, tb):
    """the three-buffer loop of the bf16 direct-to-LDS kernel at 1 .. 4 slabs (prologue shorter than / equal to / longer than
    its depth; NT plain = three buffers, gated = two), with a ragged last row tile; the NN arm lands on the thin-K kernel"""
    M, N = 57344 + 40, 128
    a, g = rnd(M, K), rnd(M, K, seed=5)
    b = rnd(N, K, seed=1) if tb else rnd(K, N, seed=1)
    out = hip.gemm(a.to(DEV), b.to(DEV), trans_b=tb)
    v = hip.lib().indm_igemm_last_variant().decode()
    assert ("glds" in v and "bf16" in v) if tb else "thin_k" in v, v   # (NN with K <= 128 is the thin-K kernel's, also in bf16 mode)
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M)])
    close(out[rows], _emu.gemm(_r16(a[rows]).double(), _r16(b).double(), trans_b=tb), rtol=1e-5, atol=1e-5 * math.sqrt(K),
          name="bf16 glds short K")
    out = hip.gemm(a.to(DEV), b.to(DEV), trans_b=tb, gate=g.to(DEV))
    close(out[rows], _emu.gemm(_r16(a[rows] * g[rows]).double(), _r16(b).double(), trans_b=tb), rtol=1e-5,
          atol=1e-5 * math.sqrt(K), name="bf16 glds short K gated")


def test_bf16_glds_conv_short_k(bf16_mode):
    """16-channel convolutions on the same loop: 1 x 1 (one slab), 3 x 3 (nine slabs, one tap each), forward and data-grad"""
    for k in (1, 3):
        x, w = rnd(64, 32, 32, 16), rnd(k, k, 16, 128, seed=1, scale=0.2)
        out = hip.conv2d(x.to(DEV), w.to(DEV), pad=k // 2)
        imgs = [0, 63]
        close(out[imgs], _emu.conv2d(_r16(x[imgs]).double(), _r16(w).double(), pad=k // 2), rtol=1e-5, atol=1e-4,
              name=f"bf16 glds conv {k}x{k} 16 ch")
        wd = rnd(k, k, 128, 16, seed=2, scale=0.2)
        dy = rnd(64, 32, 32, 16, seed=4)
        out = hip.conv2d_dgrad(dy.to(DEV), wd.to(DEV), (32, 32), pad=k // 2)
        close(out[imgs], _emu.conv2d_dgrad(_r16(dy[imgs]).double(), _r16(wd).double(), (32, 32), pad=k // 2), rtol=1e-5,
              atol=1e-4, name=f"bf16 glds dgrad {k}x{k} 16 ch")


def test_bf16_glds_conv(bf16_mode):
    x, w = rnd(64, 32, 32, 128), rnd(3, 3, 128, 128, seed=1, scale=0.1)
    out = hip.conv2d(x.to(DEV), w.to(DEV))
    assert "glds" in hip.lib().indm_igemm_last_variant().decode()
    imgs = [0, 31, 63]
    ref = _emu.conv2d(_r16(x[imgs]).double(), _r16(w).double())
    close(out[imgs], ref, rtol=1e-5, atol=2e-4, name="bf16 glds conv")
    dy = rnd(64, 32, 32, 128, seed=4)
    out = hip.conv2d_dgrad(dy.to(DEV), w.to(DEV), (32, 32))
    close(out[imgs], _emu.conv2d_dgrad(_r16(dy[imgs]).double(), _r16(w).double(), (32, 32)), rtol=1e-5, atol=2e-4,
          name="bf16 glds dgrad")


@pytest.mark.parametrize("M,J", [(256, 27), (8192, 108)])
def test_bf16_ires_chain(bf16_mode, M, J):
    JP = 32 if J <= 32 else 128
    t3, d2, d1 = rnd(M, 512, seed=1), rnd(M, 512, seed=2), rnd(M, 512, seed=3)
    w2t = rnd(512, 512, seed=4, scale=512 ** -0.5)
    w1p = torch.zeros(512, JP)
    w1p[:, :J] = rnd(512, J, seed=5, scale=512 ** -0.5)
    out = hip.ires_chain(t3.to(DEV), d2.to(DEV), w2t.to(DEV), d1.to(DEV), w1p.to(DEV), J)
    t2 = (_r16(t3 * d2).double() @ _r16(w2t).double()).float()
    ref = _r16(t2 * d1).double() @ _r16(w1p[:, :J]).double()
    # t2 is rounded to fp32 here and accumulated in a different order on the GPU: its bf16 rounding may flip for
    # values next to a rounding boundary (relative 2^-9 on a single term of a 512-term sum)
    close(out, ref, rtol=2e-4, atol=2e-4, name="bf16 ires_chain")


@pytest.mark.parametrize("M,K,J", [(256, 512, 27), (8192, 512, 108), (131072, 512, 27), (64, 32, 5), (192, 48, 128)])
def test_bf16_ires_chain_packed_weights(bf16_mode, M, K, J):
    """W2T pre-packed into bf16 fragment order and read straight into registers (three rotating A/gate buffers): the
    same rounding points as the LDS-image kernel, so the two agree to accumulation order; also vs exact products"""
    JP = 32 if J <= 32 else 128
    t3, d2, d1 = rnd(M, K, seed=1), rnd(M, K, seed=2), rnd(M, 512, seed=3)
    w2t = rnd(K, 512, seed=4, scale=K ** -0.5).to(DEV)
    w1p = torch.zeros(512, JP)
    w1p[:, :J] = rnd(512, J, seed=5, scale=512 ** -0.5)
    args = (t3.to(DEV), d2.to(DEV), w2t, d1.to(DEV), w1p.to(DEV), J)
    plain, t2_plain = hip.ires_chain(*args, want_t2=True)
    hip.pack_w2t_bf16([w2t])
    assert id(w2t) in hip._w2t_packed
    # the packed layout itself
    pk = hip._w2t_packed[id(w2t)][2].cpu().float()
    s_, b_, lane, e = 1 % (K // 16), 7, 45, 6
    assert pk[s_, b_, lane >> 5, lane & 31, e >> 2, e & 3] == _r16(w2t.cpu())[16 * s_ + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3), 32 * b_ + (lane & 31)]
    out, t2 = hip.ires_chain(*args, want_t2=True)
    close(t2, t2_plain.cpu().double(), rtol=1e-5, atol=1e-5, name="packed-weight chain t2 vs LDS-image kernel")
    close(out, plain.cpu().double(), rtol=2e-4, atol=2e-4, name="packed-weight chain P vs LDS-image kernel")
    ref_t2 = _r16(t3 * d2).double() @ _r16(w2t.cpu()).double()
    close(t2, ref_t2, rtol=1e-5, atol=1e-4, name="packed-weight chain t2 vs exact products")
    out1 = hip.ires_chain(*args)
    close(out1, out.cpu().double(), rtol=1e-6, atol=1e-6, name="packed-weight chain without the t2 store")
    w2t.add_(1.0)   # a modified weight must not hit the stale packed copy
    out2 = hip.ires_chain(*args)
    assert not torch.allclose(out2, out)
    del hip._w2t_packed[id(w2t)]


@pytest.mark.parametrize("C", [3, 12])
def test_bf16_thin_kernels(bf16_mode, C):
    """bf16 mode on the HBM-bound thin kernels (skinny 3x3 convs of the flow, their transposes and weight grads):
    operands rounded to bf16 at the fragment read, exact fp32 MFMA == products of the rounded inputs"""
    N, H, W, Cw = 8, 32, 32, 512
    x, w = rnd(N, H, W, C), rnd(3, 3, C, Cw, seed=1, scale=0.3)
    b = rnd(Cw, seed=2)
    out = hip.conv2d(x.to(DEV), w.to(DEV), bias=b.to(DEV))
    assert "thin" in hip.lib().indm_igemm_last_variant().decode()
    close(out, _emu.conv2d(_r16(x).double(), _r16(w).double(), bias=b.double()), rtol=1e-5, atol=1e-4, name="bf16 thin conv")
    v = rnd(N, H, W, C, seed=3)
    w3 = rnd(3, 3, Cw, C, seed=4, scale=0.05)
    out = hip.conv2d_dgrad(v.to(DEV), w3.to(DEV), (H, W))                     # skinny dy -> wide (thin conv, sign +1)
    close(out, _emu.conv2d_dgrad(_r16(v).double(), _r16(w3).double(), (H, W)), rtol=1e-5, atol=1e-4, name="bf16 thin conv_t")
    wide, g = rnd(N, H, W, Cw, seed=5), rnd(N, H, W, Cw, seed=6)
    out = hip.conv2d_wgrad(x.to(DEV), wide.to(DEV), 3, 3, dy_gate=g.to(DEV))  # thin weight grad, gated wide operand
    ref = _emu.conv2d_wgrad(_r16(x).double(), _r16(wide * g).double(), 3, 3)
    close(out, ref, rtol=1e-5, atol=2e-3, name="bf16 thin wgrad")


@pytest.mark.parametrize("split,up,inverse", [(0, 1, 0), (0, 0, 0), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 0, 0)])
def test_prior_flow_coupling(split, up, inverse):
    """affine coupling of the 64-d prior flow (coupling.py:13-177, transform.py:49-81): forward / inverse, both split
    types, value + log-det + gradients"""
    B, D = 37, 64
    x, prm = rnd(B, D, seed=1, scale=2.0), rnd(B, D, seed=2, scale=3.0)
    (y, ld), (yr, ldr) = run_both("coupling_fwd", x, prm, split, up, inverse)
    close(y, yr, name="coupling y")
    close(ld, ldr, rtol=2e-5, atol=2e-5, name="coupling logdet")
    dy, dld = rnd(B, D, seed=3), rnd(B, seed=4)
    (dx, dp), (dxr, dpr) = run_both("coupling_bwd", dy, dld, x, prm, split, up, inverse)
    close(dx, dxr, name="coupling dx")
    close(dp, dpr, rtol=3e-5, atol=3e-5, name="coupling dparams")


@pytest.mark.parametrize("R,C", [(128, 512), (2, 12), (1024, 3), (1, 256), (1025, 64)])
def test_colreduce_short(R, C):
    """R <= 1024, mode 0: the single-launch form (per-image bias grads -> bias grad); R = 1025 takes the two-stage one"""
    x = rnd(R, C)
    o, r = run_both("colreduce", x, None, mode=0, alpha=0.5)
    close(o, r, rtol=1e-5, atol=1e-5 * math.sqrt(R), name="colreduce short")


@pytest.mark.parametrize("R,sizes", [(128, [128] * 10 + [256] * 34), (2, [128, 256, 4]), (7, [5]), (3, [1, 2, 3, 4, 5])])
def test_split_cols(R, sizes):
    x = rnd(R, sum(sizes))
    flat, ref = run_both("split_cols", x, sizes)
    assert torch.equal(flat.cpu().double(), ref)
    back = hip.split_cols(flat, sizes, inverse_of=R)
    assert torch.equal(back.cpu(), x)
    # the autograd wrapper hands out dense per-block views and sends their gradients back as one matrix
    from indm_amd import ops
    xg = x.to(DEV).requires_grad_(True)
    parts = ops.split_cols(xg, sizes)
    o = 0
    for p, c in zip(parts, sizes):
        assert p.is_contiguous() and torch.equal(p.detach().cpu(), x[:, o:o + c])
        o += c
    sum((i + 1.0) * p.sum() for i, p in enumerate(parts) if i != 1 or len(sizes) == 1).backward()
    want = torch.cat([torch.full((R, c), (i + 1.0) if (i != 1 or len(sizes) == 1) else 0.0) for i, c in enumerate(sizes)], dim=1)
    assert torch.equal(xg.grad.cpu(), want)


@pytest.mark.parametrize("O,I", [(128, 256), (64, 64), (5, 7), (300, 1000)])
def test_weightnorm(O, I):
    v, g, dw = rnd(O, I), rnd(O, 1, seed=1) + 2.0, rnd(O, I, seed=2)
    (w, n), (rw, rn) = run_both("weightnorm_fwd", v, g)
    close(w, rw, rtol=1e-5, atol=1e-6, name="weightnorm w")
    close(n, rn, rtol=1e-5, atol=1e-6, name="weightnorm norm")
    (dv, dg), (rdv, rdg) = run_both("weightnorm_bwd", dw, v, g, rn.float())
    close(dv, rdv, rtol=1e-4, atol=1e-5, name="weightnorm dv")
    close(dg, rdg, rtol=1e-4, atol=1e-5, name="weightnorm dg")
    # against autograd of the reference expression (nnet/weight_norm.py: nn.utils.weight_norm)
    vd, gd = v.double().requires_grad_(True), g.double().requires_grad_(True)
    (vd * (gd / vd.norm(dim=1, keepdim=True)) * dw.double()).sum().backward()
    close(dv, vd.grad, rtol=1e-4, atol=1e-5, name="weightnorm dv vs autograd")
    close(dg, gd.grad.reshape(-1), rtol=1e-4, atol=1e-5, name="weightnorm dg vs autograd")


def test_dequantize_and_scale():
    """run_lib.py:86-87 ((255 x + u)/256 then the data scaler) as one launch (indm_amd/datasets.py)"""
    from indm_amd import configs, datasets
    from indm_amd.rng import DeviceRng
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    x = torch.round(rnd(8, 3, 32, 32).abs().clamp(0, 1) * 255) / 255
    for centered in (True, False):
        cfg.data.centered = centered
        out = datasets.dequantize_and_scale(cfg, x.to(DEV), DeviceRng(torch.device(DEV), seed=5))
        u = DeviceRng(torch.device(DEV), seed=5).rand(*x.shape).cpu()
        want = (255.0 * x + u) / 256.0
        want = want * 2.0 - 1.0 if centered else want
        close(out, want.double(), rtol=1e-6, atol=1e-6, name="dequantize")
