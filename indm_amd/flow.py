"""The INDM flow (wolf VAE shell around a conditional Residual Flow) on the HIP kernels.

Mirror of flow_models/flow_model.py (flow_forward / create_flow_model), flow_models/wolf/wolf.py
(WolfCore.forward :81-130) and flow_models/wolf/flows/resflow/** with the reference's state-dict
names.  MI355X-first differences in HOW it is computed (results identical within fp32 tolerance):

  * log-det engine: the reference differentiates the g-net with autograd (n+3 VJPs per block,
    one of them create_graph=True -> double backward).  Here the VJP  v -> v^T dg/dx  is an explicit
    forward chain of transposed MFMA convolutions / GEMMs and cos(2 pi p) gates saved by the fused
    Sin kernel, so the Neumann series is a plain loop and the estimator's gradient is ordinary
    first-order autograd through that chain (DESIGN.md "log-det engine").
  * Lop weight normalisation is evaluated ONCE per flow call and shared by g and all VJPs
    (the reference recomputes it in every conv call, lipschitz.py:350-363).
  * the conditioning input (a1 + Linear(h)) W2 + b2 is folded into the GEMM epilogue as a per-image
    bias  (Linear(h) W2 + b2).
  * slogdet of the 64x64 prior matrix runs on the GPU (one-wavefront Gauss-Jordan) instead of a
    host round trip (permutation.py:107-108).
Tensors are NHWC inside; the NCHW boundary is converted once in flow_forward.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import hip, ops


def poisson_1mcdf(lamb, k, offset):
    """iresblock.py:309-318"""
    if k <= offset:
        return 1.0
    k = k - offset
    s = 1.0
    for i in range(1, k):
        s += lamb ** i / math.factorial(i)
    return 1 - np.exp(-lamb) * s


# ----------------------------------------------------------------------------- Lop convolutions
class LopConv(nn.Module):
    """LopConv2d / LopCondConv2d (layers/base/lipschitz.py:321-441); weight RSCK (3x3) or [Cin,Cout] (1x1)."""

    def __init__(self, in_ch, out_ch, k, cond=False, coeff=0.98):
        super().__init__()
        ref = torch.nn.Conv2d(in_ch, out_ch, k, 1, k // 2)  # nn.Conv2d default init (kaiming-uniform)
        w = ref.weight.detach()
        self.k, self.coeff, self.in_ch, self.out_ch = k, coeff, in_ch, out_ch
        self.weight = nn.Parameter(w.permute(2, 3, 1, 0).contiguous() if k == 3 else w[:, :, 0, 0].t().contiguous())
        self.bias = nn.Parameter(ref.bias.detach().clone())
        self.register_buffer("scale", torch.tensor(0.0))
        if cond:
            h_net = nn.Module()
            net = nn.Module()
            lin = torch.nn.Linear(64, in_ch)
            net.weight = nn.Parameter(lin.weight.detach().clone())
            net.bias = nn.Parameter(lin.bias.detach().clone())
            h_net.net = net
            self.h_net = h_net

    last_l1 = None  # l1 norms of the latest batched normalisation (ResidualFlow.prepare)

    def normalized(self):
        wn, l1 = ops.lop_normalize(self.weight, self.coeff)
        with torch.no_grad():
            self.scale.copy_(l1.max())  # lipschitz.py:353-354
        self.last_l1 = None
        return wn

    def sync_scale(self):
        """write the `scale` buffer the reference updates on every call (lipschitz.py:353-354) from the last norms"""
        if self.last_l1 is not None:
            with torch.no_grad():
                self.scale.copy_(self.last_l1.max())
            self.last_l1 = None


class Placeholder(nn.Module):
    """Stands in for the parameter-free Sin modules so nn.Sequential indices match the reference."""


class IResBlock(nn.Module):
    """iResBlock (layers/iresblock.py:14-174) with nnet = [Sin?] Lop3x3 Sin LopCond1x1 Sin Lop3x3."""

    def __init__(self, ch, idim, preact):
        super().__init__()
        self.geom_p = nn.Parameter(torch.tensor(np.log(0.5) - np.log(1.0 - 0.5), dtype=torch.float32))
        self.lamb = nn.Parameter(torch.tensor(2.0))
        self.register_buffer("last_n_samples", torch.zeros(1))
        self.register_buffer("last_firmom", torch.zeros(1))
        self.register_buffer("last_secmom", torch.zeros(1))
        mods = ([Placeholder()] if preact else []) + [LopConv(ch, idim, 3), Placeholder(), LopConv(idim, idim, 1, True),
                                                     Placeholder(), LopConv(idim, ch, 3)]
        self.nnet = nn.Sequential(*mods)
        self.preact = preact
        self.convs = [m for m in mods if isinstance(m, LopConv)]
        self.lamb_value = 2.0  # lamb never receives a gradient (iresblock.py:110 uses .item())

    def weights(self):
        return [c.normalized() for c in self.convs]

    def g(self, x, cb, W, want_gates):
        """g(x,h) and (optionally) the cos gates d0,d1,d2 for the VJP chain.  cb = Linear_h(h) @ W2 is the folded
        conditioning bias [B, idim] (computed for all blocks at once by ResidualFlow.prepare)."""
        c1, c2, c3 = self.convs
        W1, W2, W3 = W[:3]
        d0 = None
        a0 = x
        if self.preact:
            a0, d0 = ops.sincos(x)
        p1 = ops.conv2d(a0, W1, c1.bias)
        a1, d1 = ops.sincos(p1)
        p2 = ops.conv1x1(a1, W2, c2.bias, cb)
        a2, d2 = ops.sincos(p2)
        g = ops.conv2d(a2, W3, c3.bias)
        return g, ((d0, d1, d2) if want_gates else None)

    def vjp(self, v, W, gates, hw):
        """v^T dg/dx as an explicit chain (replaces torch.autograd.grad(g, x, v), iresblock.py:257,269-271)."""
        W1, W2, W3 = W[:3]
        d0, d1, d2 = gates
        t = ops.conv2d_t(v, W3, hw)
        # (t .* d2) @ W2^T with the cos gate on the GEMM's A loader; W[3] = W2^T stored row-major (prepare()): the gated
        # direct-to-LDS kernel runs 8 % faster on a [K, N] weight (589 vs 637 us at [131072, 512] x [512, 512])
        Nn, Hh, Ww, Cc = t.shape
        t = ops.gemm(t.reshape(-1, Cc), W[3], weight_b=True, gate=d2.reshape(-1, Cc)).reshape(Nn, Hh, Ww, -1)
        t = ops.conv2d_t(t, W1, hw, gate=d1)    # (t .* d1) through the transposed 3x3
        if d0 is not None:
            t = ops.mul(t, d0)
        return t

    def vjp_acc(self, v, W, gates, hw, acc, coeff):
        """One series iteration without a graph: (v^T dg/dx, acc + coeff * v^T dg/dx); acc may be None.  Same
        contractions as vjp().  On the skinny (3-/12-channel) production shapes the two wide links
        ((t3 .* d2) W2^T .* d1) W1^T run as ONE back-to-back launch that keeps the 512-wide intermediate on chip
        (indm_ires_chain_f32; W[4] = W1 as a zero-padded [512, 32|128] panel from prepare()), and the skinny tail
        (9-tap gather, input gate, accumulation into the Neumann sum) is one launch instead of three."""
        W1, W2, W3, W2T = (w if w.is_contiguous() else w.contiguous() for w in W[:4])
        d0, d1, d2 = gates
        N, H, Wd, _ = v.shape
        t = hip.conv2d_dgrad(v, W3, hw)
        C = t.shape[-1]
        Ci = W1.shape[2]
        if not hip._skinny(3, 3, 1, 1, Ci, W1.shape[3]):
            t = hip.gemm(t.view(-1, C), W2T, gate=d2.view(-1, C))
            t = hip.conv2d_dgrad(t.view(N, H, Wd, -1), W1, hw, gate=d1)
            if d0 is not None:
                t = hip.mul(t, d0)
            return t, (hip.axpby(acc, t, 1.0, coeff) if acc is not None else None)
        P = hip.ires_chain(t.view(-1, C), d2.view(-1, C), W2T, d1.view(-1, d1.shape[-1]), W[4], 9 * Ci) \
            if len(W) > 4 and W[4] is not None else None
        if P is None:
            t = hip.gemm(t.view(-1, C), W2T, gate=d2.view(-1, C))
            P = hip.gemm(t, W1.reshape(9 * Ci, W1.shape[3]), trans_b=True, gate=d1.view(-1, d1.shape[-1]))
        return hip.tap_gather_acc(P.view(N, H, Wd, 9 * Ci), -1, d0, acc, coeff)

    def forward_logdet(self, x, cb, W, rng, stats):
        """_logdetgrad iresblock.py:90-174 (poisson, n_samples=1, neumann_grad in training)."""
        lamb = self.lamb_value
        n = rng.poisson(lamb)
        training = self.training
        n_exact = 2 if training else 20
        K = n + n_exact

        def coeff(k):
            return 1.0 / poisson_1mcdf(lamb, k, n_exact) * (1.0 if n >= k - n_exact else 0.0)

        N, H, Wd, C = x.shape
        eps = rng.randn_image(N, C, H, Wd)
        hw = (H, Wd)
        if training:  # neumann_logdet_estimator :264-273 -- forward AND backward hand-written (_IResTrain)
            c1, c2, c3 = self.convs
            y, logdet = _IResTrain.apply(x, cb, W[0], W[1], W[2], W[3], c1.bias, c2.bias, c3.bias, eps, self,
                                         [(-1) ** k * coeff(k) for k in range(1, K + 1)], W[4] if len(W) > 4 else None)
            if stats is not None:
                stats.append((self, n, logdet.detach()))
            return y, logdet
        if (hip.block_drivers() and x.is_cuda and not torch.is_grad_enabled() and len(W) > 4 and W[4] is not None):
            # the whole block as ONE C-ABI call (indm_iresblock_fwd_logdet, eval mode); same kernels as below
            c1, c2, c3 = self.convs
            out = hip.iresblock_fwd(_cc(x), _cc(cb), _cc(W[0]), _cc(W[3]), W[2], W[4], c1.bias, c2.bias, c3.bias, eps,
                                    [(-1) ** (k + 1) / k * coeff(k) for k in range(1, K + 1)], self.preact, False,
                                    W2=_cc(W[1]))
            if out is not None:
                return out[0], out[1]
        g, gates = self.g(x, cb, W, True)
        with torch.no_grad():  # basic_logdet_estimator :253-261 (create_graph=False)
            vjp = eps
            logdet = None
            for k in range(1, K + 1):
                vjp, _ = self.vjp_acc(vjp, W, gates, hw, None, 0.0)
                tr = hip.rowreduce(vjp, eps, 0, (-1) ** (k + 1) / k * coeff(k))
                logdet = tr if logdet is None else logdet + tr
        return ops.axpby(x, g), logdet

    def inverse(self, y, cb, W, atol=1e-5, rtol=1e-5, iters=None):
        """_inverse_fixed_point iresblock.py:78-88"""
        if hip.block_drivers() and y.is_cuda and not torch.is_grad_enabled():
            c1, c2, c3 = self.convs
            out = hip.iresblock_inverse(_cc(y), _cc(cb), _cc(W[0]), _cc(W[3]), W[2], c1.bias, c2.bias, c3.bias,
                                        self.preact, atol, rtol)
            if out is not None:
                if iters is not None:
                    iters.append(out[1])
                return out[0]
        g, _ = self.g(y, cb, W, False)
        x, x_prev = ops.axpby(y, g, 1.0, -1.0), y
        i = 0
        while float(hip.fixed_point_check(x, x_prev, y, atol, rtol)) != 0.0:
            g, _ = self.g(x, cb, W, False)
            x, x_prev = ops.axpby(y, g, 1.0, -1.0), x
            i += 1
            if i > 1000:
                break
        if iters is not None:
            iters.append(i)
        return x



def _cc(t):
    return t if t.is_contiguous() else t.contiguous()


class _IResTrain(torch.autograd.Function):
    """Training-mode iResBlock: y = x + g(x, h) and the Neumann surrogate of log|det(I + dg/dx)|
    (iresblock.py:141-142, 264-273) with a HAND-WRITTEN backward.

    The reference obtains the estimator's gradient by differentiating torch.autograd.grad(..., create_graph=True) a
    second time.  Here the last vector-Jacobian product is an explicit chain
        vj = d0 .* C1^T( d1 .* ( (d2 .* C3^T w) W2^T ) ),     logdet = <vj, eps>,
    (C^T = transposed 3x3 convolution, d_i = cos(2 pi p_i) the Sin derivatives, w = eps + sum (-1)^k c_k v_k) and its
    reverse-mode sweep is written out below together with g's own backward.  Compared with autograd over the op chain
    (round 1) each Sin level costs ONE fused pass (indm_ires_bwd_act: 6-7 tensor passes instead of mul2 +
    sincos_act_bwd = 10, bias-gradient column sums included), the cotangents of the gates (dd = du .* t) and the gated
    copies (du .* d) are never written -- they ride on the operand loaders of the contractions that consume them --
    and no accumulate kernels run for tensors used twice.  Same values and gradients (golden-pinned through
    parity_cases.case_flow_forward_modes / case_two_training_steps)."""

    @staticmethod
    def forward(ctx, x, cb, W1, W2, W3, W2T, b1, b2, b3, eps, blk, coeffs, panel=None):
        x, cb, W1, W2, W3, W2T, eps = (t if t.is_contiguous() else t.contiguous() for t in (x, cb, W1, W2, W3, W2T, eps))
        N, H, Wd, C = x.shape
        hw = (H, Wd)
        Cw = W2.shape[0]
        ctx.driver = False
        if hip.block_drivers() and x.is_cuda and panel is not None:
            # the whole block as ONE C-ABI call (indm_iresblock_fwd_logdet): the launch sequence below, enqueued from C++
            out = hip.iresblock_fwd(x, cb, W1, W2T, W3, panel, b1, b2, b3, eps, coeffs, blk.preact, True, W2=W2)
            if out is not None:
                y, logdet, (a0, d0, a1, d1, a2, d2, t3, t2, c1, w) = out
                ctx.save_for_backward(a0, d0, a1, d1, a2, d2, t3, t2, c1, w, eps, W1, W2, W3, W2T)
                ctx.cfg = (blk.preact, hw)
                ctx.driver = True
                ctx.panel = panel
                return y, logdet
        a0, d0 = hip.sincos_fwd(x, True) if blk.preact else (x, None)
        p1 = hip.conv2d(a0, W1, bias=b1)
        a1, d1 = hip.sincos_fwd(p1, True)
        del p1
        # a1 W2 through the transposed copy ([N, K] weight form: the ungated direct-to-LDS kernel runs 125 TF on it
        # against 109 TF on the [K, N] form; the gated one prefers [K, N])
        p2 = hip.gemm(a1.view(-1, Cw), W2T, trans_b=True, bias=b2, bias2=cb, rows_per_group=H * Wd)
        a2, d2 = hip.sincos_fwd(p2.view(N, H, Wd, -1), True)
        del p2
        g = hip.conv2d(a2, W3, bias=b3)
        y = hip.axpby(x, g, 1.0, 1.0)
        Ws, gates = (W1, W2, W3, W2T, panel), (d0, d1, d2)
        v, w = eps, eps
        for c in coeffs:                                   # v_k = J^T v_{k-1};  w = eps + sum (-1)^k c_k v_k
            v, w = blk.vjp_acc(v, Ws, gates, hw, w, c)
        t3 = hip.conv2d_dgrad(w, W3, hw)
        Ci = W1.shape[2]
        pt = None
        if panel is not None and hip._skinny(3, 3, 1, 1, Ci, W1.shape[3]):
            # the gradient-carrying chain on the back-to-back kernel too: it stores t2 (its backward needs it) from the
            # accumulators and still skips the gated re-read of t2 and d1 by a separate thin-N launch
            pt = hip.ires_chain(t3.view(-1, Cw), d2.view(-1, Cw), W2T, d1.view(-1, Cw), panel, 9 * Ci, want_t2=True)
        if pt is not None:
            t2 = pt[1].view(N, H, Wd, -1)
            c1 = hip.tap_gather(pt[0].view(N, H, Wd, 9 * Ci), None, -1)
        else:
            t2 = hip.gemm(t3.view(-1, Cw), W2T, gate=d2.view(-1, Cw)).view(N, H, Wd, -1)
            c1 = hip.conv2d_dgrad(t2, W1, hw, gate=d1)
        vj = hip.mul(c1, d0) if blk.preact else c1
        logdet = hip.rowreduce(vj, eps, 0)
        ctx.save_for_backward(a0, d0, a1, d1, a2, d2, t3, t2, c1 if blk.preact else None, w, eps, W1, W2, W3, W2T)
        ctx.cfg = (blk.preact, hw)
        return y, logdet

    @staticmethod
    def backward(ctx, gy, gld):
        a0, d0, a1, d1, a2, d2, t3, t2, c1, w, eps, W1, W2, W3, W2T = ctx.saved_tensors
        preact, hw = ctx.cfg
        N, H, Wd, C = eps.shape
        f16 = a1.dtype == torch.bfloat16   # bf16 storage: F16 buffers from the forward driver
        gy = torch.zeros_like(eps) if gy is None else (gy if gy.is_contiguous() else gy.contiguous())
        if f16 and ctx.driver and gld is not None and hip.block_drivers():
            out = hip.iresblock_bwd(gy, _cc(gld), (a0, d0, a1, d1, a2, d2, t3, t2, c1, w), eps, W1, W2, W2T, W3, preact,
                                    panel=ctx.panel)
            if out is not None:
                dx, dcb, dW1, dW2, dW3, dW2T, db1, db2, db3 = out
                return dx, dcb, dW1, dW2, dW3, dW2T, db1, db2, db3, None, None, None, None
        if f16:   # a backward the F16 driver does not cover (no log-det cotangent): unpack and use the fp32-tensor path
            Cw_ = W2.shape[0]
            a1, d1, a2, d2, t3, t2 = (hip.f16_unpack(t, N * H * Wd, Cw_).view(N, H, Wd, Cw_) for t in (a1, d1, a2, d2, t3, t2))
        Cw = a1.shape[-1]
        rps = H * Wd
        seg_ok = rps % 64 == 0
        f2 = lambda t: t.view(-1, Cw)  # noqa: E731
        gy = torch.zeros_like(eps) if gy is None else (gy if gy.is_contiguous() else gy.contiguous())
        if ctx.driver and gld is not None and hip.block_drivers():
            out = hip.iresblock_bwd(gy, _cc(gld), (a0, d0, a1, d1, a2, d2, t3, t2, c1, w), eps, W1, W2, W2T, W3, preact)
            if out is not None:
                dx, dcb, dW1, dW2, dW3, dW2T, db1, db2, db3 = out
                return dx, dcb, dW1, dW2, dW3, dW2T, db1, db2, db3, None, None, None, None
        # ---- g path, level 3 ----
        da2 = hip.conv2d_dgrad(gy, W3, hw)
        dW3 = hip.conv2d_wgrad(a2, gy, 3, 3)
        db3 = hip.colreduce(gy, None, 0)
        dd0 = dW2T = None
        if gld is not None:
            # ---- reverse sweep of the last VJP chain ----
            dvj = hip.scale_rows(eps, gld if gld.is_contiguous() else gld.contiguous())
            if preact:
                q0, dd0 = hip.mul2(dvj, d0, c1)
            else:
                q0 = dvj
            q1 = hip.conv2d(q0, W1)                                              # dL/d(t2 .* d1)
            dW1 = hip.conv2d_wgrad(q0, t2, 3, 3, dy_gate=d1)
            du2 = hip.gemm(f2(q1), W2, gate=f2(d1))                              # (q1 .* d1) W2 = dL/d(t3 .* d2)
            dW3 = hip.axpby(dW3, hip.conv2d_wgrad(du2.view(N, H, Wd, -1), w, 3, 3, x_gate=d2), 1.0, 1.0)
            dp2, _, dcb = hip.ires_bwd_act(f2(da2), du2, f2(d2), f2(a2), f2(t3), False, rps if seg_ok else None)
            del du2, da2
        else:
            dp2, dcb = hip.sincos_bwd(da2, None, a2, d2), None
            q1 = dW1 = None
        if dcb is None:
            dcb = hip.segreduce(dp2, rps)
        db2 = hip.colreduce(dcb, None, 0)
        # ---- level 2: p2 = a1 W2 + b2 + cb[image] ----
        da1 = hip.gemm(f2(dp2), W2, trans_b=True)
        dW2 = hip.gemm(f2(a1), f2(dp2), trans_a=True, splitk=True)
        del dp2
        if gld is not None:
            dp1, dt2, seg1 = hip.ires_bwd_act(da1, f2(q1), f2(d1), f2(a1), f2(t2), True, rps if seg_ok else None)
            del da1, q1
            dW2T = hip.gemm(f2(t3), dt2, trans_a=True, splitk=True, gate=f2(d2))   # (t3 .* d2)^T (q1 .* d1)
            del dt2
        else:
            dp1, seg1 = hip.sincos_bwd(da1, None, f2(a1), f2(d1)), None
            dW2T = torch.zeros_like(W2T)
        db1 = hip.colreduce(seg1 if seg1 is not None else dp1, None, 0)
        # ---- level 1: p1 = conv(a0, W1) + b1 ----
        dp1 = dp1.view(N, H, Wd, -1)
        da0 = hip.conv2d_dgrad(dp1, W1, hw)
        dW1g = hip.conv2d_wgrad(a0, dp1, 3, 3)
        dW1 = dW1g if dW1 is None else hip.axpby(dW1, dW1g, 1.0, 1.0)
        if preact:
            dx = hip.axpby(gy, hip.sincos_bwd(da0, dd0, a0, d0), 1.0, 1.0)
        else:
            dx = hip.axpby(gy, da0, 1.0, 1.0)
        return dx, dcb, dW1, dW2, dW3, dW2T, db1, db2, db3, None, None, None, None


class SqueezeLayer(nn.Module):
    pass


class StackediResBlocks(nn.Module):
    def __init__(self, ch, idim, n_blocks, squeeze, first_resblock):
        super().__init__()
        chain = [IResBlock(ch, idim, preact=not (first_resblock and i == 0)) for i in range(n_blocks)]
        if squeeze:
            chain.append(SqueezeLayer())
        self.chain = nn.ModuleList(chain)


class ResidualFlow(nn.Module):
    """flows/resflow/resflow_.py:20-335 with the options every shipped config uses."""

    def __init__(self, in_ch, n_blocks, idim):
        super().__init__()
        tr = []
        c = in_ch
        for i, nb in enumerate(n_blocks):
            tr.append(StackediResBlocks(c, idim, nb, squeeze=i < len(n_blocks) - 1, first_resblock=i == 0))
            c *= 4
        self.transforms = nn.ModuleList(tr)

    def blocks(self):
        return [[b for b in t.chain if isinstance(b, IResBlock)] for t in self.transforms]

    def prepare(self, h):
        """Per-call constants of every block, computed up front in a few large launches:
          * Lop-normalised weights (once per flow call, shared by g and every VJP),
          * the folded conditioning bias  cb_b = Linear_b(h) @ W2_b  for ALL blocks: one [B,64]x[64, NB*idim]
            GEMM + one batched [NB] x ([B,idim] x [idim,idim]) GEMM instead of 2*NB short-M launches.
        Returns {block: (W, cb)}."""
        blks = [b for sc in self.blocks() for b in sc]
        convs = [c for b in blks for c in b.convs]
        wns, l1s = ops.lop_normalize_multi([c.weight for c in convs], convs[0].coeff)  # all weights: two launches
        for c, l1 in zip(convs, l1s):
            c.last_l1 = l1          # `scale` (= max l1, lipschitz.py:353-354) is materialised on demand: sync_scales()
        Ws = [wns[3 * i:3 * i + 3] for i in range(len(blks))]
        nb, B = len(blks), h.shape[0]
        hw = torch.cat([b.convs[1].h_net.net.weight for b in blks], dim=0)      # [NB*idim, 64]
        hb = torch.cat([b.convs[1].h_net.net.bias for b in blks], dim=0)
        idim = blks[0].convs[1].in_ch
        c = ops.linear(h, hw, hb).reshape(B, nb, idim).permute(1, 0, 2).contiguous()  # [NB, B, idim]
        w2 = torch.stack([W[1] for W in Ws], dim=0)                                    # [NB, idim, idim]
        cb = ops.gemm(c, w2)                                                           # [NB, B, idim]
        cbs = cb.unbind(0)  # backward = one stack
        w2ts = w2.transpose(1, 2).contiguous().unbind(0)   # W2^T of every block, one launch (used by the VJP chains)
        if hip.compute_dtype() == "bf16":
            hip.pack_w2t_bf16(list(w2ts))   # bf16 fragment-order copies for the chain kernel (two launches)
            if hip.storage_dtype() == "bf16":
                hip.pack_w2t_bf16([W[1] for W in Ws])   # ... and of W2 for the F16 kernels (csrc/flow_f16.hip)
        # W1 of every block as the [idim, 32 | 128] zero-padded panel of the back-to-back chain kernel (no-grad series
        # iterations only): one stack + transpose + pad per scale
        panels = {}
        with torch.no_grad():
            for sc in self.blocks():
                if not sc:
                    continue
                w1 = torch.stack([Ws[blks.index(b)][0].detach() for b in sc], dim=0)        # [nb, 3, 3, C, idim]
                nb_, _, _, C_, I_ = w1.shape
                if hip._skinny(3, 3, 1, 1, C_, I_):
                    hip.prepack_weights([Ws[blks.index(b)][2] for b in sc])   # W3's two GEMM layouts, whole scale
                if hip._skinny(3, 3, 1, 1, C_, I_) and 9 * C_ <= 128:
                    pn = torch.nn.functional.pad(w1.reshape(nb_, 9 * C_, I_).transpose(1, 2),
                                                 (0, (32 if 9 * C_ <= 32 else 128) - 9 * C_)).contiguous()
                    qs = pn.unbind(0)
                    for b, q in zip(sc, qs):
                        panels[b] = q
                    if hip.storage_dtype() == "bf16":   # the thin-N panels of the F16 kernels: W1^T and W3 as [idim, 9C]
                        hip.pack_w2t_bf16(list(qs))
                        hip.pack_w2t_bf16([hip._packed_weight(Ws[blks.index(b)][2], "ci_tapco") for b in sc])
                        if hip.CHAIN_CONV["on"]:   # W3^T for the one-launch series iteration (indm_f16_chain_conv)
                            hip.pack_kpad_bf16([Ws[blks.index(b)][2] for b in sc], "w3t_thin")
                            # ... and W1 for the one-launch g-net (indm_f16_gnet) and the backward's conv + gate + GEMM launch
                            hip.pack_kpad_bf16([Ws[blks.index(b)][0] for b in sc], "w1_thin")
                elif hip.storage_dtype() == "bf16" and hip.f16_wide(C_) and I_ == 512:
                    # the 48-channel scale of the 64 x 64 flows on the F16 drivers (round 6): panels [idim, 9C -> 448] for the chain /
                    # thin-N stages, W1 and W3^T as K-padded GEMM operands for the convolutions on the im2col operand
                    w3s = [Ws[blks.index(b)][2] for b in sc]
                    hip.prepack_weights(w3s)
                    pn = torch.nn.functional.pad(w1.reshape(nb_, 9 * C_, I_).transpose(1, 2), (0, -(9 * C_) % 32)).contiguous()
                    qs = pn.unbind(0)
                    for b, q in zip(sc, qs):
                        panels[b] = q
                    hip.pack_w2t_bf16(list(qs))
                    hip.pack_w2t_bf16([hip._packed_weight(w, "ci_tapco") for w in w3s])
                    hip.pack_kpad_bf16([Ws[blks.index(b)][0] for b in sc], "w1")
                    hip.pack_kpad_bf16(w3s, "w3t")
        return {b: (list(W) + [w2ts[i], panels.get(b)], cbs[i]) for i, (b, W) in enumerate(zip(blks, Ws))}

    def fwdpass(self, x, h, eval_logdet, rng, stats=None):
        """:310-324 -- returns (z, logpx) with logpx = -sum logdet (as the reference)."""
        scales = self.blocks()
        prep = self.prepare(h)
        lds = []
        for si, blocks in enumerate(scales):
            for b in blocks:
                W, cb = prep[b]
                if eval_logdet:
                    x, ld = b.forward_logdet(x, cb, W, rng, stats)
                    lds.append(ld)
                else:
                    g, _ = b.g(x, cb, W, False)
                    x = ops.axpby(x, g)
            if si < len(scales) - 1:
                x = ops.squeeze2x(x)
        for _ in range(len(scales) - 1):
            x = ops.unsqueeze2x(x)
        if not eval_logdet:
            return x
        # logpx = -sum_b logdet_b: one stack + one reduction (and views in backward) instead of a chain of 31 [B]-sized
        # subtractions with 31 negations in backward
        return x, -(torch.stack(lds).sum(0) if len(lds) > 1 else lds[0])

    def bwdpass(self, z, h, iters=None):
        """:326-335 + inverse :237-267"""
        scales = self.blocks()
        prep = self.prepare(h)
        for _ in range(len(scales) - 1):
            z = ops.squeeze2x(z)
        for si in range(len(scales) - 1, -1, -1):
            if si < len(scales) - 1:
                z = ops.unsqueeze2x(z)
            for b in reversed(scales[si]):
                W, cb = prep[b]
                z = b.inverse(z, cb, W, iters=iters)
        return z


# ----------------------------------------------------------------------------- encoder (discriminator)
class _ConvNoBias(nn.Module):
    def __init__(self, in_ch, out_ch, k, stride):
        super().__init__()
        ref = torch.nn.Conv2d(in_ch, out_ch, k, stride, k // 2, bias=False)
        w = ref.weight.detach()
        self.weight = nn.Parameter(w.permute(2, 3, 1, 0).contiguous())
        self.k, self.stride = k, stride

    def forward(self, x):
        return ops.conv2d(x, self.weight, None, None, None, self.stride, self.k // 2)


class _BatchNorm(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(ch))
        self.bias = nn.Parameter(torch.zeros(ch))
        self.register_buffer("running_mean", torch.zeros(ch))
        self.register_buffer("running_var", torch.ones(ch))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x, elu):
        stats = []
        y = ops.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, 1e-5,
                               elu, stats)
        if self.training:
            mean, var = stats[0]
            n = x.numel() // x.shape[-1]
            if _BatchNorm.pending is not None:
                _BatchNorm.pending.append((self, mean, var, n))   # the encoder applies all of them in a few launches
            else:
                _BatchNorm.update_running([(self, mean, var, n)])
        return y

    pending = None   # list while an encoder forward is collecting the running-statistics updates of its layers

    @staticmethod
    def update_running(items):
        """nn.BatchNorm2d momentum 0.1, unbiased running var -- for many layers at once (multi-tensor launches;
        one add per distinct sample count n, whose unbiasing factor is part of the add's alpha)."""
        if not items:
            return
        with torch.no_grad():
            rm, rv = [m.running_mean for m, *_ in items], [m.running_var for m, *_ in items]
            torch._foreach_mul_(rm + rv, 0.9)
            torch._foreach_add_(rm, [mean for _, mean, _, _ in items], alpha=0.1)
            for n in sorted({n for *_, n in items}):
                sel = [(m.running_var, var) for m, _, var, k in items if k == n]
                torch._foreach_add_([a for a, _ in sel], [b for _, b in sel], alpha=0.1 * n / (n - 1))
            torch._foreach_add_([m.num_batches_tracked for m, *_ in items], 1)


class ResNetBlockBatchNorm(nn.Module):
    """nnet/resnets/resnet_batchnorm.py:18-76 (activation = ELU)"""

    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = _ConvNoBias(inplanes, planes, 3, stride)
        self.bn1 = _BatchNorm(planes)
        self.conv2 = _ConvNoBias(planes, planes, 3, 1)
        self.bn2 = _BatchNorm(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(_ConvNoBias(inplanes, planes, 1, stride), _BatchNorm(planes))

    def forward(self, x):
        out = self.bn1(self.conv1(x), True)
        out = self.bn2(self.conv2(out), False)
        res = x
        if self.downsample is not None:
            res = self.downsample[1](self.downsample[0](x), False)
        return ops.elu(ops.axpby(out, res))


class _ResNet(nn.Module):
    def __init__(self, inplanes, planes, strides):
        super().__init__()
        blocks = []
        for p, s in zip(planes, strides):
            blocks.append(ResNetBlockBatchNorm(inplanes, p, s))
            inplanes = p
        self.main = nn.Sequential(*blocks)


class GlobalResNetEncoderBatchNorm(nn.Module):
    """modules/encoders/global_encoder.py:12-44"""

    def __init__(self, levels, in_planes, out_planes, hidden_planes, activation="elu"):
        super().__init__()
        net = nn.Module()
        self.levels = levels
        for lvl in range(levels):
            setattr(net, f"resnet{lvl}", _ResNet(in_planes, [hidden_planes[lvl]] * 2, [1, 2]))
            in_planes = hidden_planes[lvl]
        ref = torch.nn.Conv2d(in_planes, out_planes, 1, bias=True)
        top = nn.Module()
        top.weight = nn.Parameter(ref.weight.detach()[:, :, 0, 0].t().contiguous())  # [Cin, Cout]
        top.bias = nn.Parameter(ref.bias.detach().clone())
        net.top = top
        self.net = net

    def forward(self, x):
        outer, _BatchNorm.pending = _BatchNorm.pending, []
        try:
            for lvl in range(self.levels):
                for blk in getattr(self.net, f"resnet{lvl}").main:
                    x = blk(x)
        finally:
            items, _BatchNorm.pending = _BatchNorm.pending, outer
            _BatchNorm.update_running(items)
        x = ops.elu(ops.conv1x1(x, self.net.top.weight, self.net.top.bias))
        # reference flattens NCHW ([B, out_planes*h*w], global_encoder.py:36)
        return ops.to_nchw(x).reshape(x.shape[0], -1)


class LinearWeightNorm(nn.Module):
    """nnet/weight_norm.py:8-40: nn.utils.weight_norm(Linear), init N(0, 0.05)."""

    def __init__(self, in_f, out_f):
        super().__init__()
        lin = nn.Module()
        ref = torch.nn.Linear(in_f, out_f)
        w = torch.nn.init.normal_(ref.weight.detach().clone(), mean=0.0, std=0.05)
        lin.bias = nn.Parameter(torch.zeros(out_f))
        lin.weight_g = nn.Parameter(w.norm(dim=1, keepdim=True))
        lin.weight_v = nn.Parameter(w)
        self.linear = lin

    def forward(self, x):
        return ops.linear(x, ops.weight_norm(self.linear.weight_v, self.linear.weight_g), self.linear.bias)


# ----------------------------------------------------------------------------- prior flow on h (64-d)
class _Lin(nn.Module):
    def __init__(self, in_f, out_f):
        super().__init__()
        ref = torch.nn.Linear(in_f, out_f)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(torch.zeros(out_f))

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class NICEMLPBlock(nn.Module):
    """flows/couplings/blocks.py:11-48"""

    def __init__(self, in_f, out_f, hidden):
        super().__init__()
        self.fc1 = _Lin(in_f, hidden)
        self.fc2 = _Lin(hidden, hidden)
        self.fc3 = LinearWeightNorm(hidden, out_f)

    def forward(self, x):
        return self.fc3(ops.elu(self.fc2(ops.elu(self.fc1(x)))))


class NICE1d(nn.Module):
    """flows/couplings/coupling.py:13-177 with the Affine transform (transform.py:49-81)."""

    def __init__(self, features, hidden, split_type, up):
        super().__init__()
        out_f = features // 2
        in_f = features - out_f
        self.split_type, self.up = split_type, up
        self.z1_features = in_f if up else out_f
        self.net = NICEMLPBlock(in_f, out_f * 2, hidden)

    def _run(self, x, backward):
        """x: [B, 1, D] (the reference's 3-D latent).  The conditioning half goes through the MLP (three small GEMMs);
        split / affine transform / recombination / log-det are ONE fused launch (indm_coupling_*)."""
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if self.split_type == "continuous":
            z1, z2 = x2[:, :self.z1_features], x2[:, self.z1_features:]
        else:
            z1, z2 = x2[:, 0::2], x2[:, 1::2]
        z = z1 if self.up else z2
        params = self.net(z.contiguous())
        y, logdet = ops.coupling(x2, params, 0 if self.split_type == "continuous" else 1, self.up, backward)
        return y.reshape(shp), logdet.reshape(shp[0], -1).sum(1)


class ActNorm1dFlow(nn.Module):
    """flows/normalization.py:13-75"""

    def __init__(self, features):
        super().__init__()
        self.log_scale = nn.Parameter(torch.nn.init.normal_(torch.empty(features), mean=0, std=0.05))
        self.bias = nn.Parameter(torch.zeros(features))

    def _run(self, x, backward):
        if not backward:
            return x * self.log_scale.exp() + self.bias, self.log_scale.sum(0, keepdim=True)
        return (x - self.bias) / (self.log_scale.exp() + 1e-8), -self.log_scale.sum(0, keepdim=True)


class InvertibleLinearFlow(nn.Module):
    """flows/permutation.py:75-136.  weight_inv is a buffer refreshed only by sync() -- the reference never
    calls sync() from run_lib, so sampling uses whatever the constructor / checkpoint left there."""

    def __init__(self, features):
        super().__init__()
        self.weight = nn.Parameter(torch.nn.init.orthogonal_(torch.empty(features, features)))
        self.register_buffer("weight_inv", self.weight.data.inverse().clone())

    def sync(self):
        self.weight_inv.copy_(self.weight.data.inverse())

    def _run(self, x, backward):
        w = self.weight_inv if backward else self.weight
        shp = x.shape
        out = ops.gemm(x.reshape(-1, shp[-1]), w, trans_b=True, weight_b=True).reshape(shp)
        return out, ops.slogdet(w)


class PriorFlowUnit(nn.Module):
    ORDER = ("coupling1_up", "coupling1_dn", "actnorm", "coupling2_up", "coupling2_dn")

    def __init__(self, features, hidden):
        super().__init__()
        self.coupling1_up = NICE1d(features, hidden, "continuous", True)
        self.coupling1_dn = NICE1d(features, hidden, "continuous", False)
        self.actnorm = ActNorm1dFlow(features)
        self.coupling2_up = NICE1d(features, hidden, "skip", True)
        self.coupling2_dn = NICE1d(features, hidden, "skip", False)


class PriorFlowStep(nn.Module):
    def __init__(self, features, hidden):
        super().__init__()
        self.actnorm = ActNorm1dFlow(features)
        self.linear = InvertibleLinearFlow(features)
        self.unit = PriorFlowUnit(features, hidden)


class PriorFlow(nn.Module):
    """modules/discriminators/priors/flow.py:16-201 (inverse=True flow)."""

    def __init__(self, num_steps, features, hidden):
        super().__init__()
        self.steps = nn.ModuleList([PriorFlowStep(features, hidden) for _ in range(num_steps)])

    def run(self, x, backward):
        lds, B = [], x.shape[0]
        steps = list(self.steps)
        for st in (steps if not backward else reversed(steps)):
            seq = [st.actnorm, st.linear] + [getattr(st.unit, n) for n in PriorFlowUnit.ORDER]
            for m in (seq if not backward else reversed(seq)):
                x, ld = m._run(x, backward)
                lds.append(ld.expand(B))   # per-sample [B] or a shared [1]
        return x, torch.stack(lds).sum(0)


class FlowPrior(nn.Module):
    def __init__(self, num_steps, in_features, hidden_features, **_):
        super().__init__()
        self.flow = PriorFlow(num_steps, in_features, hidden_features)

    def calcKL(self, z, eps, mu, logvar):
        """priors/flow.py:232-253"""
        dim = z.size(2)
        cc = math.log(math.pi * 2.0)
        log_post = ((logvar.unsqueeze(1) + eps.pow(2)).sum(dim=2) + cc * dim).mean(dim=1) * -0.5
        e, logdet = self.flow.run(z, backward=False)  # bwdpass of an inverse flow = Flow.forward
        e = e.reshape(e.size(0), -1)
        log_prior = ((e * e).sum(dim=1) + cc * e.size(1)) * -0.5 + logdet
        return log_post - log_prior / z.size(1)

    def sample(self, nsamples, dim, rng):
        """priors/flow.py:226-230"""
        eps = rng.randn(nsamples, dim)
        z, _ = self.flow.run(eps, backward=True)
        return z


class GaussianDiscriminator(nn.Module):
    """modules/discriminators/gaussian.py:14-100"""

    def __init__(self, wolf_cfg):
        super().__init__()
        e = wolf_cfg.encoder
        self.dim = wolf_cfg.dim
        self.encoder = GlobalResNetEncoderBatchNorm(e.levels, e.in_planes, e.out_planes, list(e.hidden_planes))
        self.fc = LinearWeightNorm(wolf_cfg.in_dim, 2 * wolf_cfg.dim)
        p = wolf_cfg.prior
        self.prior = FlowPrior(p.num_steps, p.in_features, p.hidden_features)

    def sampling_and_KL(self, x, rng):
        c = self.fc(self.encoder(x))
        mu, logvar = c.chunk(2, dim=1)
        eps = rng.randn(mu.shape[0], 1, mu.shape[1])
        z = eps * logvar.mul(0.5).exp().unsqueeze(1) + mu.unsqueeze(1)
        return z, self.prior.calcKL(z, eps, mu, logvar)


class Generator(nn.Module):
    def __init__(self, flow):
        super().__init__()
        self.flow = flow


class WolfCore(nn.Module):
    """flow_models/wolf/wolf.py:18-145 (NHWC in / NHWC out; see flow_forward for the NCHW boundary)."""

    def __init__(self, config):
        super().__init__()
        c = config
        in_ch = c.data.num_channels * (4 if c.flow.squeeze else 1)
        self.generator = Generator(ResidualFlow(in_ch, list(map(int, c.flow.nblocks.split("-"))),
                                                c.flow.intermediate_dim))
        self.discriminator = GaussianDiscriminator(c.flow.wolf)
        self.config = config

    def forward(self, data, y=None, n_bits=8, nsamples=1, reverse=False, eval_logdet=True, rng=None, stats=None,
                iters=None):
        if rng is None:
            from .rng import default_rng
            rng = default_rng(data.device)
        if reverse:
            h = self.discriminator.prior.sample(data.shape[0], self.discriminator.dim, rng)
            return self.generator.flow.bwdpass(data, h.contiguous(), iters)
        z, kl = self.discriminator.sampling_and_KL(data, rng)
        h = z.reshape(-1, z.size(2)).contiguous()
        if eval_logdet:
            st = [] if (self.training and stats is None) else stats
            eps, logpx = self.generator.flow.fwdpass(data, h, True, rng, st)
            if self.training and st:
                self._update_moment_buffers(st)
            return eps, -logpx - kl
        return self.generator.flow.fwdpass(data, h, False, rng)

    @staticmethod
    def _update_moment_buffers(st):
        """iresblock.py:169-173, batched: one stack + two reductions instead of 3 host syncs per block."""
        with torch.no_grad():
            L = torch.stack([ld for _, _, ld in st])  # [nblocks, B]
            fir, sec = L.mean(1), (L * L).mean(1)
            for i, (blk, n, _) in enumerate(st):   # written into the buffers on demand (sync_scales): 96 launches saved
                blk.pending_moments = (float(n), fir[i:i + 1], sec[i:i + 1])

    # ------------------------------------------------------------------ checkpoint boundary
    @staticmethod
    def _to_native(k, v):
        if v.dim() == 4:
            if v.shape[2] == 1 and v.shape[3] == 1 and (".nnet." in k or k.endswith("net.top.weight")):
                return v[:, :, 0, 0].t().contiguous()
            return v.permute(2, 3, 1, 0).contiguous()
        return v

    def load_reference_state_dict(self, sd):
        own = self.state_dict()
        new = {}
        for k, v in sd.items():
            k2 = k[len("module."):] if k.startswith("module.") else k
            if k2 not in own:
                raise KeyError(k2)
            v = self._to_native(k2, torch.as_tensor(v))
            if tuple(v.shape) != tuple(own[k2].shape):
                v = v.reshape(own[k2].shape)
            new[k2] = v.to(own[k2].dtype)
        missing = set(own) - set(new)
        if missing:
            raise KeyError(f"missing keys: {sorted(missing)[:5]}")
        self.load_state_dict(new)
        self._drop_pending()
        for m in self.modules():
            if isinstance(m, IResBlock):
                m.lamb_value = float(m.lamb.detach())

    @staticmethod
    def to_reference_layout(k, v):
        """Convert a tensor shaped like parameter `k` (weight, gradient, EMA shadow) to the reference layout."""
        v = v.detach()
        if v.dim() == 4:
            return v.permute(3, 2, 0, 1).contiguous()
        if v.dim() == 2 and (".nnet." in k and k.endswith(".weight") and "h_net" not in k
                             or k.endswith("net.top.weight")):
            return v.t().contiguous()[:, :, None, None]
        return v

    def from_reference_layout(self, k, v):
        return self._to_native(k, torch.as_tensor(v))

    def sync_scales(self):
        """materialise the buffers the reference updates eagerly on every call (Lop `scale`, iResBlock moments)"""
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, LopConv):
                    m.sync_scale()
                elif isinstance(m, IResBlock) and getattr(m, "pending_moments", None) is not None:
                    n, fir, sec = m.pending_moments
                    m.last_n_samples.fill_(n)
                    m.last_firmom.copy_(fir)
                    m.last_secmom.copy_(sec)
                    m.pending_moments = None

    def state_dict(self, *args, **kwargs):
        self.sync_scales()
        return super().state_dict(*args, **kwargs)

    def _drop_pending(self):
        for m in self.modules():
            if isinstance(m, LopConv):
                m.last_l1 = None
            elif isinstance(m, IResBlock):
                m.pending_moments = None

    def reference_state_dict(self, prefix="module."):
        return {prefix + k: self.to_reference_layout(k, v) for k, v in self.state_dict().items()}


def create_flow_model(config):
    """flow_models/flow_model.py:86-111 ('wolf' branch)."""
    if config.flow.model != "wolf":
        raise NotImplementedError(config.flow.model)
    return WolfCore(config)


def flow_forward(config, flow_model, x, log_det=0, reverse=False, rng=None, stats=None, iters=None):
    """flow_models/flow_model.py:7-69 ('wolf' branch :53-67).  x, z are NCHW like the reference."""
    if config.flow.model != "wolf":
        raise NotImplementedError(config.flow.model)
    m = flow_model.module if hasattr(flow_model, "module") else flow_model
    h = ops.to_nhwc(x)
    if config.flow.squeeze:
        h = ops.squeeze2x(h)
    if not reverse:
        if log_det == 0:
            z, logdet_kl = m(h, y=None, n_bits=config.flow.n_bits, nsamples=config.flow.train_k, reverse=False,
                             eval_logdet=True, rng=rng, stats=stats)
        else:
            z = m(h, y=None, n_bits=config.flow.n_bits, nsamples=config.flow.train_k, reverse=False,
                  eval_logdet=False, rng=rng)
            logdet_kl = -1
    else:
        z = m(h, reverse=True, rng=rng, iters=iters)
        logdet_kl = -1
    if config.flow.squeeze:
        z = ops.unsqueeze2x(z)
    return ops.to_nchw(z), logdet_kl
