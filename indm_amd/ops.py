"""Differentiable operators built on the HIP kernels (indm_amd.hip).

Every op is a torch.autograd.Function whose forward AND backward call the C ABI -- autograd is
used only for bookkeeping.  Activations are NHWC ([N,H,W,C] contiguous); convolution weights are
RSCK ([KH,KW,Cin,Cout]).  Transposed convolutions / transposed GEMMs are first-class forward ops
(they are what the explicit vector-Jacobian chains of the residual flow are made of), so nothing
here needs double backward: see flow.py and DESIGN.md "log-det engine".
"""
import math

import torch
from torch.autograd import Function

from . import hip


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- dense contractions
class _Gemm(Function):
    """C = alpha*(op(A .* G) op(B) + bias + bias2[row // rpg]) + beta*res   (G optional)"""

    @staticmethod
    def forward(ctx, a, b, bias, bias2, res, trans_a, trans_b, alpha, beta, rpg, wgrad_b_splitk, gate):
        a, b, res, gate = _c(a), _c(b), _c(res), _c(gate)
        ctx.save_for_backward(a, b, gate)
        ctx.cfg = (trans_a, trans_b, alpha, beta, rpg, wgrad_b_splitk, bias is not None, bias2 is not None,
                   res is not None)
        return hip.gemm(a, b, trans_a, trans_b, alpha, _c(bias), _c(bias2), rpg, res, beta, gate=gate)

    @staticmethod
    def backward(ctx, dc):
        a, b, gate = ctx.saved_tensors
        ta, tb, alpha, beta, rpg, sk, has_bias, has_bias2, has_res = ctx.cfg
        dc = _c(dc)
        da = db = dbias = dbias2 = dres = dgate = None
        if ctx.needs_input_grad[0] or (gate is not None and ctx.needs_input_grad[11]):
            if not ta:
                dag = hip.gemm(dc, b, False, not tb, alpha)            # dC op(B)^T
            else:
                dag = hip.gemm(b, dc, tb, True, alpha)                 # op(B) dC^T  (stored [K,M])
            if gate is None:
                da = dag
            elif ctx.needs_input_grad[0] and ctx.needs_input_grad[11]:
                da, dgate = hip.mul2(dag, gate, a)      # both products in one pass over dag
            else:
                da = hip.mul(dag, gate) if ctx.needs_input_grad[0] else None
                dgate = hip.mul(dag, a) if ctx.needs_input_grad[11] else None
        if ctx.needs_input_grad[1]:
            if not tb:
                db = hip.gemm(a, dc, not ta, False, alpha, splitk=sk, gate=gate)  # op(A.*G)^T dC
            elif gate is None:
                db = hip.gemm(dc, a, True, ta, alpha, splitk=sk)                   # dC^T op(A)  (stored [N,K])
            else:  # the gate rides on the A-operand loader: form op(A.*G)^T dC = dB^T and transpose the small result
                db = hip.gemm(a, dc, not ta, False, alpha, splitk=sk, gate=gate).transpose(-1, -2).contiguous()
        if has_bias2 and ctx.needs_input_grad[3]:
            dbias2 = hip.segreduce(dc, rpg, alpha)           # per-image sums: one pass over dc
        if has_bias and ctx.needs_input_grad[2]:
            # the per-image sums already hold the whole reduction: no second pass over dc
            dbias = hip.colreduce(dbias2, None, 0) if dbias2 is not None else hip.colreduce(dc, None, 0, alpha)
        if has_res and ctx.needs_input_grad[4]:
            dres = hip.axpby(dc, None, beta, 0.0)
        return da, db, dbias, dbias2, dres, None, None, None, None, None, None, dgate


def gemm(a, b, bias=None, bias2=None, res=None, trans_a=False, trans_b=False, alpha=1.0, beta=1.0, rows_per_group=1,
         weight_b=False, gate=None):
    """weight_b=True marks b as a weight: its gradient reduces over all rows of a (split-K).
    gate: optional tensor of a's shape multiplied into a inside the GEMM's operand loader."""
    return _Gemm.apply(a, b, bias, bias2, res, trans_a, trans_b, alpha, beta, rows_per_group, weight_b, gate)


def linear(x, w, bias=None):
    """torch-layout Linear: w is [out, in]; x is [..., in]."""
    shp = x.shape
    y = gemm(x.reshape(-1, shp[-1]), w, bias, trans_b=True, weight_b=True)
    return y.reshape(*shp[:-1], w.shape[0])


def conv1x1(x, w, bias=None, bias2=None, res=None, alpha=1.0, beta=1.0, pack=None):
    """1x1 convolution / NIN on NHWC as a plain GEMM; w is [Cin, Cout]; bias2 is [N, Cout]."""
    if x.dtype == torch.bfloat16:
        return _Conv2dH16.apply(x, w, bias, bias2, res, 1, 0, alpha, beta, pack)
    N, H, W, C = x.shape
    r = res.reshape(-1, w.shape[1]) if res is not None else None
    y = gemm(x.reshape(-1, C), w, bias, bias2, r, alpha=alpha, beta=beta, rows_per_group=H * W, weight_b=True)
    return y.reshape(N, H, W, w.shape[1])


def conv1x1_t(v, w, gate=None):
    """Transpose of conv1x1 w.r.t. its input: (v .* gate) [N,H,W,Cout] -> [N,H,W,Cin]  ((v.*gate) @ w^T)."""
    N, H, W, C = v.shape
    y = gemm(v.reshape(-1, C), w, trans_b=True, weight_b=True, gate=None if gate is None else gate.reshape(-1, C))
    return y.reshape(N, H, W, w.shape[0])


class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, w, bias, bias2, res, stride, pad, alpha, beta):
        x, w, res = _c(x), _c(w), _c(res)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, alpha, beta, bias is not None, bias2 is not None, res is not None)
        return hip.conv2d(x, w, stride, pad, alpha, _c(bias), _c(bias2), res, beta)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, alpha, beta, has_bias, has_bias2, has_res = ctx.cfg
        dy = _c(dy)
        dx = dw = dbias = dbias2 = dres = None
        if ctx.needs_input_grad[0]:
            dx = hip.conv2d_dgrad(dy, w, (x.shape[1], x.shape[2]), stride, pad, alpha)
        if ctx.needs_input_grad[1]:
            dw = hip.conv2d_wgrad(x, dy, w.shape[0], w.shape[1], stride, pad, alpha)
        if has_bias2 and ctx.needs_input_grad[3]:
            dbias2 = hip.segreduce(dy, dy.shape[1] * dy.shape[2], alpha)   # per-image sums: one pass over dy
        if has_bias and ctx.needs_input_grad[2]:
            dbias = hip.colreduce(dbias2, None, 0) if dbias2 is not None else hip.colreduce(dy, None, 0, alpha)
        if has_res and ctx.needs_input_grad[4]:
            dres = hip.axpby(dy, None, beta, 0.0)
        return dx, dw, dbias, dbias2, dres, None, None, None, None


def conv2d(x, w, bias=None, bias2=None, res=None, stride=1, pad=1, alpha=1.0, beta=1.0, pack=None, res_grad_unscaled=False):
    """y = alpha*(conv(x, w) + bias + bias2[n]) + beta*res ; x NHWC, w RSCK.
    bf16 x (H16 storage): bf16 result; pack = the weight's (fwd, dg) bf16 packs when the caller holds them."""
    if x.dtype == torch.bfloat16:
        return _Conv2dH16.apply(x, w, bias, bias2, res, stride, pad, alpha, beta, pack, res_grad_unscaled)
    return _Conv2d.apply(x, w, bias, bias2, res, stride, pad, alpha, beta)


class _Conv2dT(Function):
    """t = d<conv(x,w), v .* gate>/dx : the transposed convolution as a forward op (gate optional)."""

    @staticmethod
    def forward(ctx, v, w, in_hw, stride, pad, gate):
        v, w, gate = _c(v), _c(w), _c(gate)
        ctx.save_for_backward(v, w, gate)
        ctx.cfg = (in_hw, stride, pad)
        return hip.conv2d_dgrad(v, w, in_hw, stride, pad, gate=gate)

    @staticmethod
    def backward(ctx, dt):
        v, w, gate = ctx.saved_tensors
        in_hw, stride, pad = ctx.cfg
        dt = _c(dt)
        dv = dw = dgate = None
        if ctx.needs_input_grad[0] or (gate is not None and ctx.needs_input_grad[5]):
            dvg = hip.conv2d(dt, w, stride, pad)
            if gate is None:
                dv = dvg
            elif ctx.needs_input_grad[0] and ctx.needs_input_grad[5]:
                dv, dgate = hip.mul2(dvg, gate, v)
            else:
                dv = hip.mul(dvg, gate) if ctx.needs_input_grad[0] else None
                dgate = hip.mul(dvg, v) if ctx.needs_input_grad[5] else None
        if ctx.needs_input_grad[1]:
            dw = hip.conv2d_wgrad(dt, v, w.shape[0], w.shape[1], stride, pad, dy_gate=gate)
        return dv, dw, None, None, None, dgate


def conv2d_t(v, w, in_hw, stride=1, pad=1, gate=None):
    return _Conv2dT.apply(v, w, in_hw, stride, pad, gate)


# ----------------------------------------------------------------------------- bf16-stored activations ("H16", csrc/unet_h16*.hip)
class _ToBf16(Function):
    @staticmethod
    def forward(ctx, x):
        return hip.to_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return hip.to_f32(g)


class _ToF32(Function):
    @staticmethod
    def forward(ctx, x):
        return hip.to_f32(x)

    @staticmethod
    def backward(ctx, g):
        return hip.to_bf16(g)


def to_bf16(x):
    """fp32 -> bf16 activations (the gradient comes back as fp32)"""
    return x if x.dtype == torch.bfloat16 else _ToBf16.apply(x)


def to_f32(x):
    return x if x.dtype == torch.float32 else _ToF32.apply(x)


class _AxpbyMixed(Function):
    """bf16(a * x + b * y) for an fp32 x and a bf16 y"""

    @staticmethod
    def forward(ctx, x, y, a, b):
        ctx.ab = (a, b)
        return hip.h16_axpby_mixed(_c(x), _c(y), a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.ab
        g = _c(g)
        gx = hip.to_f32(g, a) if ctx.needs_input_grad[0] else None
        gy = (g if b == 1.0 else hip.h16_scale(g, b)) if ctx.needs_input_grad[1] else None
        return gx, gy, None, None


def axpby_mixed(x32, y16, a=1.0, b=1.0):
    return _AxpbyMixed.apply(x32, y16, a, b)


class _Conv2dH16(Function):
    """conv2d / conv1x1 on bf16 NHWC activations: forward, data-grad and weight-grad on the H16 kernels; a geometry they do not
    cover (channel counts of the mini test configs, strided data-grads) runs on fp32 copies through the fp32-storage kernels."""

    @staticmethod
    def forward(ctx, x, w, bias, bias2, res, stride, pad, alpha, beta, pack, res_grad_unscaled=False):
        x, res = _c(x), _c(res)
        ctx.res_grad_unscaled = res_grad_unscaled   # the consumer of d(res) applies beta itself (group_norm_res)
        w4 = w if w.dim() == 4 else w.view(1, 1, *w.shape)
        KH, KW, Ci, Co = w4.shape
        y = None
        dg = None
        if Ci % 32 == 0 and Co % 2 == 0:
            if pack is None:
                pack = hip.h16_pack_weights([w4])[0]
            fwd, dg = pack
            y = hip.h16_conv2d(x, fwd, KH, KW, stride, pad, alpha, _c(bias), _c(bias2), res, beta)
        if y is None:
            r32 = None if res is None else hip.to_f32(res)
            y = hip.to_bf16(hip.conv2d(hip.to_f32(x), _c(w4), stride, pad, alpha, _c(bias), _c(bias2), r32, beta))
        ctx.save_for_backward(x, w, dg)
        ctx.cfg = (stride, pad, alpha, beta, bias is not None, bias2 is not None, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, dg = ctx.saved_tensors
        stride, pad, alpha, beta, has_bias, has_bias2, has_res = ctx.cfg
        w4 = w if w.dim() == 4 else w.view(1, 1, *w.shape)
        KH, KW, Ci, Co = w4.shape
        dy = _c(dy)
        N, OH, OW, _ = dy.shape
        dx = dw = dbias = dbias2 = dres = None
        dy32 = None
        if ctx.needs_input_grad[0]:
            if dg is not None and stride == 1 and Co % 32 == 0:
                dx = hip.h16_conv2d(dy, dg, KH, KW, 1, KH - 1 - pad, alpha)
            if dx is None:
                dy32 = hip.to_f32(dy)
                dx = hip.to_bf16(hip.conv2d_dgrad(dy32, _c(w4), (x.shape[1], x.shape[2]), stride, pad, alpha))
        if ctx.needs_input_grad[1]:
            dw = hip.h16_conv2d_wgrad(x, dy, KH, KW, stride, pad, alpha)
            if dw is None:
                dy32 = hip.to_f32(dy) if dy32 is None else dy32
                dw = hip.conv2d_wgrad(hip.to_f32(x), dy32, KH, KW, stride, pad, alpha)
            dw = dw.view_as(w)
        if Co % 8 == 0:
            if has_bias2 and ctx.needs_input_grad[3]:
                dbias2 = hip.h16_segreduce(dy, OH * OW, alpha)
            if has_bias and ctx.needs_input_grad[2]:
                dbias = hip.colreduce(dbias2, None, 0) if dbias2 is not None else hip.h16_segreduce(dy, N * OH * OW, alpha).view(Co)
        else:
            dy32 = hip.to_f32(dy) if dy32 is None else dy32
            if has_bias2 and ctx.needs_input_grad[3]:
                dbias2 = hip.segreduce(dy32, OH * OW, alpha)
            if has_bias and ctx.needs_input_grad[2]:
                dbias = hip.colreduce(dbias2, None, 0) if dbias2 is not None else hip.colreduce(dy32, None, 0, alpha)
        if has_res and ctx.needs_input_grad[4]:
            dres = dy if (beta == 1.0 or ctx.res_grad_unscaled) else hip.h16_scale(dy, beta)
        return dx, dw, dbias, dbias2, dres, None, None, None, None, None, None


class _GroupNormResH16(Function):
    """(y, x_skip) = (GroupNorm[+SiLU][+dropout](x), x): the block input leaves this node twice -- normalised, and as the skip
    tensor the block's last convolution (or its 1x1 shortcut) consumes -- so that BOTH gradients come back to ONE backward call:
    dx = GroupNorm'(dy) + res_scale * d(x_skip) in the GroupNorm backward kernel's own pass (no scale launch, no accumulate
    launch, three tensor passes less per residual block).  res_scale: the factor the skip's consumer left to us."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, act, drop_p, seed, res_scale):
        x = _c(x)
        sd = hip.SEED_DEV[0] if drop_p > 0 else None
        out = hip.h16_groupnorm_fwd(x, gamma, beta, G, eps, act, drop_p, seed, sd)
        ctx.h16 = out is not None
        if out is None:
            y32, mean, rstd = hip.groupnorm_fwd(hip.to_f32(x), gamma, beta, G, eps, act, drop_p, seed, sd)
            out = (hip.to_bf16(y32), mean, rstd)
        y, mean, rstd = out
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (G, act, drop_p, seed, sd, res_scale)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        G, act, drop_p, seed, sd, res_scale = ctx.cfg
        dskip = _c(dskip)
        if dy is None:   # (only the skip was used)
            return (dskip if res_scale == 1.0 else hip.h16_scale(dskip, res_scale)), None, None, None, None, None, None, None, None
        dy = _c(dy)
        if ctx.h16:
            dx, dg, db = hip.h16_groupnorm_bwd(dy, x, gamma, beta, mean, rstd, G, act, drop_p, seed, sd, dskip, res_scale)
        else:
            dx, dg, db = hip.groupnorm_bwd(hip.to_f32(dy), hip.to_f32(x), gamma, beta, mean, rstd, G, act, drop_p, seed, sd)
            if dskip is not None:
                dx = dx + res_scale * hip.to_f32(dskip)
            dx = hip.to_bf16(dx)
        return dx, dg, db, None, None, None, None, None, None


def group_norm_res(x, gamma, beta, eps=1e-6, silu=False, drop_p=0.0, seed=0, res_scale=1.0):
    """bf16 x -> (GroupNorm(x), x_skip); see _GroupNormResH16"""
    C = x.shape[-1]
    return _GroupNormResH16.apply(x, gamma, beta, min(C // 4, 32), eps, 1 if silu else 0, float(drop_p), int(seed), float(res_scale))


class _AttnCoreH16(Function):
    """AttnBlockpp after its GroupNorm on bf16 activations (models/layerspp.py:88-104): the three NIN projections as ONE GEMM
    (N = 3 C), the fused bf16 attention kernel, the output projection with the skip connection in its epilogue -- and the
    hand-written backward of that chain.  packs = (fwd_qkv [3C, C], dg_qkv [C, 3C], fwd_3 [C, C], dg_3 [C, C]) bf16."""

    @staticmethod
    def forward(ctx, h, x, W0, b0, W1, b1, W2, b2, W3, b3, packs, scale, a):
        h, x = _c(h), _c(x)
        fwd_qkv, dg_qkv, fwd_3, dg_3 = packs
        B, H, Wd, C = h.shape
        T = H * Wd
        bqkv = torch.cat([b0.detach(), b1.detach(), b2.detach()])
        qkv = hip.h16_conv2d(h, fwd_qkv, 1, 1, 1, 0, 1.0, bqkv)
        if qkv is None:
            raise RuntimeError("indm_h16_conv2d declined the QKV projection (the caller checks the geometry)")
        out = hip.h16_attn_fwd(qkv.view(B * T, 3 * C), B, T, C, scale)
        if out is None:
            raise RuntimeError("indm_h16_attn_fwd declined the launch (the caller checks the geometry)")
        o, lse = out
        y = hip.h16_conv2d(o.view(B, H, Wd, C), fwd_3, 1, 1, 1, 0, a, _c(b3), None, x, a)
        ctx.save_for_backward(h, qkv, o, lse, dg_qkv, dg_3)
        ctx.cfg = (scale, a)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, qkv, o, lse, dg_qkv, dg_3 = ctx.saved_tensors
        scale, a = ctx.cfg
        dy = _c(dy)
        B, H, Wd, C = h.shape
        T = H * Wd
        do = hip.h16_conv2d(dy, dg_3, 1, 1, 1, 0, a)
        dW3 = hip.h16_conv2d_wgrad(o.view(B, H, Wd, C), dy, 1, 1, 1, 0, a).view(C, C)
        db3 = hip.h16_segreduce(dy, B * T, a).view(C)
        dx = dy if a == 1.0 else hip.h16_scale(dy, a)
        dqkv = hip.h16_attn_bwd(qkv.view(B * T, 3 * C), o, do.view(B * T, C), lse, B, T, C, scale)
        dq4 = dqkv.view(B, H, Wd, 3 * C)
        dh = hip.h16_conv2d(dq4, dg_qkv, 1, 1, 1, 0)
        dWqkv = hip.h16_conv2d_wgrad(h, dq4, 1, 1, 1, 0).view(C, 3 * C)
        dbqkv = hip.h16_segreduce(dqkv, B * T).view(3 * C)
        dWs = [dWqkv[:, i * C:(i + 1) * C].contiguous() for i in range(3)]   # (contiguous HERE: inside a captured backward)
        dbs = [dbqkv[i * C:(i + 1) * C] for i in range(3)]
        return dh, dx, dWs[0], dbs[0], dWs[1], dbs[1], dWs[2], dbs[2], dW3, db3, None, None, None


def attn_core_h16(h, x, nins, packs, scale, a):
    (W0, b0), (W1, b1), (W2, b2), (W3, b3) = nins
    return _AttnCoreH16.apply(h, x, W0, b0, W1, b1, W2, b2, W3, b3, packs, scale, a)


def attn_h16_covered(T, C):
    return T in (64, 128, 256) and C >= 128 and C % 128 == 0


class _GroupNormH16(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, act, drop_p, seed):
        x = _c(x)
        sd = hip.SEED_DEV[0] if drop_p > 0 else None
        out = hip.h16_groupnorm_fwd(x, gamma, beta, G, eps, act, drop_p, seed, sd)
        ctx.h16 = out is not None
        if out is None:
            y32, mean, rstd = hip.groupnorm_fwd(hip.to_f32(x), gamma, beta, G, eps, act, drop_p, seed, sd)
            out = (hip.to_bf16(y32), mean, rstd)
        y, mean, rstd = out
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (G, act, drop_p, seed, sd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        G, act, drop_p, seed, sd = ctx.cfg
        dy = _c(dy)
        if ctx.h16:
            dx, dg, db = hip.h16_groupnorm_bwd(dy, x, gamma, beta, mean, rstd, G, act, drop_p, seed, sd)
        else:
            dx, dg, db = hip.groupnorm_bwd(hip.to_f32(dy), hip.to_f32(x), gamma, beta, mean, rstd, G, act, drop_p, seed, sd)
            dx = hip.to_bf16(dx)
        return dx, dg, db, None, None, None, None, None


# ----------------------------------------------------------------------------- normalisation / softmax
class _GroupNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, act, drop_p, seed):
        x = _c(x)
        sd = hip.SEED_DEV[0] if drop_p > 0 else None
        y, mean, rstd = hip.groupnorm_fwd(x, gamma, beta, G, eps, act, drop_p, seed, sd)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (G, act, drop_p, seed, sd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        G, act, drop_p, seed, sd = ctx.cfg
        dx, dg, db = hip.groupnorm_bwd(_c(dy), x, gamma, beta, mean, rstd, G, act, drop_p, seed, sd)
        return dx, dg, db, None, None, None, None, None


def group_norm(x, gamma, beta, eps=1e-6, silu=False, drop_p=0.0, seed=0):
    """nn.GroupNorm(min(C/4,32), C, eps) [+ SiLU] [+ dropout], fused (models/layerspp.py:232-279)."""
    C = x.shape[-1]
    fn = _GroupNormH16 if x.dtype == torch.bfloat16 else _GroupNorm
    return fn.apply(x, gamma, beta, min(C // 4, 32), eps, 1 if silu else 0, float(drop_p), int(seed))


class _BatchNormAct(Function):
    """Training-mode BatchNorm2d (+ELU).  Returns y; batch mean / biased var are stashed on ctx.stats."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act, stats_out):
        x = _c(x)
        mean, var, rstd = hip.bn_stats(x, eps)
        stats_out.append((mean, var))
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.act = act
        return hip.bn_apply(x, mean, rstd, gamma, beta, act)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        dx, dg, db = hip.bn_bwd(_c(dy), x, mean, rstd, gamma, beta, ctx.act)
        return dx, dg, db, None, None, None


class _BatchNormEval(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, mean, rstd, act):
        x = _c(x)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.act = act
        return hip.bn_apply(x, mean, rstd, gamma, beta, act)

    @staticmethod
    def backward(ctx, dy):  # statistics are constants
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        C = x.shape[-1]
        dyu = _c(dy)
        if ctx.act == 2:
            u = hip.bn_apply(x, mean, rstd, gamma, beta, 0)
            dyu = hip.act_bwd(dyu, u, hip.ACT_ELU)
        xh = hip.bn_apply(x, mean, rstd, torch.ones_like(gamma), torch.zeros_like(beta), 0)
        dg = hip.colreduce(dyu, xh, 1)
        db = hip.colreduce(dyu, None, 0)
        dx = hip.bn_apply(dyu, torch.zeros_like(mean), torch.ones_like(rstd), gamma * rstd, torch.zeros_like(beta), 0)
        return dx, dg, db, None, None, None


def batch_norm_act(x, gamma, beta, running_mean, running_var, training, eps=1e-5, elu=False, stats_out=None):
    act = 2 if elu else 0
    if training:
        return _BatchNormAct.apply(x, gamma, beta, eps, act, stats_out if stats_out is not None else [])
    rstd = torch.rsqrt(running_var + eps)
    return _BatchNormEval.apply(x, gamma, beta, running_mean, rstd, act)


class _Softmax(Function):
    @staticmethod
    def forward(ctx, x, scale):
        y = hip.softmax_fwd(_c(x), scale)
        ctx.save_for_backward(y)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return hip.softmax_bwd(_c(dy), y, ctx.scale), None


def softmax(x, scale=1.0):
    """softmax(scale * x) over the last dim."""
    return _Softmax.apply(x, scale)


class _Coupling(Function):
    """Affine coupling of the prior flow: (y, logdet) from x [B, 2F] and the MLP's params [B, 2F] (one launch each way)."""

    @staticmethod
    def forward(ctx, x, prm, split, up, inverse):
        x, prm = _c(x), _c(prm)
        ctx.save_for_backward(x, prm)
        ctx.cfg = (split, up, inverse)
        return hip.coupling_fwd(x, prm, split, up, inverse)

    @staticmethod
    def backward(ctx, dy, dld):
        x, prm = ctx.saved_tensors
        split, up, inverse = ctx.cfg
        dy = torch.zeros_like(x) if dy is None else _c(dy)
        dx, dprm = hip.coupling_bwd(dy, None if dld is None else _c(dld), x, prm, split, up, inverse)
        return dx, dprm, None, None, None


def coupling(x, prm, split, up, inverse):
    return _Coupling.apply(x, prm, split, up, inverse)


class _Attention(Function):
    """h = softmax(q k^T * scale) v: fused forward (the [T,T] logits never reach HBM).  Backward = recompute the
    probabilities with one batched MFMA GEMM + the wavefront softmax and run the five batched GEMMs of the textbook
    gradient: on this part that composition takes ~330 us per block at batch 128 against 575 us for the fully fused
    recompute-in-registers backward (indm_attn_bwd_f32: 7 GEMM units at one wave per SIMD) -- the
    transient [B,T,T] tensors (33.5 MB each at T = 256) live only inside this call.  INDM_ATTN_FUSED_BWD=1 selects the
    fused backward (kept, tested: it is the memory-lean variant)."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        q, k, v = _c(q), _c(k), _c(v)
        out = hip.attn_fwd(q, k, v, scale)
        if out is None:   # attention() pre-checks the geometry; what is left is a pointer that is not 16-byte aligned
            raise RuntimeError("indm_attn_fwd_f32 declined the launch (q/k/v must be 16-byte aligned [B,T,C] tensors)")
        h, lse = out
        ctx.save_for_backward(q, k, v, h, lse)
        ctx.scale = scale
        return h

    @staticmethod
    def backward(ctx, dh):
        q, k, v, h, lse = ctx.saved_tensors
        dh = _c(dh)
        if FUSED_ATTN_BWD["on"]:
            dq, dk, dv = hip.attn_bwd(q, k, v, h, dh, lse, ctx.scale)
            return dq, dk, dv, None
        p = hip.softmax_fwd(hip.gemm(q, k, trans_b=True), ctx.scale)          # [B,T,T] probabilities, recomputed
        dv = hip.gemm(p, dh, trans_a=True)                                    # P^T dh
        ds = hip.softmax_bwd(hip.gemm(dh, v, trans_b=True), p, ctx.scale)     # dL/d(logits)
        del p
        dq = hip.gemm(ds, k)                                                  # dS K
        dk = hip.gemm(ds, q, trans_a=True)                                    # dS^T Q
        return dq, dk, dv, None


import os as _os
FUSED_ATTN_BWD = {"on": _os.environ.get("INDM_ATTN_FUSED_BWD", "0") == "1"}


def attention(q, k, v, scale):
    """Single-head attention over [B, T, C] (models/layerspp.py:88-104).  Fused kernel when it covers the geometry
    (T in {64, 128, 256}, C % 128 == 0, fp32 mode), else the batched-GEMM + softmax composition."""
    B, T, C = q.shape
    if (hip.compute_dtype() == "f32" and T in (64, 128, 256) and C >= 128 and C % 128 == 0
            and all(t.data_ptr() % 16 == 0 and t.is_contiguous() for t in (q, k, v))):
        return _Attention.apply(q, k, v, scale)
    w = softmax(gemm(q, k, trans_b=True), scale)
    return gemm(w, v)


# ----------------------------------------------------------------------------- elementwise
class _SinCos(Function):
    """(a, d) = (sin(2 pi p)/(2 pi), cos(2 pi p)) -- the Sin activation and its derivative in one pass."""

    @staticmethod
    def forward(ctx, p):
        a, d = hip.sincos_fwd(_c(p), True)
        ctx.save_for_backward(a, d)
        return a, d

    @staticmethod
    def backward(ctx, da, dd):
        a, d = ctx.saved_tensors
        return hip.sincos_bwd(_c(da), _c(dd), a, d)


def sincos(p):
    return _SinCos.apply(p)


class _Act(Function):
    @staticmethod
    def forward(ctx, x, kind):
        x = _c(x)
        ctx.save_for_backward(x)
        ctx.kind = kind
        return hip.act_fwd(x, kind)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return hip.act_bwd(_c(dy), x, ctx.kind), None


def silu(x):
    return _Act.apply(x, hip.ACT_SILU)


def elu(x):
    return _Act.apply(x, hip.ACT_ELU)


class _Axpby(Function):
    @staticmethod
    def forward(ctx, x, y, a, b, c):
        ctx.ab = (a, b)
        return hip.axpby(_c(x), _c(y), a, b, c)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.ab
        g = _c(g)
        gx = hip.axpby(g, None, a, 0.0) if ctx.needs_input_grad[0] else None
        gy = hip.axpby(g, None, b, 0.0) if ctx.needs_input_grad[1] else None
        return gx, gy, None, None, None


def axpby(x, y, a=1.0, b=1.0, c=0.0):
    """a*x + b*y + c"""
    return _Axpby.apply(x, y, a, b, c)


class _Mul(Function):
    @staticmethod
    def forward(ctx, x, y):
        x, y = _c(x), _c(y)
        ctx.save_for_backward(x, y)
        return hip.mul(x, y)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = _c(g)
        return (hip.mul(g, y) if ctx.needs_input_grad[0] else None,
                hip.mul(g, x) if ctx.needs_input_grad[1] else None)


def mul(x, y):
    return _Mul.apply(x, y)


class _SplitCols(Function):
    """[R, sum C_b] -> the per-block dense [R, C_b] matrices (views of one flat buffer): one launch each way."""

    @staticmethod
    def forward(ctx, x, *sizes):
        x = _c(x)
        flat = hip.split_cols(x, sizes)
        ctx.sizes, ctx.R = sizes, x.shape[0]
        ctx.set_materialize_grads(False)
        outs, o = [], 0
        for c in sizes:
            outs.append(flat[o:o + ctx.R * c].view(ctx.R, c))
            o += ctx.R * c
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        R = ctx.R
        ref = next((g for g in grads if g is not None), None)
        if ref is None:
            return (None,) * (1 + len(ctx.sizes))
        gs = [g if g is not None else ref.new_zeros((R, c)) for g, c in zip(grads, ctx.sizes)]
        return (torch.cat(gs, dim=1),) + (None,) * len(ctx.sizes)


def split_cols(x, sizes):
    """tuple of contiguous [R, C_b] column blocks of x [R, sum(sizes)]"""
    return _SplitCols.apply(x, *[int(v) for v in sizes])


class _WeightNorm(Function):
    @staticmethod
    def forward(ctx, v, g):
        v, g = _c(v), _c(g)
        w, n = hip.weightnorm_fwd(v, g)
        ctx.save_for_backward(v, g, n)
        return w

    @staticmethod
    def backward(ctx, dw):
        v, g, n = ctx.saved_tensors
        dv, dg = hip.weightnorm_bwd(_c(dw), v, g, n)
        return dv, dg.view_as(g)


def weight_norm(v, g):
    """nn.utils.weight_norm: v [out, in] * g [out(,1)] / ||v||_row  (one launch forward, one backward)"""
    return _WeightNorm.apply(v, g)


class _RowDot(Function):
    """out[b] = <x[b], y[b]>"""

    @staticmethod
    def forward(ctx, x, y):
        x, y = _c(x), _c(y)
        ctx.save_for_backward(x, y)
        return hip.rowreduce(x, y, 0)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = _c(g)
        return (hip.scale_rows(y, g) if ctx.needs_input_grad[0] else None,
                hip.scale_rows(x, g) if ctx.needs_input_grad[1] else None)


def rowdot(x, y):
    return _RowDot.apply(x, y)


class _RowSumSq(Function):
    """out[b] = alpha * sum x[b]^2"""

    @staticmethod
    def forward(ctx, x, alpha):
        x = _c(x)
        ctx.save_for_backward(x)
        ctx.alpha = alpha
        return hip.rowreduce(x, None, 1, alpha)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return hip.scale_rows(x, _c(g) * (2.0 * ctx.alpha)), None


def rowsumsq(x, alpha=1.0):
    return _RowSumSq.apply(x, alpha)


class _ScaleRows(Function):
    @staticmethod
    def forward(ctx, x, s):
        x, s = _c(x), _c(s)
        ctx.save_for_backward(x, s)
        return hip.scale_rows(x, s)

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        g = _c(g)
        gx = hip.scale_rows(g, s) if ctx.needs_input_grad[0] else None
        gs = hip.rowreduce(g, x, 0) if ctx.needs_input_grad[1] else None
        return gx, gs


def scale_rows(x, s):
    """x[b,...] * s[b]"""
    return _ScaleRows.apply(x, s)


# ----------------------------------------------------------------------------- resampling / layout
class _Resample(Function):
    """kind: 'up' (nearest x2), 'down' (2x2 mean), 'squeeze', 'unsqueeze', 'to_nhwc', 'to_nchw'."""
    INV = {"squeeze": "unsqueeze", "unsqueeze": "squeeze", "to_nhwc": "to_nchw", "to_nchw": "to_nhwc"}

    @staticmethod
    def run(kind, x, alpha=None):
        if x.dtype == torch.bfloat16:   # H16 storage: nearest / mean resampling on bf16 (other kinds never see bf16)
            if kind not in ("up", "down"):
                raise RuntimeError(f"_Resample: {kind} on a bfloat16 tensor")
            if x.shape[-1] % 8:
                return hip.to_bf16(_Resample.run(kind, hip.to_f32(x), alpha))
            return hip.h16_upsample2x(x, 1.0 if alpha is None else alpha) if kind == "up" else \
                hip.h16_downsample2x(x, 0.25 if alpha is None else alpha)
        if kind == "up":
            return hip.upsample2x(x, 1.0 if alpha is None else alpha)
        if kind == "down":
            return hip.downsample2x(x, 0.25 if alpha is None else alpha)
        return {"squeeze": hip.squeeze2x, "unsqueeze": hip.unsqueeze2x, "to_nhwc": hip.nchw_to_nhwc,
                "to_nchw": hip.nhwc_to_nchw}[kind](x)

    @staticmethod
    def forward(ctx, x, kind):
        ctx.kind = kind
        return _Resample.run(kind, _c(x))

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        k = ctx.kind
        if k == "up":      # adjoint of replication = 2x2 sum
            return _Resample.run("down", g, 1.0), None
        if k == "down":    # adjoint of 2x2 mean = replicate / 4
            return _Resample.run("up", g, 0.25), None
        return _Resample.run(_Resample.INV[k], g), None


def upsample2x(x):
    return _Resample.apply(x, "up")


def downsample2x(x):
    return _Resample.apply(x, "down")


def squeeze2x(x):
    return _Resample.apply(x, "squeeze")


def unsqueeze2x(x):
    return _Resample.apply(x, "unsqueeze")


def to_nhwc(x):
    return _Resample.apply(x, "to_nhwc")


def to_nchw(x):
    return _Resample.apply(x, "to_nchw")


class _UpFirDn2d(Function):
    """NHWC up-FIR-down with the gradient relations of the reference op (op/upfirdn2d.py:19-142):
    grad = the same op with up<->down, flipped kernel and g_pad."""

    @staticmethod
    def forward(ctx, x, kernel, up, down, pad):
        x = _c(x)                               # fp32, or bf16 (H16 storage: indm_upfirdn2d_bf16)
        px0, px1, py0, py1 = pad
        N, H, W, C = x.shape
        kh, kw = kernel.shape
        y = hip.upfirdn2d(x, kernel, up[0], up[1], down[0], down[1], px0, px1, py0, py1, 1)
        oh, ow = y.shape[1], y.shape[2]
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, (H, W), (kw - px0 - 1, W * up[0] - ow * down[0] + px0 - up[0] + 1,
                                     kh - py0 - 1, H * up[1] - oh * down[1] + py0 - up[1] + 1))
        return y

    @staticmethod
    def backward(ctx, gy):
        (kernel,) = ctx.saved_tensors
        up, down, _, gp = ctx.cfg
        gk = torch.flip(kernel, [0, 1]).contiguous()
        gx = hip.upfirdn2d(_c(gy), gk, down[0], down[1], up[0], up[1], gp[0], gp[1], gp[2], gp[3], 1)
        return gx, None, None, None, None


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """Reference call convention (op/upfirdn2d.py:145-156) on NHWC tensors."""
    return _UpFirDn2d.apply(x, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))


# ----------------------------------------------------------------------------- flow / SDE specials
class _LopNormalize(Function):
    @staticmethod
    def forward(ctx, w, coeff):
        w = _c(w)
        wn, l1 = hip.lop_normalize_fwd(w, coeff)
        ctx.save_for_backward(w, l1)
        ctx.coeff = coeff
        ctx.mark_non_differentiable(l1)
        return wn, l1

    @staticmethod
    def backward(ctx, dwn, _):
        w, l1 = ctx.saved_tensors
        return hip.lop_normalize_bwd(_c(dwn), w, l1, ctx.coeff), None


class _LopNormalizeMulti(Function):
    """Lop normalisation of many weights at once (two launches forward, two backward)."""

    @staticmethod
    def forward(ctx, coeff, *ws):
        ws = [_c(w) for w in ws]
        wns, l1s = hip.lop_normalize_multi_fwd(ws, coeff)
        ctx.save_for_backward(*ws, *l1s)
        ctx.coeff, ctx.n = coeff, len(ws)
        ctx.mark_non_differentiable(*l1s)
        ctx.set_materialize_grads(False)   # no zero-fill launch per l1 output (96 per step on the flow)
        return tuple(wns) + tuple(l1s)

    @staticmethod
    def backward(ctx, *grads):
        n = ctx.n
        ws, l1s = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        dwns = [_c(g) if g is not None else torch.zeros_like(w) for g, w in zip(grads[:n], ws)]
        dws = hip.lop_normalize_multi_bwd(dwns, list(ws), list(l1s), ctx.coeff)
        return (None,) + tuple(dws)


def lop_normalize_multi(ws, coeff=0.98):
    """-> (list of normalised weights, list of l1[Cout] vectors)"""
    out = _LopNormalizeMulti.apply(coeff, *ws)
    n = len(ws)
    return list(out[:n]), list(out[n:])


def lop_normalize(w, coeff=0.98):
    """w / max(1, ||w[..., co]||_1 / coeff)  (layers/base/lipschitz.py:350-359 on RSCK weights) -> (wn, l1)."""
    return _LopNormalize.apply(w, coeff)


class _SdePerturb(Function):
    @staticmethod
    def forward(ctx, x, noise, mc, sd):
        ctx.save_for_backward(mc)
        return hip.sde_perturb(_c(x), _c(noise), mc, sd)

    @staticmethod
    def backward(ctx, g):
        (mc,) = ctx.saved_tensors
        return hip.scale_rows(_c(g), mc), None, None, None


def sde_perturb(x, noise, mean_coef, std):
    """mean_coef[b]*x + std[b]*noise (sde_lib.py:153-157 + losses.py:105); differentiable in x only."""
    return _SdePerturb.apply(x, noise, mean_coef, std)


class _DsmLoss(Function):
    @staticmethod
    def forward(ctx, out, z, std, w, kind, mode, reduce_mean):
        out, z = _c(out), _c(z)
        ctx.save_for_backward(out, z, std, w)
        ctx.cfg = (kind, mode, reduce_mean)
        return hip.dsm_loss_fwd(out, z, std, w, kind, mode, reduce_mean)

    @staticmethod
    def backward(ctx, gl):
        out, z, std, w = ctx.saved_tensors
        kind, mode, rm = ctx.cfg
        return hip.dsm_loss_bwd(_c(gl), out, z, std, w, kind, mode, rm), None, None, None, None, None, None


def dsm_loss(out, z, std, w, kind, mode, reduce_mean):
    """Per-image denoising score matching loss from the raw network output (losses.py:106-118)."""
    return _DsmLoss.apply(out, z, std, w, kind, mode, reduce_mean)


class _SlogdetInv(Function):
    @staticmethod
    def forward(ctx, w):
        ld, inv = hip.slogdet_inv(_c(w))
        ctx.save_for_backward(inv)
        return ld

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        return inv.t().contiguous() * g


def slogdet(w):
    """log|det w| as a [1] tensor, differentiable (d/dw = w^-T)."""
    return _SlogdetInv.apply(w)


INV_SQRT2 = 1.0 / math.sqrt(2.0)
