"""Plain-PyTorch statement of what every entry point of indm_amd.hip computes.

TEST INFRASTRUCTURE ONLY.  Two uses:
  * `-m gpu` kernel tests compare each hand-written HIP kernel with the function of the same
    name here (run on CPU in fp64/fp32) -- the "plain PyTorch reference of the same op".
  * `-m "not gpu"` host-logic tests monkeypatch indm_amd.hip with these functions (see
    conftest.emulate_hip) so the autograd wiring / module graph can be checked against the
    oracle without a GPU.  The package itself never imports this file.
"""
import math

import torch
import torch.nn.functional as F


def gemm(a, b, trans_a=False, trans_b=False, alpha=1.0, bias=None, bias2=None, rows_per_group=1, res=None,
         beta=1.0, splitk=False, gate=None):
    if gate is not None:
        a = a * gate
    A = a.transpose(-1, -2) if trans_a else a
    Bm = b.transpose(-1, -2) if trans_b else b
    out = torch.matmul(A, Bm)
    if bias is not None:
        out = out + bias
    if bias2 is not None:
        out = out + bias2.repeat_interleave(rows_per_group, dim=0)
    out = alpha * out
    if res is not None:
        out = out + beta * res
    return out.contiguous()


def tap_gather(P, bias, sign, alpha=1.0):
    N, H, W, C9 = P.shape
    C = C9 // 9
    Pp = F.pad(P, (0, 0, 1, 1, 1, 1))
    acc = torch.zeros((N, H, W, C), dtype=P.dtype)
    for kh in range(3):
        for kw in range(3):
            dh, dw = sign * (kh - 1), sign * (kw - 1)
            t = kh * 3 + kw
            acc = acc + Pp[:, 1 + dh:1 + dh + H, 1 + dw:1 + dw + W, t * C:(t + 1) * C]
    if bias is not None:
        acc = acc + bias
    return alpha * acc


def tap_gather_acc(P, sign, mulp=None, acc_in=None, coeff=0.0, alpha=1.0):
    y = tap_gather(P, None, sign, alpha)
    if mulp is not None:
        y = y * mulp
    return y, (acc_in + coeff * y if acc_in is not None else None)


def conv3x3_thin(x, w2d, sign, alpha=1.0, bias=None, bias2=None, res=None, beta=1.0):
    N, H, W, C = x.shape
    xcol = tap_spread(x, sign).reshape(-1, 9 * C)
    r = res.reshape(-1, w2d.shape[1]) if res is not None else None
    return gemm(xcol, w2d, alpha=alpha, bias=bias, bias2=bias2, rows_per_group=H * W, res=r, beta=beta).view(N, H, W, -1)


def gemm_sin(a, b_nk, bias=None, bias2=None, rows_per_group=1):
    return sincos_fwd(gemm(a, b_nk, trans_b=True, bias=bias, bias2=bias2, rows_per_group=rows_per_group))


def conv3x3_thin_act(x, w2d, sign=-1, bias=None, bias2=None):
    return sincos_fwd(conv3x3_thin(x, w2d, sign, 1.0, bias, bias2))


def tap_spread(y, sign, alpha=1.0):
    N, H, W, C = y.shape
    yp = F.pad(y, (0, 0, 1, 1, 1, 1))
    outs = []
    for kh in range(3):
        for kw in range(3):
            dh, dw = -sign * (kh - 1), -sign * (kw - 1)
            outs.append(yp[:, 1 + dh:1 + dh + H, 1 + dw:1 + dw + W, :])
    return alpha * torch.cat(outs, dim=-1).contiguous()


def conv_out_size(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


def _w_oihw(w):
    return w.permute(3, 2, 0, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def conv2d(x, w, stride=1, pad=1, alpha=1.0, bias=None, bias2=None, res=None, beta=1.0):
    y = F.conv2d(_nchw(x), _w_oihw(w), None, stride, pad).permute(0, 2, 3, 1)
    if bias is not None:
        y = y + bias
    if bias2 is not None:
        y = y + bias2[:, None, None, :]
    y = alpha * y
    if res is not None:
        y = y + beta * res
    return y.contiguous()


def conv2d_dgrad(dy, w, in_hw, stride=1, pad=1, alpha=1.0, res=None, beta=1.0, gate=None):
    if gate is not None:
        dy = dy * gate
    N, OH, OW, Co = dy.shape
    KH, KW, Ci, _ = w.shape
    H, W = in_hw
    dx = torch.nn.grad.conv2d_input((N, Ci, H, W), _w_oihw(w), _nchw(dy), stride, pad)
    dx = alpha * dx.permute(0, 2, 3, 1)
    if res is not None:
        dx = dx + beta * res
    return dx.contiguous()


def conv2d_wgrad(x, dy, kh, kw, stride=1, pad=1, alpha=1.0, dy_gate=None, x_gate=None):
    if dy_gate is not None:
        dy = dy * dy_gate
    if x_gate is not None:
        x = x * x_gate
    N, H, W, Ci = x.shape
    Co = dy.shape[-1]
    dw = torch.nn.grad.conv2d_weight(_nchw(x), (Co, Ci, kh, kw), _nchw(dy), stride, pad)
    return (alpha * dw.permute(2, 3, 1, 0)).contiguous()


def _coupling_split(x, split, up):
    F_ = x.shape[1] // 2
    if split == 0:
        z1, z2 = x[:, :F_], x[:, F_:]
    else:
        z1, z2 = x[:, 0::2], x[:, 1::2]
    return (z1, z2) if up else (z2, z1)


def _coupling_merge(z, zp, split, up):
    z1, z2 = (z, zp) if up else (zp, z)
    if split == 0:
        return torch.cat([z1, z2], dim=1)
    return torch.stack([z1, z2], dim=-1).reshape(z.shape[0], -1)


def coupling_fwd(x, prm, split, up, inverse):
    z, zp = _coupling_split(x, split, up)
    mu, ls = prm.chunk(2, dim=1)
    sc = torch.sigmoid(ls + 2.0) + 1e-3
    if not inverse:
        return _coupling_merge(z, sc * zp + mu, split, up).contiguous(), sc.log().sum(1)
    return _coupling_merge(z, (zp - mu) / (sc + 1e-12), split, up).contiguous(), -sc.log().sum(1)


def coupling_bwd(dy, dld, x, prm, split, up, inverse):
    xx, pp = x.detach().clone().requires_grad_(True), prm.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        y, ld = coupling_fwd(xx, pp, split, up, inverse)
        gl = torch.zeros_like(ld) if dld is None else dld
        gx, gp = torch.autograd.grad([y, ld], [xx, pp], [dy, gl])
    return gx, gp


def attn_fwd(q, k, v, scale):
    s = torch.matmul(q, k.transpose(1, 2)) * scale
    return torch.matmul(torch.softmax(s, dim=-1), v).contiguous(), torch.logsumexp(s, dim=-1)


def attn_bwd(q, k, v, h, dh, lse, scale):
    s = torch.matmul(q, k.transpose(1, 2)) * scale
    pr = torch.exp(s - lse[..., None])
    dv = torch.matmul(pr.transpose(1, 2), dh)
    dp = torch.matmul(dh, v.transpose(1, 2))
    ds = pr * (dp - (dh * h).sum(-1, keepdim=True)) * scale
    return torch.matmul(ds, k).contiguous(), torch.matmul(ds.transpose(1, 2), q).contiguous(), dv.contiguous()


def ires_chain(t3, d2, w2t, d1, w1p, J, want_t2=False):
    t2 = (t3 * d2) @ w2t
    P = ((t2 * d1) @ w1p[:, :J]).contiguous()
    return (P, t2.contiguous()) if want_t2 else P


def ires_bwd_act(da, du, d, a, t, want_dt=False, rows_per_seg=None):
    dp = da * d - (2.0 * math.pi) ** 2 * a * (du * t)
    dt = du * d if want_dt else None
    seg = None
    if rows_per_seg is not None:
        C = a.shape[-1]
        seg = dp.reshape(-1, rows_per_seg, C).sum(1)
    return dp, dt, seg


def gemm_ires_bwd_act(x, b_nk, du, d, a, t, want_dt=False, rows_per_seg=None):
    return ires_bwd_act(gemm(x, b_nk, trans_b=True), du, d, a, t, want_dt, rows_per_seg)


def gemm_gated_ires_bwd_act(x, gate, b_kn, da, d, a, t, rows_per_seg=None):
    du = gemm(x, b_kn, gate=gate)
    dp, _, seg = ires_bwd_act(da, du, d, a, t, False, rows_per_seg)
    return du, dp, seg


def _dropout_mask_unavailable(drop_p):
    if drop_p > 0:
        raise NotImplementedError("Philox dropout masks exist only in the HIP kernel")


def _gn_u(x, gamma, beta, mean, rstd, G):
    N, C = x.shape[0], x.shape[-1]
    xs = x.reshape(N, -1, G, C // G)
    xh = (xs - mean[:, None, :, None]) * rstd[:, None, :, None]
    xh = xh.reshape(x.shape)
    return xh, xh * gamma + beta


def groupnorm_fwd(x, gamma, beta, G, eps, act=0, drop_p=0.0, seed=0, seed_dev=None):
    _dropout_mask_unavailable(drop_p)
    N, C = x.shape[0], x.shape[-1]
    xs = x.reshape(N, -1, G, C // G)
    mean = xs.mean(dim=(1, 3))
    var = xs.var(dim=(1, 3), unbiased=False)
    rstd = torch.rsqrt(var + eps)
    _, u = _gn_u(x, gamma, beta, mean, rstd, G)
    y = F.silu(u) if act == 1 else u
    return y.contiguous(), mean, rstd


def groupnorm_bwd(dy, x, gamma, beta, mean, rstd, G, act=0, drop_p=0.0, seed=0, seed_dev=None):
    _dropout_mask_unavailable(drop_p)
    N, C = x.shape[0], x.shape[-1]
    xh, u = _gn_u(x, gamma, beta, mean, rstd, G)
    d = dy
    if act == 1:
        s = torch.sigmoid(u)
        d = dy * s * (1 + u * (1 - s))
    red = tuple(range(x.dim() - 1))
    dgamma = (d * xh).sum(red)
    dbeta = d.sum(red)
    gd = (d * gamma).reshape(N, -1, G, C // G)
    xhs = xh.reshape(N, -1, G, C // G)
    s1 = gd.mean(dim=(1, 3), keepdim=True)
    s2 = (gd * xhs).mean(dim=(1, 3), keepdim=True)
    dx = rstd[:, None, :, None] * (gd - s1 - xhs * s2)
    return dx.reshape(x.shape).contiguous(), dgamma, dbeta


def bn_stats(x, eps):
    v = x.reshape(-1, x.shape[-1])
    mean = v.mean(0)
    var = v.var(0, unbiased=False)
    return mean, var, torch.rsqrt(var + eps)


def bn_apply(x, mean, rstd, gamma, beta, act=0):
    u = (x - mean) * rstd * gamma + beta
    return F.elu(u) if act == 2 else u


def bn_bwd(dy, x, mean, rstd, gamma, beta, act=0):
    xh = (x - mean) * rstd
    u = xh * gamma + beta
    d = dy * torch.where(u > 0, torch.ones_like(u), torch.exp(u)) if act == 2 else dy
    red = tuple(range(x.dim() - 1))
    R = x.numel() // x.shape[-1]
    dbeta = d.sum(red)
    dgamma = (d * xh).sum(red)
    dx = gamma * rstd * (d - dbeta / R - xh * dgamma / R)
    return dx, dgamma, dbeta


def softmax_fwd(x, scale=1.0):
    return torch.softmax(x * scale, dim=-1)


def softmax_bwd(dy, y, scale=1.0):
    return scale * y * (dy - (dy * y).sum(-1, keepdim=True))


def colreduce(x, y=None, mode=0, alpha=1.0):
    v = x if mode == 0 else (x * y if mode == 1 else x * x)
    return alpha * v.reshape(-1, x.shape[-1]).sum(0)


def mul2(x, y1, y2):
    return x * y1, x * y2


def segreduce(x, rows_per_seg, alpha=1.0):
    C = x.shape[-1]
    return alpha * x.reshape(-1, rows_per_seg, C).sum(1)


def sincos_fwd(p, want_d=True):
    a = torch.sin(2 * math.pi * p) / (2 * math.pi)
    return a, (torch.cos(2 * math.pi * p) if want_d else None)


def sincos_bwd(da, dd, a, d):
    g = torch.zeros_like(a)
    if da is not None:
        g = g + da * d
    if dd is not None:
        g = g - dd * (4 * math.pi ** 2) * a
    return g


ACT_SILU, ACT_ELU = 1, 2


def act_fwd(x, act):
    return F.silu(x) if act == 1 else F.elu(x)


def act_bwd(dy, x, act):
    if act == 1:
        s = torch.sigmoid(x)
        return dy * s * (1 + x * (1 - s))
    return dy * torch.where(x > 0, torch.ones_like(x), torch.exp(x))


def axpby(x, y, a, b, c=0.0):
    return a * x + (b * y if y is not None else 0) + c


def mul(x, y, alpha=1.0):
    return alpha * x * y


def scale_rows(x, s):
    return x * s.reshape(-1, *([1] * (x.dim() - 1)))


def rowreduce(x, y=None, mode=0, alpha=1.0):
    B = x.shape[0]
    v = x * y if mode == 0 else (x * x if mode == 1 else x)
    return alpha * v.reshape(B, -1).sum(1)


def upsample2x(x, alpha=1.0):
    return (alpha * x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)).contiguous()


def downsample2x(x, alpha=0.25):
    N, H, W, C = x.shape
    return alpha * x.reshape(N, H // 2, 2, W // 2, 2, C).sum(dim=(2, 4))


def squeeze2x(x):
    N, H, W, C = x.shape
    v = x.reshape(N, H // 2, 2, W // 2, 2, C)  # n h dy w dx c
    return v.permute(0, 1, 3, 5, 2, 4).reshape(N, H // 2, W // 2, 4 * C).contiguous()


def unsqueeze2x(x):
    N, H, W, C4 = x.shape
    C = C4 // 4
    v = x.reshape(N, H, W, C, 2, 2)  # n h w c dy dx
    return v.permute(0, 1, 4, 2, 5, 3).reshape(N, 2 * H, 2 * W, C).contiguous()


def nchw_to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nhwc_to_nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def lop_normalize_fwd(w, coeff):
    Co = w.shape[-1]
    l1 = w.abs().reshape(-1, Co).sum(0)
    return w / torch.clamp(l1 / coeff, min=1.0), l1


def lop_normalize_bwd(dwn, w, l1, coeff):
    Co = w.shape[-1]
    s = l1 / coeff
    f = torch.clamp(s, min=1.0)
    dot = (dwn * w).reshape(-1, Co).sum(0)
    g = dwn / f
    corr = torch.sign(w) * dot / (f * f * coeff)
    return g - torch.where(s > 1, corr, torch.zeros_like(corr))


def multi_copy(srcs, dsts):
    for s_, d in zip(srcs, dsts):
        d.copy_(s_.reshape(d.shape))


def lop_normalize_multi_fwd(ws, coeff):
    outs = [lop_normalize_fwd(w, coeff) for w in ws]
    return [o[0] for o in outs], [o[1] for o in outs]


def lop_normalize_multi_bwd(dwns, ws, l1s, coeff):
    return [lop_normalize_bwd(d, w, l, coeff) for d, w, l in zip(dwns, ws, l1s)]


def _b(v, x):
    return v.reshape(-1, *([1] * (x.dim() - 1)))


def sde_perturb(x, noise, mc, sd):
    out = _b(mc, x) * x
    if noise is not None:
        out = out + _b(sd, x) * noise
    return out


def _dsm_e(out, z, std, kind, mode):
    sd = _b(std, out)
    sc = -out / sd if kind == 0 else out
    return (sc * sd + z) if mode == 0 else (sc + z / sd)


def dsm_loss_fwd(out, z, std, w, kind, mode, reduce_mean):
    e = _dsm_e(out, z, std, kind, mode)
    B = out.shape[0]
    r = (e * e).reshape(B, -1)
    return w * (r.mean(1) if reduce_mean else r.sum(1))


def dsm_loss_bwd(gl, out, z, std, w, kind, mode, reduce_mean):
    e = _dsm_e(out, z, std, kind, mode)
    B = out.shape[0]
    D = out.numel() // B
    sd = _b(std, out)
    dsc = -1.0 / sd if kind == 0 else torch.ones_like(sd)
    de = dsc * sd if mode == 0 else dsc
    rs = 1.0 / D if reduce_mean else 1.0
    return _b(gl * w, out) * rs * 2 * e * de


def affine_update(x, s, z, a, b, c, want_mean=True):
    m = _b(a, x) * x
    if s is not None:
        m = m + _b(b, x) * s
    xn = m + _b(c, x) * z if z is not None else m
    return xn, (m if want_mean else None)


def langevin_step(grad_sq, noise_sq, alpha, snr):
    gn = grad_sq.sqrt().mean()
    nn_ = noise_sq.sqrt().mean()
    step = (snr * nn_ / gn) ** 2 * 2 * alpha
    return step, torch.sqrt(step * 2)


def sqnorm(g):
    return (g.double() ** 2).sum().float().reshape(1)


def _need_z(z, philox):
    assert philox is None, "the emulation has no device Philox state: pass the noise tensor"
    return z


def pc_step_em(x, net_out, f_lin, g, k, dt, probability_flow=False, z=None, philox=None, want_mean=True):
    """sampling.py:191-197 with RSDE.sde sde_lib.py:96-103"""
    z = _need_z(z, philox)
    score = _b(k, x) * net_out
    drift = _b(f_lin, x) * x - _b(g, x) ** 2 * score * (0.5 if probability_flow else 1.0)
    m = x + drift * dt
    xn = m if probability_flow else m + _b(g, x) * math.sqrt(-dt) * z
    return xn, (m if want_mean else None)


def pc_step_rd(x, net_out, f_lin, G, k, probability_flow=False, z=None, philox=None, want_mean=True):
    """sampling.py:205-210 with RSDE.discretize sde_lib.py:105-118"""
    z = _need_z(z, philox)
    rev_f = _b(f_lin, x) * x - _b(G, x) ** 2 * (_b(k, x) * net_out) * (0.5 if probability_flow else 1.0)
    m = x - rev_f
    xn = m if probability_flow else m + _b(G, x) * z
    return xn, (m if want_mean else None)


def pc_step_langevin(x, net_out, k, alpha, snr, z=None, philox=None):
    """sampling.py:279-290"""
    z = _need_z(z, philox)
    grad = _b(k, x) * net_out
    B = x.shape[0]
    gn = grad.reshape(B, -1).norm(dim=-1).mean()
    nn_ = z.reshape(B, -1).norm(dim=-1).mean()
    step = (snr * nn_ / gn) ** 2 * 2 * alpha
    m = x + _b(step, x) * grad
    return m + torch.sqrt(_b(step, x) * 2) * z, m


def hutch_div(grad_fn_eps, eps):
    """likelihood.py:27-38"""
    return (grad_fn_eps * eps).reshape(eps.shape[0], -1).sum(1)


def adamw_ema(p, g, m, v, ema, lr, beta1, beta2, eps, wd, step, clip, gnorm_sq, gscale, ema_one_minus_decay):
    coef = gscale
    if clip >= 0 and gnorm_sq is not None:
        nrm = gnorm_sq.sqrt().item() * gscale
        coef = coef * min(1.0, clip / (nrm + 1e-6))
    gg = g * coef
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(gg, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p.addcdiv_(m, denom, value=-lr / bc1)
    if ema is not None:
        ema.sub_(ema_one_minus_decay * (ema - p))


def fixed_point_check(x, xp, y, atol, rtol):
    tol = atol + y.abs() * rtol
    return (~((x - xp) ** 2 / tol < 1)).sum().float().reshape(1)


def ema_update(s, p, one_minus_decay):
    s.sub_(one_minus_decay * (s - p))


def slogdet_inv(w):
    return torch.slogdet(w)[1].reshape(1), torch.linalg.inv(w)


def upfirdn2d(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, layout):
    if layout == 1:
        x = x.permute(0, 3, 1, 2)
    N, C, H, W = x.shape
    kh, kw = kernel.shape
    v = x.reshape(N * C, 1, H, 1, W, 1)
    v = F.pad(v, [0, up_x - 1, 0, 0, 0, up_y - 1])
    v = v.reshape(N * C, 1, H * up_y, W * up_x)
    v = F.pad(v, [max(pad_x0, 0), max(pad_x1, 0), max(pad_y0, 0), max(pad_y1, 0)])
    v = v[:, :, max(-pad_y0, 0): v.shape[2] - max(-pad_y1, 0), max(-pad_x0, 0): v.shape[3] - max(-pad_x1, 0)]
    v = F.conv2d(v, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(v.dtype))
    v = v[:, :, ::down_y, ::down_x]
    v = v.reshape(N, C, v.shape[2], v.shape[3])
    if layout == 1:
        v = v.permute(0, 2, 3, 1)
    return v.contiguous()


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    v = x
    if bias is not None and bias.numel() > 0:
        v = v + bias.reshape(1, -1, *([1] * (x.dim() - 2)))
    if act == 1:
        o = v if grad < 2 else torch.zeros_like(v)
    else:
        if grad == 0:
            o = torch.where(v > 0, v, v * alpha)
        elif grad == 1:
            o = torch.where(ref > 0, v, v * alpha)
        else:
            o = torch.zeros_like(v)
    return o * scale


def split_cols(x, sizes, inverse_of=None):
    sizes = [int(v) for v in sizes]
    if inverse_of is None:
        return torch.cat([t.reshape(-1) for t in x.split(sizes, dim=1)])
    R = int(inverse_of)
    out, o = [], 0
    for c in sizes:
        out.append(x[o:o + R * c].view(R, c))
        o += R * c
    return torch.cat(out, dim=1)


def weightnorm_fwd(v, g):
    n = v.norm(dim=1)
    return v * (g.reshape(-1) / n)[:, None], n


def weightnorm_bwd(dw, v, g, n):
    g = g.reshape(-1)
    dot = (dw * v).sum(1)
    return dw * (g / n)[:, None] - v * (dot * g / n ** 3)[:, None], dot / n


def prepack_weights(ws):
    return None


def pack_w2t_bf16(w2ts):
    return None
