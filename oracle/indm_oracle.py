"""CPU ORACLE for the INDM hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU, fp32, NCHW, eager autograd) restatement of the reference's algorithm for
one INDM training / NELBO / sampling step, written from the reference's behaviour and keyed by
the reference's state-dict names.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; the product path (indm_amd/) never does.

Parity status: PINNED -- tests/test_oracle_golden.py checks every function here against golden
vectors captured by importing the reference itself in the build container
(tests/golden/make_golden.py).  The reference ships no tests of its own.

Each function cites the reference file:line it restates (paths relative to the reference root).
Randomness is drawn through an `Rng` object in exactly the reference's call order
 so recorded noise can be replayed.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- randomness
class Rng:
    """Noise source. Default: torch / numpy global generators, like the reference."""

    def randn(self, *shape):
        return torch.randn(*shape)

    def randn_like(self, x):
        return torch.randn_like(x)

    def rand(self, *shape):
        return torch.rand(*shape)

    def rademacher_like(self, x):  # likelihood.py:85, 205
        return torch.randint_like(x, low=0, high=2).float() * 2 - 1.0

    def poisson(self, lam):  # iresblock.py:305-306
        return int(np.random.poisson(lam, 1)[0])

    def np_rand(self):  # sde_lib.py:208-215: the sampled truncation time of training.st
        return float(np.random.rand())


class ReplayRng(Rng):
    """Feeds recorded draws back in order; shape-checked."""

    def __init__(self, events):
        self.events = list(events)
        self.i = 0

    def _next(self, kind, shape=None):
        assert self.i < len(self.events), f"replay exhausted at {kind}"
        k, v = self.events[self.i]
        self.i += 1
        assert k == kind, f"replay event {self.i - 1}: recorded {k}, requested {kind}"
        if shape is not None:
            assert tuple(v.shape) == tuple(shape), (k, tuple(v.shape), tuple(shape))
        return v

    def randn(self, *shape):
        return torch.as_tensor(self._next("randn", shape)).clone()

    def randn_like(self, x):
        return torch.as_tensor(self._next("randn", x.shape)).clone()

    def rand(self, *shape):
        return torch.as_tensor(self._next("rand", shape)).clone()

    def rademacher_like(self, x):
        return torch.as_tensor(self._next("rademacher", x.shape)).clone()

    def poisson(self, lam):
        return int(self._next("poisson"))

    def np_rand(self):
        return float(self._next("np_rand"))

    def done(self):
        return self.i == len(self.events)


# ----------------------------------------------------------------------------- SDEs
class VPSDE:
    """sde_lib.py:123-215"""

    def __init__(self, truncation_time=1e-5, beta_min=0.1, beta_max=20.0, N=1000):
        self.beta_0, self.beta_1, self.eps, self.N = beta_min, beta_max, truncation_time, N
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1.0 - self.discrete_betas
        self.T = 1

    def sde(self, x, t):  # :147-151
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        return -0.5 * beta_t[:, None, None, None] * x, torch.sqrt(beta_t)

    def marginal_prob(self, x, t):  # :153-157
        lmc = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.exp(lmc[:, None, None, None]) * x, torch.sqrt(1.0 - torch.exp(2.0 * lmc))

    def prior_sampling(self, shape, rng, data_mean=None):  # :159-163
        return rng.randn(*shape) + (0.0 if data_mean is None else data_mean)

    def prior_logp(self, z):  # :165-169
        n = np.prod(z.shape[1:])
        return -n / 2.0 * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.0

    def discretize(self, x, t, next_t=None):  # :171-184
        if next_t is None:
            ts = (t * (self.N - 1) / self.T).long()
            beta, alpha = self.discrete_betas[ts], self.alphas[ts]
            return torch.sqrt(alpha)[:, None, None, None] * x - x, torch.sqrt(beta)
        G = torch.sqrt((t - next_t) * (self.beta_0 + (self.beta_1 - self.beta_0) * t))
        return torch.sqrt(1.0 - G ** 2)[:, None, None, None] * x - x, G

    def integral_beta(self, t):  # :186-187
        return 0.5 * t ** 2 * (self.beta_1 - self.beta_0) + t * self.beta_0

    def antiderivative(self, t, stabilizing_constant=0.0):  # :189-192
        if isinstance(t, (float, int)):
            t = torch.tensor(t).float()
        return torch.log(1.0 - torch.exp(-self.integral_beta(t)) + stabilizing_constant) + self.integral_beta(t)

    def normalizing_constant(self, t_min):  # :194-195
        return self.antiderivative(self.T) - self.antiderivative(t_min)

    def get_diffusion_time(self, batch_size, t_min, importance_sampling, rng):  # :197-206
        if importance_sampling:
            Z = self.normalizing_constant(t_min)
            u = rng.rand(batch_size)
            t = (-self.beta_0 + torch.sqrt(self.beta_0 ** 2 + 2 * (self.beta_1 - self.beta_0) * torch.log(
                1.0 + torch.exp(Z * u + self.antiderivative(t_min))))) / (self.beta_1 - self.beta_0)
            return t, Z
        return rng.rand(batch_size) * (self.T - t_min) + t_min, 1


class VESDE:
    """sde_lib.py:257-350"""

    def __init__(self, truncation_time=1e-5, sigma_min=0.01, sigma_max=50.0, N=1000):
        self.sigma_min, self.sigma_max, self.eps, self.N = sigma_min, sigma_max, truncation_time, N
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), N))
        self.T = 1

    def sde(self, x, t):  # :277-282
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        return torch.zeros_like(x), sigma * torch.sqrt(
            torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min))))

    def marginal_prob(self, x, t):  # :284-287
        return x, self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def prior_sampling(self, shape, rng, data_mean=None):  # :289-293
        return rng.randn(*shape) * self.sigma_max + (0.0 if data_mean is None else data_mean)

    def prior_logp(self, z):  # :295-298
        n = np.prod(z.shape[1:])
        return -n / 2.0 * np.log(2 * np.pi * self.sigma_max ** 2) - torch.sum(z ** 2, dim=(1, 2, 3)) / (
            2 * self.sigma_max ** 2)

    def discretize(self, x, t, next_t=None):  # :310-323
        if next_t is None:
            ts = (t * (self.N - 1) / self.T).long()
            sigma = self.discrete_sigmas[ts]
            adj = torch.where(ts == 0, torch.zeros_like(t), self.discrete_sigmas[ts - 1])
            return torch.zeros_like(x), torch.sqrt(sigma ** 2 - adj ** 2)
        _, s0 = self.marginal_prob(x, t)
        _, s1 = self.marginal_prob(x, next_t)
        return torch.zeros_like(x), torch.sqrt(s0 ** 2 - s1 ** 2)

    def antiderivative(self, t):  # :325-328
        if isinstance(t, (float, int)):
            t = torch.tensor(t).float()
        return 2.0 * torch.log(self.sigma_min * (self.sigma_max / self.sigma_min) ** t)

    def normalizing_constant(self, t_min):
        return self.antiderivative(self.T) - self.antiderivative(t_min)

    def get_diffusion_time(self, batch_size, t_min, importance_sampling, rng):  # :333-341
        if importance_sampling:
            Z = self.normalizing_constant(t_min)
            u = rng.rand(batch_size)
            return t_min + ((Z * u) / (2.0 * (np.log(self.sigma_max) - np.log(self.sigma_min)))), Z
        return rng.rand(batch_size) * (self.T - t_min) + t_min, 1


def get_sde(config):  # sde_lib.py:469-481
    name = config.training.sde.lower()
    if name == "vpsde":
        return VPSDE(config.training.truncation_time, config.model.beta_min, config.model.beta_max,
                     config.model.num_scales)
    if name == "vesde":
        return VESDE(config.training.truncation_time, config.model.sigma_min, config.model.sigma_max,
                     config.model.num_scales)
    raise NotImplementedError(name)


def reverse_sde(sde, score_fn, probability_flow=False):
    """sde_lib.py:74-120 -> (rsde_sde(x,t), rsde_discretize(x,t,next_t))"""
    pf = 0.5 if probability_flow else 1.0

    def rsde(x, t):  # :96-103
        drift, diffusion = sde.sde(x, t)
        drift = drift - diffusion[:, None, None, None] ** 2 * score_fn(x, t) * pf
        return drift, (0.0 if probability_flow else diffusion)

    def rdisc(x, t, next_t=None):  # :105-118
        if next_t is None or next_t[0].item() > 0:
            f, G = sde.discretize(x, t, next_t)
        else:
            f = torch.zeros(x.shape)
            _, G = sde.sde(x, t)
            G = G * torch.sqrt(t - next_t)
        rev_f = f - G[:, None, None, None] ** 2 * score_fn(x, t) * pf
        return rev_f, (torch.zeros_like(G) if probability_flow else G)

    return rsde, rdisc


# ----------------------------------------------------------------------------- native ops (CPU fallbacks = oracles)
def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """op/upfirdn2d.py:159-200 (upfirdn2d_native) with up_x=up_y, pad_x=pad_y as called at :146-150."""
    return upfirdn2d_native(x, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])


def upfirdn2d_native(x, kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    n, c, in_h, in_w = x.shape
    kh, kw = kernel.shape
    out = x.reshape(-1, in_h, 1, in_w, 1, 1)
    out = F.pad(out, [0, 0, 0, up_x - 1, 0, 0, 0, up_y - 1])
    out = out.reshape(-1, in_h * up_y, in_w * up_x, 1)
    out = F.pad(out, [0, 0, max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    out = out[:, max(-py0, 0): out.shape[1] - max(-py1, 0), max(-px0, 0): out.shape[2] - max(-px1, 0), :]
    out = out.permute(0, 3, 1, 2)
    out = F.conv2d(out, torch.flip(kernel, [0, 1]).view(1, 1, kh, kw))
    out = out.permute(0, 2, 3, 1)[:, ::down_y, ::down_x, :]
    return out.reshape(n, c, out.shape[1], out.shape[2])


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    """op/fused_act.py:87-94 (CPU branch; NOTE it hard-codes slope 0.2) vs the CUDA kernel
    (fused_bias_act_kernel.cu:42) which honours alpha -- both are stated here."""
    rest = [1] * (x.ndim - bias.ndim - 1)
    return F.leaky_relu(x + bias.view(1, bias.shape[0], *rest), negative_slope=negative_slope) * scale


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    """op/fused_bias_act_kernel.cu:18-49 behaviour table."""
    v = x
    if bias is not None and bias.numel():
        v = v + bias.view(1, -1, *([1] * (x.ndim - 2)))
    if act == 1:
        out = v if grad < 2 else torch.zeros_like(v)
    elif act == 3:
        if grad == 0:
            out = torch.where(v > 0, v, v * alpha)
        elif grad == 1:
            out = torch.where(ref > 0, v, v * alpha)
        else:
            out = torch.zeros_like(v)
    else:
        raise ValueError(act)
    return out * scale


def _setup_kernel(k):  # up_or_down_sampling.py:181-188
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k /= np.sum(k)
    return k


def upsample_2d(x, k, factor=2, gain=1):  # up_or_down_sampling.py:195-224
    k = _setup_kernel(k) * (gain * factor ** 2)
    p = k.shape[0] - factor
    return upfirdn2d(x, torch.tensor(k), up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))


def downsample_2d(x, k, factor=2, gain=1):  # :227-257
    k = _setup_kernel(k) * gain
    p = k.shape[0] - factor
    return upfirdn2d(x, torch.tensor(k), down=factor, pad=((p + 1) // 2, p // 2))


def conv_downsample_2d(x, w, k, factor=2, gain=1):  # :144-178
    conv_w = w.shape[-1]
    k = _setup_kernel(k) * gain
    p = (k.shape[0] - factor) + (conv_w - 1)
    x = upfirdn2d(x, torch.tensor(k), pad=((p + 1) // 2, p // 2))
    return F.conv2d(x, w, stride=factor, padding=0)


def naive_upsample_2d(x, factor=2):  # :59-63
    n, c, h, w = x.shape
    return x.reshape(-1, c, h, 1, w, 1).repeat(1, 1, 1, factor, 1, factor).reshape(-1, c, h * factor, w * factor)


def naive_downsample_2d(x, factor=2):  # :66-69
    n, c, h, w = x.shape
    return torch.mean(x.reshape(-1, c, h // factor, factor, w // factor, factor), dim=(3, 5))


# ----------------------------------------------------------------------------- NCSN++ (models/ncsnpp.py)
def get_timestep_embedding(timesteps, dim, max_positions=10000):  # models/layers.py:515-529
    half = dim // 2
    emb = math.log(max_positions) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    emb = timesteps.float()[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


def _gn(P, pre, x):
    c = x.shape[1]
    return F.group_norm(x, min(c // 4, 32), P[pre + ".weight"], P[pre + ".bias"], eps=1e-6)


def _conv(P, pre, x, stride=1, padding=1):
    return F.conv2d(x, P[pre + ".weight"], P[pre + ".bias"], stride=stride, padding=padding)


def _nin(P, pre, x):  # models/layers.py:546-556
    y = torch.einsum("bhwc,cd->bhwd", x.permute(0, 2, 3, 1), P[pre + ".W"]) + P[pre + ".b"]
    return y.permute(0, 3, 1, 2)


def resnet_block_biggan(P, pre, cfg, x, temb, up=False, down=False, train=False, rng=None):
    """models/layerspp.py:225-287"""
    fir, fk = cfg.model.fir, cfg.model.fir_kernel
    out_ch = P[pre + ".Conv_0.weight"].shape[0]
    in_ch = x.shape[1]
    h = F.silu(_gn(P, pre + ".GroupNorm_0", x))
    if up:
        if fir:
            h, x = upsample_2d(h, fk, 2), upsample_2d(x, fk, 2)
        else:
            h, x = naive_upsample_2d(h, 2), naive_upsample_2d(x, 2)
    elif down:
        if fir:
            h, x = downsample_2d(h, fk, 2), downsample_2d(x, fk, 2)
        else:
            h, x = naive_downsample_2d(h, 2), naive_downsample_2d(x, 2)
    h = _conv(P, pre + ".Conv_0", h)
    if temb is not None:
        h = h + F.linear(F.silu(temb), P[pre + ".Dense_0.weight"], P[pre + ".Dense_0.bias"])[:, :, None, None]
    h = F.silu(_gn(P, pre + ".GroupNorm_1", h))
    if train and cfg.model.dropout > 0:
        h = F.dropout(h, cfg.model.dropout, True)
    h = _conv(P, pre + ".Conv_1", h)
    if in_ch != out_ch or up or down:
        x = _conv(P, pre + ".Conv_2", x, padding=0)
    return (x + h) / np.sqrt(2.0) if cfg.model.skip_rescale else x + h


def attn_block(P, pre, cfg, x):
    """models/layerspp.py:75-104"""
    B, C, H, W = x.shape
    h = _gn(P, pre + ".GroupNorm_0", x)
    q, k, v = _nin(P, pre + ".NIN_0", h), _nin(P, pre + ".NIN_1", h), _nin(P, pre + ".NIN_2", h)
    w = torch.einsum("bchw,bcij->bhwij", q, k) * (int(C) ** (-0.5))
    w = F.softmax(w.reshape(B, H, W, H * W), dim=-1).reshape(B, H, W, H, W)
    h = torch.einsum("bhwij,bcij->bchw", w, v)
    h = _nin(P, pre + ".NIN_3", h)
    return (x + h) / np.sqrt(2.0) if cfg.model.skip_rescale else x + h


def pyramid_downsample(P, pre, cfg, x):
    """layerspp.Downsample(with_conv=True) models/layerspp.py:142-176 -> up_or_down_sampling.Conv2d(down=True) :23-56"""
    if cfg.model.fir:
        y = conv_downsample_2d(x, P[pre + ".Conv2d_0.weight"], cfg.model.fir_kernel)
        return y + P[pre + ".Conv2d_0.bias"].reshape(1, -1, 1, 1)
    x = F.pad(x, (0, 1, 0, 1))
    return _conv(P, pre + ".Conv_0", x, stride=2, padding=0)


def ncsnpp_forward(P, cfg, x, time_cond, train=False, rng=None):
    """models/ncsnpp.py:251-414 for resblock_type='biggan', progressive='none',
    progressive_input in {'none','residual'}, conditional=True.  P: state dict without 'module.'."""
    m = cfg.model
    nf, nrb, nres = m.nf, m.num_res_blocks, len(m.ch_mult)
    pre = "all_modules."
    i = 0
    if m.embedding_type == "fourier":  # :257-261; layerspp.py:45-54
        used_sigmas = time_cond
        xp = torch.log(used_sigmas)[:, None] * P[pre + "0.W"][None, :] * 2 * np.pi
        temb = torch.cat([torch.sin(xp), torch.cos(xp)], dim=-1)
        i += 1
    else:  # :263-267
        used_sigmas = P["sigmas"][time_cond.long()]
        temb = get_timestep_embedding(time_cond, nf)
    temb = F.linear(temb, P[f"{pre}{i}.weight"], P[f"{pre}{i}.bias"]); i += 1
    temb = F.linear(F.silu(temb), P[f"{pre}{i}.weight"], P[f"{pre}{i}.bias"]); i += 1
    if not cfg.data.centered:
        x = 2 * x - 1.0
    input_pyramid = x if m.progressive_input != "none" else None
    hs = [_conv(P, f"{pre}{i}", x)]; i += 1
    for lvl in range(nres):
        for _ in range(nrb):
            h = resnet_block_biggan(P, f"{pre}{i}", cfg, hs[-1], temb, train=train, rng=rng); i += 1
            if h.shape[-1] in m.attn_resolutions and m.attention:
                h = attn_block(P, f"{pre}{i}", cfg, h); i += 1
            hs.append(h)
        if lvl != nres - 1:
            if m.auxiliary_resblock:
                h = resnet_block_biggan(P, f"{pre}{i}", cfg, hs[-1], temb, down=True, train=train, rng=rng); i += 1
            if m.progressive_input == "residual":
                input_pyramid = pyramid_downsample(P, f"{pre}{i}", cfg, input_pyramid); i += 1
                input_pyramid = (input_pyramid + h) / np.sqrt(2.0) if m.skip_rescale else input_pyramid + h
                h = input_pyramid
            if m.auxiliary_resblock:
                hs.append(h)
    h = hs[-1]
    if not m.auxiliary_resblock:
        hs.pop()
    h = resnet_block_biggan(P, f"{pre}{i}", cfg, h, temb, train=train, rng=rng); i += 1
    h = attn_block(P, f"{pre}{i}", cfg, h); i += 1
    h = resnet_block_biggan(P, f"{pre}{i}", cfg, h, temb, train=train, rng=rng); i += 1
    nup = nrb + 1 if m.auxiliary_resblock else nrb
    for lvl in reversed(range(nres)):
        for _ in range(nup):
            h = resnet_block_biggan(P, f"{pre}{i}", cfg, torch.cat([h, hs.pop()], dim=1), temb, train=train, rng=rng)
            i += 1
        if h.shape[-1] in m.attn_resolutions and m.attention:
            h = attn_block(P, f"{pre}{i}", cfg, h); i += 1
        if lvl != 0 and m.auxiliary_resblock:
            h = resnet_block_biggan(P, f"{pre}{i}", cfg, h, temb, up=True, train=train, rng=rng); i += 1
    assert not hs
    h = F.silu(_gn(P, f"{pre}{i}", h)); i += 1
    h = _conv(P, f"{pre}{i}", h); i += 1
    if m.scale_by_sigma:
        h = h / used_sigmas.reshape(x.shape[0], 1, 1, 1)
    return h


def get_score_fn(P, cfg, sde, train=False, rng=None):
    """models/utils.py:140-197 (continuous=True branches)"""
    if isinstance(sde, VPSDE):
        def score_fn(x, t):
            out = ncsnpp_forward(P, cfg, x, t * 999, train=train, rng=rng)
            std = sde.marginal_prob(torch.zeros_like(x), t)[1]
            return -out / std[:, None, None, None] if cfg.training.ddpm_score else out
    else:
        def score_fn(x, t):
            labels = sde.marginal_prob(torch.zeros_like(x), t)[1]
            return ncsnpp_forward(P, cfg, x, labels, train=train, rng=rng)
    return score_fn


# ----------------------------------------------------------------------------- wolf flow
def squeeze(x, f=2):  # flows/resflow/layers/squeeze.py:32-45
    b, c, h, w = x.shape
    return x.reshape(b, c, h // f, f, w // f, f).permute(0, 1, 3, 5, 2, 4).reshape(b, c * f * f, h // f, w // f)


def lop_weight(w, coeff=0.98):  # layers/base/lipschitz.py:350-359 (domain=codomain=inf -> L1 over (ci,kh,kw))
    scale = w.abs().sum(dim=(1, 2, 3), keepdim=True)
    return w / torch.max(torch.ones(1), scale / coeff)


def sin_act(x):  # layers/base/activations.py:7-12
    return torch.sin(2.0 * math.pi * x) / math.pi * 0.5


def poisson_1mcdf(lamb, k, offset):  # iresblock.py:309-318
    if k <= offset:
        return 1.0
    k = k - offset
    s = 1.0
    for i in range(1, k):
        s += lamb ** i / math.factorial(i)
    return 1 - np.exp(-lamb) * s


def _iresblock_keys(P, pre):
    """nnet indices: [Sin?] Lop(3x3) Sin LopCond(1x1) Sin Lop(3x3)  (resflow_.py:432-479)"""
    idx = sorted({int(k[len(pre + ".nnet."):].split(".")[0]) for k in P if k.startswith(pre + ".nnet.")})
    return idx  # three conv indices; leading Sin present iff idx[0] == 1


def iresblock_g(P, pre, x, h):
    """nnet_forward iresblock.py:55-61; LopCondConv2d.forward lipschitz.py:429-435"""
    i0, i1, i2 = _iresblock_keys(P, pre)
    n = pre + ".nnet."
    if i0 == 1:
        x = sin_act(x)
    x = F.conv2d(x, lop_weight(P[f"{n}{i0}.weight"]), P[f"{n}{i0}.bias"], 1, 1)
    x = sin_act(x)
    c = F.linear(h, P[f"{n}{i1}.h_net.net.weight"], P[f"{n}{i1}.h_net.net.bias"])
    x = x + c[:, :, None, None]
    x = F.conv2d(x, lop_weight(P[f"{n}{i1}.weight"]), P[f"{n}{i1}.bias"], 1, 0)
    x = sin_act(x)
    return F.conv2d(x, lop_weight(P[f"{n}{i2}.weight"]), P[f"{n}{i2}.bias"], 1, 1)


def iresblock_logdet(P, pre, x, h, training, rng, stats=None):
    """_logdetgrad iresblock.py:90-174 (poisson, n_samples=1, exact_trace=False, n_power_series=None)."""
    lamb = float(P[pre + ".lamb"].detach())
    n = rng.poisson(lamb)
    n_exact = 2 if training else 20  # :118, :129
    K = n + n_exact

    def coeff(k):  # :120-121 / :131-132 ; the indicator is 1 for k <= K with n_samples=1
        return 1.0 / poisson_1mcdf(lamb, k, n_exact) * (1.0 if n >= k - n_exact else 0.0)

    vareps = rng.randn_like(x)
    with torch.enable_grad():
        if not x.requires_grad:
            x = x.requires_grad_(True)
        g = iresblock_g(P, pre, x, h)
        if training:  # neumann_logdet_estimator :264-273
            vjp, nvjp = vareps, vareps
            with torch.no_grad():
                for k in range(1, K + 1):
                    vjp = torch.autograd.grad(g, x, vjp, retain_graph=True)[0]
                    nvjp = nvjp + (-1) ** k * coeff(k) * vjp
            vjp_jac = torch.autograd.grad(g, x, nvjp, create_graph=True)[0]
            logdet = torch.sum(vjp_jac.reshape(x.shape[0], -1) * vareps.reshape(x.shape[0], -1), 1)
        else:  # basic_logdet_estimator :253-261
            vjp = vareps
            logdet = torch.tensor(0.0)
            for k in range(1, K + 1):
                vjp = torch.autograd.grad(g, x, vjp, create_graph=False, retain_graph=True)[0]
                tr = torch.sum(vjp.reshape(x.shape[0], -1) * vareps.reshape(x.shape[0], -1), 1)
                logdet = logdet + (-1) ** (k + 1) / k * coeff(k) * tr
    if stats is not None:
        stats.append((pre, n, logdet.detach()))
    return g, logdet.view(-1, 1)


def iresblock_inverse(P, pre, y, h, atol=1e-5, rtol=1e-5, iters=None):
    """_inverse_fixed_point iresblock.py:78-88"""
    x, x_prev = y - iresblock_g(P, pre, y, h), y
    i = 0
    tol = atol + y.abs() * rtol
    while not torch.all((x - x_prev) ** 2 / tol < 1):
        x, x_prev = y - iresblock_g(P, pre, x, h), x
        i += 1
        if i > 1000:
            break
    if iters is not None:
        iters.append(i)
    return x


def _flow_blocks(P):
    pre = "generator.flow.transforms."
    scales = sorted({int(k[len(pre):].split(".")[0]) for k in P if k.startswith(pre)})
    out = []
    for s in scales:
        sp = f"{pre}{s}.chain."
        blocks = sorted({int(k[len(sp):].split(".")[0]) for k in P if k.startswith(sp)})
        out.append([f"{sp}{b}" for b in blocks])
    return out


def residual_flow_fwd(P, x, h, eval_logdet, training, rng, stats=None):
    """ResidualFlow.fwdpass resflow_.py:310-324 (+ forward :205-235, StackediResBlocks chain, SqueezeLayer
    after every scale but the last :502)."""
    shape = x.shape
    scales = _flow_blocks(P)
    logpx = 0.0
    for si, blocks in enumerate(scales):
        for pre in blocks:
            if eval_logdet:
                g, ld = iresblock_logdet(P, pre, x, h, training, rng, stats)
                x, logpx = x + g, logpx - ld
            else:
                x = x + iresblock_g(P, pre, x, h)
        if si < len(scales) - 1:
            x = squeeze(x, 2)
    out = x.reshape(shape[0], -1)
    if len(scales) > 1:  # '-' in nblocks :313-315
        out = out.view(shape[0], shape[1], 2, 2, shape[2] // 2, shape[3] // 2).permute(0, 1, 4, 2, 5, 3).reshape(shape)
    else:
        out = out.view(shape)
    return (out, logpx.squeeze(-1)) if eval_logdet else out


def residual_flow_bwd(P, x, h, iters=None):
    """ResidualFlow.bwdpass resflow_.py:326-335 + inverse :237-267"""
    s = x.shape
    scales = _flow_blocks(P)
    if len(scales) > 1:
        x = x.view(s[0], s[1], s[2] // 2, s[3] // 2, 2, 2).permute(0, 1, 5, 2, 3, 4).reshape(
            s[0], s[1], s[2], 2, s[3] // 2).permute(0, 1, 3, 2, 4).reshape(s)
    k = len(scales) - 1
    z = x.reshape(s[0], s[1] * 4 ** k, s[2] // 2 ** k, s[3] // 2 ** k)
    for si in range(len(scales) - 1, -1, -1):
        if si < len(scales) - 1:
            z = torch.pixel_shuffle(z, 2)  # SqueezeLayer.inverse
        for pre in reversed(scales[si]):
            z = iresblock_inverse(P, pre, z, h, iters=iters)
    return z


def _bn(P, pre, x, training, bn_updates):
    """nn.BatchNorm2d in train mode uses batch statistics and updates running stats (momentum 0.1)."""
    rm, rv = P[pre + ".running_mean"], P[pre + ".running_var"]
    if training:
        if bn_updates is not None:
            with torch.no_grad():
                n = x.numel() / x.shape[1]
                mean = x.mean(dim=(0, 2, 3))
                var = x.var(dim=(0, 2, 3), unbiased=False)
                bn_updates[pre + ".running_mean"] = 0.9 * rm + 0.1 * mean
                bn_updates[pre + ".running_var"] = 0.9 * rv + 0.1 * var * n / (n - 1)
                bn_updates[pre + ".num_batches_tracked"] = P[pre + ".num_batches_tracked"] + 1
        return F.batch_norm(x, None, None, P[pre + ".weight"], P[pre + ".bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, rm, rv, P[pre + ".weight"], P[pre + ".bias"], False, 0.1, 1e-5)


def encoder_forward(P, x, training, bn_updates=None):
    """GlobalResNetEncoderBatchNorm modules/encoders/global_encoder.py:12-44;
    ResNetBlockBatchNorm nnet/resnets/resnet_batchnorm.py:18-76 (ELU, strides [1,2] per level)."""
    pre = "discriminator.encoder.net."
    lvl = 0
    while f"{pre}resnet{lvl}.main.0.conv1.weight" in P:
        for b, stride in ((0, 1), (1, 2)):
            bp = f"{pre}resnet{lvl}.main.{b}"
            res = x
            out = F.conv2d(x, P[bp + ".conv1.weight"], None, stride, 1)
            out = F.elu(_bn(P, bp + ".bn1", out, training, bn_updates))
            out = F.conv2d(out, P[bp + ".conv2.weight"], None, 1, 1)
            out = _bn(P, bp + ".bn2", out, training, bn_updates)
            if bp + ".downsample.0.weight" in P:
                res = F.conv2d(x, P[bp + ".downsample.0.weight"], None, stride, 0)
                res = _bn(P, bp + ".downsample.1", res, training, bn_updates)
            x = F.elu(out + res)
        lvl += 1
    x = F.elu(F.conv2d(x, P[pre + "top.weight"], P[pre + "top.bias"]))
    return x.reshape(x.shape[0], -1)


def _wn_linear(P, pre, x):
    """nn.utils.weight_norm(Linear): w = g * v / ||v||_row  (nnet/weight_norm.py:8-40)"""
    v, g = P[pre + ".linear.weight_v"], P[pre + ".linear.weight_g"]
    w = v * (g / v.norm(dim=1, keepdim=True))
    return F.linear(x, w, P[pre + ".linear.bias"])


def _nice_mlp(P, pre, z):  # couplings/blocks.py:11-48
    out = F.elu(F.linear(z, P[pre + ".fc1.weight"], P[pre + ".fc1.bias"]))
    out = F.elu(F.linear(out, P[pre + ".fc2.weight"], P[pre + ".fc2.bias"]))
    return _wn_linear(P, pre + ".fc3", out)


def _nice1d(P, pre, x, split_type, up, backward):
    """NICE1d couplings/coupling.py:13-177 + Affine transform.py:49-81 (scale = sigmoid(s+2)+1e-3)."""
    d = x.shape[-1]
    if split_type == "continuous":
        z1f = d - d // 2 if up else d // 2
        z1, z2 = x[..., :z1f], x[..., z1f:]
    else:
        z1, z2 = x[..., 0::2], x[..., 1::2]
    z, zp = (z1, z2) if up else (z2, z1)
    params = _nice_mlp(P, pre + ".net", z)
    mu, ls = params.chunk(2, dim=-1)
    scale = torch.sigmoid(ls + 2.0) + 1e-3
    if not backward:
        zp = scale * zp + mu
        logdet = scale.log().reshape(x.shape[0], -1).sum(1)
    else:
        zp = (zp - mu) / (scale + 1e-12)
        logdet = -scale.log().reshape(x.shape[0], -1).sum(1)
    z1, z2 = (z, zp) if up else (zp, z)
    if split_type == "continuous":
        out = torch.cat([z1, z2], dim=-1)
    else:
        out = torch.stack([z1, z2], dim=-1).reshape(x.shape)
    return out, logdet


def _actnorm1d(P, pre, x, backward):  # flows/normalization.py:13-75 (3-D input -> num = 1)
    ls, b = P[pre + ".log_scale"], P[pre + ".bias"]
    if not backward:
        return x * ls.exp() + b, ls.sum(0, keepdim=True)
    return (x - b) / (ls.exp() + 1e-8), -ls.sum(0, keepdim=True)


def _invlinear(P, pre, x, backward):  # flows/permutation.py:75-136 (slogdet on CPU)
    w = P[pre + ".weight_inv"] if backward else P[pre + ".weight"]
    return F.linear(x, w), torch.slogdet(w)[1]


_UNIT = (("coupling1_up", "continuous", True), ("coupling1_dn", "continuous", False), ("actnorm", None, None),
         ("coupling2_up", "skip", True), ("coupling2_dn", "skip", False))


def prior_flow(P, x, backward):
    """PriorFlow priors/flow.py:166-201 (steps :110-164, unit :16-107). `backward=False` is Flow.forward
    (used by calcKL through bwdpass since inverse=True), `backward=True` is Flow.backward (sampling)."""
    pre = "discriminator.prior.flow.steps."
    nsteps = len({k[len(pre):].split(".")[0] for k in P if k.startswith(pre)})
    logdet = x.new_zeros(x.shape[0])
    order = range(nsteps) if not backward else reversed(range(nsteps))
    for s in order:
        sp = f"{pre}{s}"
        if not backward:
            x, ld = _actnorm1d(P, sp + ".actnorm", x, False); logdet = logdet + ld
            x, ld = _invlinear(P, sp + ".linear", x, False); logdet = logdet + ld
            for name, st, up in _UNIT:
                if st is None:
                    x, ld = _actnorm1d(P, f"{sp}.unit.{name}", x, False)
                else:
                    x, ld = _nice1d(P, f"{sp}.unit.{name}", x, st, up, False)
                logdet = logdet + ld
        else:
            for name, st, up in reversed(_UNIT):
                if st is None:
                    x, ld = _actnorm1d(P, f"{sp}.unit.{name}", x, True)
                else:
                    x, ld = _nice1d(P, f"{sp}.unit.{name}", x, st, up, True)
                logdet = logdet + ld
            x, ld = _invlinear(P, sp + ".linear", x, True); logdet = logdet + ld
            x, ld = _actnorm1d(P, sp + ".actnorm", x, True); logdet = logdet + ld
    return x, logdet


def sampling_and_kl(P, x, training, rng, bn_updates=None):
    """GaussianDiscriminator.sampling_and_KL gaussian.py:67-77 + FlowPrior.calcKL priors/flow.py:232-253"""
    c = _wn_linear(P, "discriminator.fc", encoder_forward(P, x, training, bn_updates))
    mu, logvar = c.chunk(2, dim=1)
    eps = rng.randn(mu.shape[0], 1, mu.shape[1])
    z = eps * logvar.mul(0.5).exp().unsqueeze(1) + mu.unsqueeze(1)
    dim = z.shape[2]
    cc = math.log(math.pi * 2.0)
    log_post = ((logvar.unsqueeze(1) + eps.pow(2)).sum(dim=2) + cc * dim).mean(dim=1) * -0.5
    e, logdet = prior_flow(P, z, backward=False)
    e = e.reshape(e.shape[0], -1)
    log_prior = ((e * e).sum(dim=1) + cc * e.shape[1]) * -0.5 + logdet
    return z, log_post - log_prior


def wolf_forward(P, x, reverse=False, eval_logdet=True, training=False, rng=None, stats=None, bn_updates=None,
                 iters=None):
    """WolfCore.forward wolf.py:81-130"""
    if reverse:
        eps = rng.randn(x.shape[0], 64)  # priors/flow.py:226-230 (fresh prior sample, not the posterior h)
        h, _ = prior_flow(P, eps, backward=True)
        return residual_flow_bwd(P, x, h, iters)
    z, kl = sampling_and_kl(P, x, training, rng, bn_updates)
    h = z.view(-1, z.size(2))
    if eval_logdet:
        e, logdet = residual_flow_fwd(P, x, h, True, training, rng, stats)
        return e, -logdet - kl
    return residual_flow_fwd(P, x, h, False, training, rng)


def flow_forward(P, cfg, x, log_det=0, reverse=False, training=False, rng=None, **kw):
    """flow_models/flow_model.py:53-67 ('wolf' branch)"""
    if cfg.flow.squeeze:
        x = squeeze(x, 2)
    if not reverse:
        if log_det == 0:
            z, ld = wolf_forward(P, x, False, True, training, rng, **kw)
        else:
            z, ld = wolf_forward(P, x, False, False, training, rng, **kw), -1
    else:
        z, ld = wolf_forward(P, x, True, training=training, rng=rng, **kw).view(x.shape), -1
    if cfg.flow.squeeze:
        z = torch.pixel_shuffle(z, 2)
    return z, ld


# ----------------------------------------------------------------------------- losses / step functions
def get_t_min(sde, cfg, st=False, rng=None):
    """SDE.get_t_min sde_lib.py:208-215 (VP) / 343-350 (VE): the truncation time, sampled when training.st is on (the
    reference calls np.random.rand() there: numpy's global generator, which the flow's Poisson draws share)"""
    if st:
        u = (rng or Rng()).np_rand()
        if cfg.training.k == 1.0:
            return sde.eps ** (1.0 - u)
        return sde.eps / (1.0 - u * (1 - sde.eps ** (cfg.training.k - 1))) ** (1.0 / (cfg.training.k - 1))
    return sde.eps


def sde_loss(Ps, cfg, sde, batch, train, rng, recon_loss=False, importance_sampling=None, st=False, variance="scoreflow"):
    """get_sde_loss_fn.loss_fn losses.py:83-142, with its optional reconstruction term (:120-140) and st"""
    if importance_sampling is None:
        importance_sampling = cfg.training.importance_sampling
    reduce_op = torch.mean if cfg.training.reduce_mean else torch.sum
    t_min = get_t_min(sde, cfg, st, rng)
    t, Z = sde.get_diffusion_time(batch.shape[0], t_min, importance_sampling, rng)
    score_fn = get_score_fn(Ps, cfg, sde, train=train, rng=rng)
    z = rng.randn_like(batch)
    mean, std = sde.marginal_prob(batch, t)
    perturbed = mean + std[:, None, None, None] * z
    score = score_fn(perturbed, t)
    if importance_sampling or not cfg.training.likelihood_weighting:
        losses = torch.square(score * std[:, None, None, None] + z)
        losses = 0.5 * Z * reduce_op(losses.reshape(losses.shape[0], -1), dim=-1)
    else:
        g2 = sde.sde(torch.zeros_like(batch), t)[1] ** 2
        losses = torch.square(score + z / std[:, None, None, None])
        losses = 0.5 * Z * reduce_op(losses.reshape(losses.shape[0], -1), dim=-1) * g2
    if recon_loss:  # losses.py:120-140
        eps_vec = torch.ones(batch.shape[0]) * t_min
        mean, std = sde.marginal_prob(batch, eps_vec)
        z = rng.randn_like(batch)
        perturbed = mean + std[:, None, None, None] * z
        score = score_fn(perturbed, eps_vec)
        alpha, beta = sde.marginal_prob(torch.ones_like(batch), eps_vec)
        q_mean = perturbed / alpha + beta[:, None, None, None] ** 2 * score / alpha
        q_std = beta if variance == "ddpm" else beta / torch.mean(alpha, axis=(1, 2, 3))
        n_dim = np.prod(batch.shape[1:])
        p_entropy = n_dim / 2.0 * (np.log(2 * np.pi) + 2 * torch.log(std) + 1.0)
        q_recon = n_dim / 2.0 * (np.log(2 * np.pi) + 2 * torch.log(q_std)) + 0.5 / (q_std ** 2) * torch.square(
            batch - q_mean).sum(axis=(1, 2, 3))
        reconstruction_loss = q_recon - p_entropy
        if cfg.training.reduce_mean:
            reconstruction_loss = reconstruction_loss / n_dim
        losses = losses + reconstruction_loss
    return losses


def calculate_logp(sde, batch, rng):  # losses.py:219-225
    Ts = torch.ones(batch.shape[0]) * sde.T
    meanT, stdT = sde.marginal_prob(batch, Ts)
    z = rng.randn_like(batch)
    return sde.prior_logp(meanT + stdT[:, None, None, None] * z)


def ema_decay(rate, num_updates):  # models/ema.py:43-51 (num_updates already incremented)
    return min(rate, (1 + num_updates) / (10 + num_updates))


class OracleState:
    """state = dict(optimizer, model, ema, step) (utils.py:76,119) for one network, functional weights."""

    def __init__(self, sd, lr, ema_rate, cfg):
        self.P = {k: v.clone() for k, v in sd.items()}
        self.trainable = [k for k, v in self.P.items() if v.is_floating_point() and self._is_param(k)]
        for k in self.trainable:
            self.P[k].requires_grad_(True)
        params = [self.P[k] for k in self.trainable]
        o = cfg.optim
        self.opt = torch.optim.AdamW(params, lr=lr, betas=(o.beta1, 0.99), eps=o.eps, weight_decay=o.weight_decay,
                                     amsgrad=o.amsgrad)  # losses.py:40
        self.ema = [p.detach().clone() for p in params]
        self.ema_rate, self.num_updates, self.step, self.cfg = ema_rate, 0, 0, cfg

    @staticmethod
    def _is_param(k):
        buffers = ("running_mean", "running_var", "num_batches_tracked", "last_n_samples", "last_firmom",
                   "last_secmom", ".scale", "weight_inv", "sigmas")
        if k.endswith(buffers):
            return False
        return not (k.endswith(".W") and "NIN" not in k)  # GaussianFourierProjection.W: requires_grad=False

    def zero_grad(self):
        for k in self.trainable:
            self.P[k].grad = None

    def optimize(self):  # losses.py:51-60 with warmup=0
        params = [self.P[k] for k in self.trainable]
        if self.cfg.optim.grad_clip >= 0:
            torch.nn.utils.clip_grad_norm_(params, max_norm=self.cfg.optim.grad_clip)
        self.opt.step()

    def ema_update(self):  # models/ema.py:32-51
        self.num_updates += 1
        d = ema_decay(self.ema_rate, self.num_updates)
        with torch.no_grad():
            for s, k in zip(self.ema, self.trainable):
                s.sub_((1.0 - d) * (s - self.P[k]))


def _micro_batches(cfg, batch):
    """losses.py:287-288 / :355-356: batch[B//n*k : B//n*(k+1)] for k < n = optim.num_micro_batch"""
    n = int(cfg.optim.num_micro_batch)
    B = batch.shape[0]
    return [batch[B // n * k: B // n * (k + 1)] for k in range(n)]


def _nll_losses(cfg, sde, state, flow_state, mb, rng, stats, importance_sampling=None, st=False, keep=None):
    bn = {}
    z, losses_flow = flow_forward(flow_state.P, cfg, mb, training=True, rng=rng, stats=stats, bn_updates=bn)
    losses_score = sde_loss(state.P, cfg, sde, z, True, rng, importance_sampling=importance_sampling, st=st)
    if keep is not None:   # training.st: the second phase of flow_step_fn_fid re-uses these (losses.py:383-389)
        keep["z"], keep["losses_score"] = z.detach(), losses_score.detach()
    losses_logp = calculate_logp(sde, z, rng)
    D = np.prod(mb.shape[1:])
    if cfg.training.reduce_mean:
        losses_flow, losses_logp = -losses_flow / D, -losses_logp / D
    else:
        losses_flow, losses_logp = -losses_flow, -losses_logp
    losses = losses_score + losses_flow + losses_logp
    torch.mean(losses).backward()      # gradients ACCUMULATE over the micro-batches (sum of per-micro-batch means)
    with torch.no_grad():
        for k, v in bn.items():
            flow_state.P[k] = v
    return losses.detach(), losses_score.detach(), losses_flow.detach(), losses_logp.detach()


def flow_step_fn_nll(cfg, sde, state, flow_state, batch, rng, stats=None):
    """losses.py:258-320 (train=True)"""
    state.zero_grad()
    flow_state.zero_grad()
    parts = [_nll_losses(cfg, sde, state, flow_state, mb, rng, stats, st=cfg.training.st)   # losses.py:294
             for mb in _micro_batches(cfg, batch)]
    state.optimize()
    flow_state.optimize()
    state.step += 1
    state.ema_update()
    flow_state.step += 1
    flow_state.ema_update()
    return tuple(torch.cat([p[i] for p in parts]) for i in range(4))


def flow_step_fn_nll_chunked(cfg, sde, state, flow_state, batch, events, chunk=16):
    """flow_step_fn_nll (losses.py:258-320, num_micro_batch = 1) evaluated in a bounded amount of host memory: the same
    step, image chunks at a time.  Images are independent units of the objective (losses.py:303-304) except through the
    train-mode BatchNorm statistics of the flow's encoder (resnet_batchnorm.py:23,31), so the encoder + prior flow run
    ONCE on the whole batch, the residual flow / score network / prior terms run per chunk on the matching rows of h,
    KL and of every recorded noise draw (`events`: the whole-batch draws in the reference's call order; the Poisson
    series lengths are per block, not per image, and are replayed unchanged for every chunk), and the gradients of
    mean(losses) accumulate over the chunks -- through h and KL back into the encoder at the end.  Pinned against the
    unchunked form on the mini golden configuration (tests/test_oracle_golden.py).  The eager double-backward graph of
    a whole batch-128 step needs ~100 GB of host memory; a 16-image chunk ~13 GB."""
    assert int(cfg.optim.num_micro_batch) == 1
    state.zero_grad()
    flow_state.zero_grad()
    B = batch.shape[0]
    P = flow_state.P
    bn = {}
    x_in = squeeze(batch, 2) if cfg.flow.squeeze else batch
    enc_rng = ReplayRng(events[:1])                       # the posterior noise [B, 1, 64] (gaussian.py:70)
    zh, kl = sampling_and_kl(P, x_in, True, enc_rng, bn)
    assert enc_rng.done()
    h_leaf = zh.view(-1, zh.size(2)).detach().requires_grad_(True)
    kl_leaf = kl.detach().requires_grad_(True)
    rest = events[1:]
    D = np.prod(batch.shape[1:])
    parts = []
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        rng = ReplayRng([(k, v if k == "poisson" else v[lo:hi]) for k, v in rest])
        e, logpx = residual_flow_fwd(P, x_in[lo:hi], h_leaf[lo:hi], True, True, rng)
        z, lf = e, -logpx - kl_leaf[lo:hi]                # wolf_forward returns (eps, -logpx - KL)  (wolf.py:81-130)
        if cfg.flow.squeeze:
            z = torch.pixel_shuffle(z, 2)
        ls = sde_loss(state.P, cfg, sde, z, True, rng)
        lp = calculate_logp(sde, z, rng)
        assert rng.done()
        if cfg.training.reduce_mean:
            lf, lp = -lf / D, -lp / D
        else:
            lf, lp = -lf, -lp
        losses = ls + lf + lp
        (losses.sum() / B).backward()                     # = this chunk's share of torch.mean(losses).backward()
        parts.append((losses.detach(), ls.detach(), lf.detach(), lp.detach()))
    torch.autograd.backward([zh.view(-1, zh.size(2)), kl], [h_leaf.grad, kl_leaf.grad])
    with torch.no_grad():
        for k, v in bn.items():
            P[k] = v
    state.optimize()
    flow_state.optimize()
    state.step += 1
    state.ema_update()
    flow_state.step += 1
    flow_state.ema_update()
    return tuple(torch.cat([p[i] for p in parts]) for i in range(4))


def flow_step_fn_fid_chunked(cfg, sde, state, flow_state, batch, events, chunk=16):
    """flow_step_fn_fid (losses.py:322-406; training.st off, num_micro_batch = 1) in a bounded amount of host memory, the
    two-phase analogue of flow_step_fn_nll_chunked: phase 1 = the NLL objective with importance_sampling=True, chunk by
    chunk, stepping the flow only; phase 2 = the flow's forward without log-det on the UPDATED flow (train mode: the
    encoder's BatchNorm statistics are whole-batch, so the encoder runs once), the score loss per chunk, stepping the score
    network.  `events`: the whole-batch draws of the step in the reference's call order -- phase 1: posterior noise, then
    per-block (eps, poisson) draws, the loss's (rand, randn), the prior's randn; phase 2: posterior noise, (rand, randn).
    Pinned against the reference's two-step goldens (tests/test_oracle_golden.py)."""
    assert int(cfg.optim.num_micro_batch) == 1 and not cfg.training.st
    state.zero_grad()
    flow_state.zero_grad()
    B = batch.shape[0]
    P = flow_state.P
    x_in = squeeze(batch, 2) if cfg.flow.squeeze else batch
    D = np.prod(batch.shape[1:])
    # ---- phase 1 (losses.py:354-376) ----
    bn = {}
    enc_rng = ReplayRng(events[:1])
    zh, kl = sampling_and_kl(P, x_in, True, enc_rng, bn)
    h_leaf = zh.view(-1, zh.size(2)).detach().requires_grad_(True)
    kl_leaf = kl.detach().requires_grad_(True)
    # how many events phase 1 consumes after the posterior noise: replay chunk 0 on a counting source
    probe = ReplayRng([(k, v if k == "poisson" else v[0:min(B, chunk)]) for k, v in events[1:]])
    parts, used = [], None
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        rng = probe if lo == 0 else ReplayRng([(k, v if k == "poisson" else v[lo:hi]) for k, v in events[1:1 + used]])
        e, logpx = residual_flow_fwd(P, x_in[lo:hi], h_leaf[lo:hi], True, True, rng)
        z, lf = e, -logpx - kl_leaf[lo:hi]
        if cfg.flow.squeeze:
            z = torch.pixel_shuffle(z, 2)
        ls = sde_loss(state.P, cfg, sde, z, True, rng, importance_sampling=True)
        lp = calculate_logp(sde, z, rng)
        if lo == 0:
            used = rng.i
        else:
            assert rng.done()
        if cfg.training.reduce_mean:
            lf, lp = -lf / D, -lp / D
        else:
            lf, lp = -lf, -lp
        losses = ls + lf + lp
        (losses.sum() / B).backward()
        parts.append((losses.detach(), lf.detach(), lp.detach()))
    torch.autograd.backward([zh.view(-1, zh.size(2)), kl], [h_leaf.grad, kl_leaf.grad])
    with torch.no_grad():
        for k, v in bn.items():
            P[k] = v
    flow_state.optimize()
    flow_state.ema_update()
    state.zero_grad()
    # ---- phase 2 (losses.py:396-404): z from the updated flow, no log-det, no graph ----
    rest = events[1 + used:]
    bn = {}
    adds = []
    with torch.no_grad():
        enc_rng = ReplayRng(rest[:1])
        zh2, _ = sampling_and_kl(P, x_in, True, enc_rng, bn)
        h2 = zh2.view(-1, zh2.size(2))
        for k, v in bn.items():
            P[k] = v
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        rng = ReplayRng([(k, v if k == "poisson" else v[lo:hi]) for k, v in rest[1:]])
        with torch.no_grad():
            z2 = residual_flow_fwd(P, x_in[lo:hi], h2[lo:hi], False, True, rng)
            if cfg.flow.squeeze:
                z2 = torch.pixel_shuffle(z2, 2)
        la = sde_loss(state.P, cfg, sde, z2.detach(), True, rng, recon_loss=False)
        assert rng.done()
        (la.sum() / B).backward()
        adds.append(la.detach())
    state.optimize()
    state.ema_update()
    state.step += 1
    flow_state.step += 1
    return (torch.cat([p[0] for p in parts]), torch.cat(adds), torch.cat([p[1] for p in parts]),
            torch.cat([p[2] for p in parts]))


def flow_step_fn_fid(cfg, sde, state, flow_state, batch, rng, stats=None):
    """losses.py:322-406, training.st included (:378-395)"""
    st = bool(cfg.training.st)
    state.zero_grad()
    flow_state.zero_grad()
    keep = {}
    parts = [_nll_losses(cfg, sde, state, flow_state, mb, rng, stats, importance_sampling=True, keep=keep)
             for mb in _micro_batches(cfg, batch)]
    flow_state.optimize()
    flow_state.ema_update()
    if not st:
        state.zero_grad()
    adds = []
    for mb in _micro_batches(cfg, batch):
        if not st:
            bn = {}
            with torch.no_grad():
                z2, _ = flow_forward(flow_state.P, cfg, mb, log_det=None, training=True, rng=rng, bn_updates=bn)
                for k, v in bn.items():
                    flow_state.P[k] = v
        else:
            z2 = keep["z"]      # the LAST micro-batch's latent of the first phase, whatever `mb` is (losses.py:383-386)
        la = sde_loss(state.P, cfg, sde, z2.detach(), True, rng, recon_loss=False, st=st)
        if st:                  # the first phase's score gradients stay and are re-scaled in place (losses.py:389-395)
            const_adj = (la.mean() / keep["losses_score"].mean()).detach()
            for k in state.trainable:
                if state.P[k].grad is not None:
                    state.P[k].grad = const_adj * state.P[k].grad
        torch.mean(la).backward()
        adds.append(la.detach())
    state.optimize()
    state.ema_update()
    state.step += 1
    flow_state.step += 1
    return (torch.cat([p[0] for p in parts]), torch.cat(adds), torch.cat([p[2] for p in parts]),
            torch.cat([p[3] for p in parts]))


# ----------------------------------------------------------------------------- NELBO / NLL
def likelihood_residual(Ps, cfg, sde, score_fn, batch, rng, eps_bpd):
    """get_likelihood_residual_fn likelihood.py:241-283 (variance='scoreflow')"""
    B = batch.shape[0]
    eps_vec = torch.ones(B) * cfg.training.truncation_time
    mean, std = sde.marginal_prob(batch, eps_vec)
    z = rng.randn_like(batch)
    score = score_fn(mean + std[:, None, None, None] * z, eps_vec)
    noise = -std[:, None, None, None] * score
    eps_vec = torch.ones(B) * eps_bpd
    mean, std = sde.marginal_prob(batch, eps_vec)
    z = rng.randn_like(batch)
    perturbed = mean + std[:, None, None, None] * z
    alpha, beta = sde.marginal_prob(torch.ones_like(batch), eps_vec)
    q_mean = perturbed / alpha - beta[:, None, None, None] * noise / alpha
    q_std = beta / torch.mean(alpha, axis=(1, 2, 3))
    n_dim = np.prod(batch.shape[1:])
    p_entropy = n_dim / 2.0 * (np.log(2 * np.pi) + 2 * torch.log(std) + 1.0)
    q_recon = n_dim / 2.0 * (np.log(2 * np.pi) + 2 * torch.log(q_std)) + 0.5 / (q_std ** 2) * torch.square(
        batch - q_mean).sum(axis=(1, 2, 3))
    return q_recon - p_entropy


def inverse_scaler(cfg, x):  # datasets.py:64-71
    return (x + 1.0) / 2.0 if cfg.data.centered else x


def nelbo_fn(Ps, Pf, cfg, sde, batch, rng, stats=None):
    """get_elbo_fn.loss_fn likelihood.py:160-236 (flow in eval mode, Rademacher)"""
    with torch.enable_grad():
        batch, log_jacob = flow_forward(Pf, cfg, batch, training=False, rng=rng, stats=stats)
        log_jacob = log_jacob.squeeze()
        B = batch.shape[0]
        score_fn = get_score_fn(Ps, cfg, sde, train=False)
        t, Z = sde.get_diffusion_time(B, sde.eps, True, rng)
        qt = 1 / sde.T
        z = rng.randn_like(batch)
        mean, std = sde.marginal_prob(batch, t)
        perturbed = (mean + std[:, None, None, None] * z).requires_grad_()
        score = score_fn(perturbed, t)
        f, g = sde.sde(perturbed, t)
        s4 = std[:, None, None, None]
        a = s4 * score
        mu = (s4 ** 2) * score - (s4 ** 2) / (g[:, None, None, None] ** 2) * f
        epsilon = rng.rademacher_like(batch)
        Mu = -(torch.autograd.grad(mu, perturbed, epsilon, create_graph=False)[0] * epsilon).reshape(B, -1).sum(
            1) * Z / qt
        Nu = -(a ** 2).reshape(B, -1).sum(1) * Z / 2 / qt
        Mus, Nus = Mu.detach(), Nu.detach()
        lp_t = torch.ones_like(t) * sde.T
        lp_z = rng.randn_like(batch)
        lp_mean, lp_std = sde.marginal_prob(batch, lp_t)
        lp = sde.prior_logp(lp_mean + lp_std[:, None, None, None] * lp_z)
        elbos = lp + Mus + Nus + log_jacob
        residual = likelihood_residual(Ps, cfg, sde, score_fn, batch, rng, cfg.training.truncation_time)
        D = np.prod(list(batch.shape[1:]))
        off = 7.0 - inverse_scaler(cfg, -1.0)
        return (-(elbos) / D / np.log(2) + off).detach(), (-(elbos - residual) / D / np.log(2) + off).detach()


def pf_ode_drift_and_div(Ps, cfg, sde, x, t, epsilon):
    """drift_fn / div_fn likelihood.py:62-71, :27-38 -- one RHS evaluation of the NLL ODE."""
    score_fn = get_score_fn(Ps, cfg, sde, train=False)
    rsde, _ = reverse_sde(sde, score_fn, probability_flow=True)
    with torch.enable_grad():
        xx = x.detach().requires_grad_(True)
        drift = rsde(xx, t)[0]
        grad = torch.autograd.grad(torch.sum(drift * epsilon), xx)[0]
    return drift.detach(), torch.sum(grad * epsilon, dim=(1, 2, 3))


def likelihood_fn(Ps, Pf, cfg, sde, data, rng, residual=True, eps_bpd=1e-5, rtol=1e-5, atol=1e-5, stats=None):
    """get_likelihood_fn.likelihood_fn likelihood.py:70-138 (SciPy RK45 over a float64 numpy state)"""
    from scipy import integrate
    with torch.no_grad():
        shape = data.shape
        B = shape[0]
        score_fn = get_score_fn(Ps, cfg, sde, train=False)
        epsilon = rng.rademacher_like(data)

        def ode_func(t, x):
            sample = torch.from_numpy(x[:-B].reshape(shape)).type(torch.float32)
            vec_t = torch.ones(B) * t
            drift, div = pf_ode_drift_and_div(Ps, cfg, sde, sample, vec_t, epsilon)
            return np.concatenate([drift.numpy().reshape(-1), div.numpy().reshape(-1)], axis=0)

        data, log_jacob = flow_forward(Pf, cfg, data, training=False, rng=rng, stats=stats)
        data, log_jacob = data.detach(), log_jacob.detach()
        if residual:
            z = rng.randn_like(data)
            mean, std = sde.marginal_prob(data, torch.ones(B) * eps_bpd)
            init = np.concatenate([(mean + std[:, None, None, None] * z).numpy().reshape(-1), np.zeros((B,))])
        else:
            init = np.concatenate([data.numpy().reshape(-1), np.zeros((B,))])
        sol = integrate.solve_ivp(ode_func, (eps_bpd, sde.T), init, rtol=rtol, atol=atol, method="RK45")
        zp = sol.y[:, -1]
        z = torch.from_numpy(zp[:-B].reshape(shape)).type(torch.float32)
        delta_logp = torch.from_numpy(zp[-B:]).type(torch.float32)
        prior_logp = sde.prior_logp(z)
        if residual:
            delta_logp = delta_logp - likelihood_residual(Ps, cfg, sde, score_fn, data, rng, eps_bpd)
        bpd = -(prior_logp + delta_logp + log_jacob) / np.log(2) / np.prod(shape[1:])
        return bpd + 7.0 - inverse_scaler(cfg, -1.0), z, sol.nfev


# ----------------------------------------------------------------------------- samplers
def em_predictor(sde, score_fn, x, t, rng, probability_flow=False):  # sampling.py:186-197 (Predictor.__init__ :139-144)
    dt = -1.0 / sde.N
    z = rng.randn_like(x)
    rsde, _ = reverse_sde(sde, score_fn, probability_flow)
    drift, diffusion = rsde(x, t)
    x_mean = x + drift * dt
    # probability_flow=True: rsde returns the float 0. as diffusion and this line raises TypeError -- the reference's own
    # behaviour (sampling.py:196; golden mini_r4.npz pf_em/raises)
    return x_mean + diffusion[:, None, None, None] * np.sqrt(-dt) * z, x_mean


def rd_predictor(sde, score_fn, x, t, rng, next_t=None, probability_flow=False):  # sampling.py:200-210
    _, rdisc = reverse_sde(sde, score_fn, probability_flow)
    f, G = rdisc(x, t, next_t)
    z = rng.randn_like(x)
    x_mean = x - f
    return x_mean + G[:, None, None, None] * z, x_mean


def langevin_corrector(sde, score_fn, x, t, rng, snr, n_steps):  # sampling.py:263-292
    if isinstance(sde, VPSDE):
        alpha = sde.alphas[(t * (sde.N - 1) / sde.T).long()]
    else:
        alpha = torch.ones_like(t)
    x_mean = x
    for _ in range(n_steps):
        grad = score_fn(x, t)
        noise = rng.randn_like(x)
        grad_norm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
        noise_norm = torch.norm(noise.reshape(noise.shape[0], -1), dim=-1).mean()
        step = (snr * noise_norm / grad_norm) ** 2 * 2 * alpha
        x_mean = x + step[:, None, None, None] * grad
        x = x_mean + torch.sqrt(step * 2)[:, None, None, None] * noise
    return x, x_mean


def pc_sampler(Ps, Pf, cfg, sde, shape, rng, eps, num_scales=None, temperature=1.0, data_mean=None, iters=None,
               trace=None):
    """get_pc_sampler.pc_sampler sampling.py:410-456 (the npz side-write at :436-445 is I/O, omitted)"""
    with torch.no_grad():
        score_fn = get_score_fn(Ps, cfg, sde, train=False)
        num_scales = num_scales or (cfg.sampling.num_scales if cfg.sampling.num_scales != sde.N else sde.N)
        x = sde.prior_sampling(shape, rng, data_mean)
        timesteps = torch.linspace(sde.T, eps, num_scales)
        pred, corr = cfg.sampling.predictor.lower(), cfg.sampling.corrector.lower()
        x_mean = x
        for i in range(num_scales):
            vec_t = torch.ones(shape[0]) * timesteps[i]
            if corr == "langevin":
                x, x_mean = langevin_corrector(sde, score_fn, x, vec_t, rng, cfg.sampling.snr,
                                               cfg.sampling.n_steps_each)
            if pred == "euler_maruyama":
                x, x_mean = em_predictor(sde, score_fn, x, vec_t, rng, cfg.sampling.probability_flow)
            elif pred == "reverse_diffusion":
                x, x_mean = rd_predictor(sde, score_fn, x, vec_t, rng, probability_flow=cfg.sampling.probability_flow)
            if trace is not None:
                trace.append(x.clone())
        before = x_mean if cfg.sampling.noise_removal else x
        after, _ = flow_forward(Pf, cfg, before * temperature, log_det=None, reverse=True, rng=rng, iters=iters)
        return inverse_scaler(cfg, before), inverse_scaler(cfg, after), sde.N * (cfg.sampling.n_steps_each + 1)


def ode_sampler(Ps, Pf, cfg, sde, shape, rng, eps, rtol=1e-5, atol=1e-5, temperature=1.0, iters=None):
    """get_ode_sampler.ode_sampler sampling.py:582-619 (+ denoise_update_fn :568-574)"""
    from scipy import integrate
    with torch.no_grad():
        score_fn = get_score_fn(Ps, cfg, sde, train=False)
        rsde, _ = reverse_sde(sde, score_fn, probability_flow=True)
        x = sde.prior_sampling(shape, rng)

        def ode_func(t, xf):
            xx = torch.from_numpy(xf.reshape(shape)).type(torch.float32)
            return rsde(xx, torch.ones(shape[0]) * t)[0].numpy().reshape(-1)

        sol = integrate.solve_ivp(ode_func, (sde.T, eps), x.numpy().reshape(-1), rtol=rtol, atol=atol, method="RK45")
        x = torch.from_numpy(sol.y[:, -1].reshape(shape)).type(torch.float32)
        if cfg.sampling.noise_removal:
            vec_eps = torch.ones(shape[0]) * eps
            _, x = rd_predictor(sde, score_fn, x, vec_eps, rng, next_t=torch.zeros_like(vec_eps), probability_flow=False)
        after, _ = flow_forward(Pf, cfg, x * temperature, log_det=None, reverse=True, rng=rng, iters=iters)
        return inverse_scaler(cfg, x), inverse_scaler(cfg, after), sol.nfev


def _pc_updates(cfg, sde, score_fn, rng):
    """shared_{corrector,predictor}_update_fn sampling.py:343-362 for the registries the configs select."""
    pred, corr = cfg.sampling.predictor.lower(), cfg.sampling.corrector.lower()

    def corrector(x, vec_t, snr):
        if corr == "langevin":
            return langevin_corrector(sde, score_fn, x, vec_t, rng, snr, cfg.sampling.n_steps_each)
        return x, x

    def predictor(x, vec_t, vec_next_t):
        if pred == "euler_maruyama":
            return em_predictor(sde, score_fn, x, vec_t, rng, cfg.sampling.probability_flow)
        if pred == "reverse_diffusion":
            return rd_predictor(sde, score_fn, x, vec_t, rng, vec_next_t, cfg.sampling.probability_flow)
        return x, x

    return corrector, predictor


def denoise_update_fn(sde, score_fn, x, rng, eps, final_time):  # sampling.py:402-408
    vec_eps = torch.ones(x.shape[0]) * eps
    return rd_predictor(sde, score_fn, x, vec_eps, rng, torch.ones_like(vec_eps) * final_time,
                        probability_flow=True)[1]


def _finish_sampler(Pf, cfg, before, temperature, rng, iters):
    after, _ = flow_forward(Pf, cfg, before * temperature, log_det=None, reverse=True, rng=rng, iters=iters)
    return inverse_scaler(cfg, before), inverse_scaler(cfg, after)


def pc_sampler_search(Ps, Pf, cfg, sde, shape, rng, eps, temperature=1.0, data_mean=None, final_time=0.0,
                      before_data=None, iters=None):
    """get_pc_sampler.pc_sampler_search sampling.py:458-495 (selected by sampling.pc_denoise)"""
    with torch.no_grad():
        score_fn = get_score_fn(Ps, cfg, sde, train=False)
        corrector, predictor = _pc_updates(cfg, sde, score_fn, rng)
        if before_data is None:
            x = sde.prior_sampling(shape, rng, data_mean)
            timesteps = torch.linspace(sde.T, eps, sde.N)
            x_mean = x
            for i in range(sde.N - 1):
                vec_t = torch.ones(shape[0]) * timesteps[i]
                vec_next_t = torch.ones(shape[0]) * timesteps[i + 1]
                x, x_mean = corrector(x, vec_t, cfg.sampling.snr)
                x, x_mean = predictor(x, vec_t, vec_next_t)
        else:
            x_mean = x = before_data
        denoise = cfg.sampling.noise_removal
        if not cfg.sampling.need_sample:
            x_mean = x = denoise_update_fn(sde, score_fn, x_mean if denoise else x, rng, eps, final_time)
        before, after = _finish_sampler(Pf, cfg, x_mean if denoise else x, temperature, rng, iters)
        return before, after, sde.N * (cfg.sampling.n_steps_each + 1)


def pc_sampler_more_step(Ps, Pf, cfg, sde, shape, rng, eps, temperature=1.0, data_mean=None, final_time=0.0,
                         before_data=None, iters=None):
    """get_pc_sampler.pc_sampler_more_step sampling.py:497-535 (selected by sampling.more_step)"""
    with torch.no_grad():
        score_fn = get_score_fn(Ps, cfg, sde, train=False)
        corrector, predictor = _pc_updates(cfg, sde, score_fn, rng)
        if before_data is None:
            # the reference calls corrector_update_fn without `snr` here (sampling.py:512) -> TypeError on step 0
            raise TypeError("shared_corrector_update_fn() missing 1 required positional argument: 'snr'")
        x_mean = x = before_data
        if not cfg.sampling.need_sample:
            timesteps = torch.exp(torch.linspace(np.log(1e-3), np.log(1e-5), 100))
            for i in range(100):
                vec_t = torch.ones(shape[0]) * timesteps[i]
                x, x_mean = corrector(x, vec_t, cfg.sampling.snr)
                x, x_mean = predictor(x, vec_t, None)
        before, after = _finish_sampler(Pf, cfg, x_mean if cfg.sampling.noise_removal else x, temperature, rng, iters)
        return before, after, sde.N * (cfg.sampling.n_steps_each + 1)


def get_bpd(Ps, Pf, cfg, sde, eval_batches, rng, eval=False, rtol=1e-5, atol=1e-5):
    """evaluation.get_bpd evaluation.py:388-494: `num_nelbo` NELBO passes, then NLL "wrong" (residual=False) and
    "correct" (residual=True) at eps_bpd, then at the training truncation time if that differs from 1e-5.
    eval_batches: list of [B,C,H,W] tensors in [0,1].  Returns the means the reference logs and every call's output."""
    num_data = cfg.eval.num_test_data if eval else 10000
    scale = (lambda x: x * 2.0 - 1.0) if cfg.data.centered else (lambda x: x)
    res, calls = {}, []

    def batches(n):
        it = iter(eval_batches)
        for _ in range((n - 1) // cfg.eval.batch_size + 1):
            try:
                b = next(it)
            except StopIteration:  # datasets.get_batch restarts the iterator (datasets.py:106-113)
                it = iter(eval_batches)
                b = next(it)
            yield scale((255.0 * b + rng.rand(*b.shape)) / 256.0)

    full, full_r = [], []
    for _ in range(cfg.eval.num_nelbo):
        a, r = [], []
        for b in batches(num_data):
            n1, n2 = nelbo_fn(Ps, Pf, cfg, sde, b, rng)
            calls.append(("nelbo", n1, n2))
            a.extend(n1.numpy().reshape(-1)); r.extend(n2.numpy().reshape(-1))
        full.append(np.mean(a)); full_r.append(np.mean(r))
    res["nelbo_bpd_per_eval"], res["nelbo_residual_bpd_per_eval"] = full, full_r
    res["nelbo_bpd"], res["nelbo_residual_bpd"] = np.mean(full), np.mean(full_r)
    if not eval:
        num_data = num_data // 10
    eps_bpd = 1e-5 if cfg.eval.truncation_time == -1.0 else cfg.eval.truncation_time

    def nll_pass(residual, e):
        v = []
        for b in batches(num_data):
            bpd, _, nfe = likelihood_fn(Ps, Pf, cfg, sde, b, rng, residual=residual, eps_bpd=e, rtol=rtol, atol=atol)
            calls.append(("nll", bpd, (residual, e, nfe)))
            v.extend(bpd.numpy().reshape(-1))
        return np.mean(v)

    if not cfg.eval.skip_nll_wrong:
        res["nll_wrong_bpd"] = nll_pass(False, eps_bpd)
    res["nll_correct_bpd"] = nll_pass(True, eps_bpd)
    if cfg.training.truncation_time != 1e-5:
        res["nll_correct_train_eps_bpd"] = nll_pass(True, cfg.training.truncation_time)
    return res, calls
