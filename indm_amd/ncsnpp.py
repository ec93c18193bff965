"""NCSN++ score network on the HIP kernels (mirror of models/ncsnpp.py + models/layerspp.py).

Same module list / state-dict names as the reference (`all_modules.N.<Conv_0|GroupNorm_0|...>`), same
call signature `score_model(x, time_cond) -> [B,C,H,W]`.  Internally everything is NHWC fp32:
  * GroupNorm+SiLU(+dropout) is one fused kernel pair, Philox dropout is regenerated in backward
  * 3x3 convs are MFMA implicit GEMMs with the temb bias, the residual add and the 1/sqrt(2)
    skip-rescale folded into the epilogue; 1x1 convs / NIN / Dense are plain MFMA GEMMs
  * attention = batched QK^T GEMM -> wavefront-per-row softmax -> PV GEMM -> out-proj GEMM whose
    epilogue applies (x + h)/sqrt(2)
Conv weights are stored RSCK ([KH,KW,Cin,Cout]) and 1x1 weights [Cin,Cout]; `load_reference_state_dict`
/ `reference_state_dict` convert to/from the reference's OIHW layout at the checkpoint boundary.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import hip, ops


# ----------------------------------------------------------------------------- initialisers (models/layers.py:54-91)
def default_init_(shape_oihw, scale=1.0):
    """variance_scaling(scale, 'fan_avg', 'uniform') with in_axis=1, out_axis=0 (DDPM init)."""
    scale = 1e-10 if scale == 0 else scale
    rf = np.prod(shape_oihw) / shape_oihw[1] / shape_oihw[0]
    fan_in, fan_out = shape_oihw[1] * rf, shape_oihw[0] * rf
    variance = scale / ((fan_in + fan_out) / 2)
    return (torch.rand(*shape_oihw) * 2.0 - 1.0) * np.sqrt(3 * variance)


class Conv3x3(nn.Module):
    """ddpm_conv3x3 (models/layers.py:113-124); weight kept RSCK."""

    def __init__(self, in_ch, out_ch, init_scale=1.0):
        super().__init__()
        torch.nn.Conv2d(in_ch, out_ch, 3)  # consume the RNG exactly like the reference's constructor
        w = default_init_((out_ch, in_ch, 3, 3), init_scale)
        self.weight = nn.Parameter(w.permute(2, 3, 1, 0).contiguous())
        self.bias = nn.Parameter(torch.zeros(out_ch))

    def forward(self, x, bias2=None, res=None, alpha=1.0, beta=1.0, res_grad_unscaled=False):
        return ops.conv2d(x, self.weight, self.bias, bias2, res, 1, 1, alpha, beta, pack=self.__dict__.pop("h16_pack", None),
                          res_grad_unscaled=res_grad_unscaled)


class Conv1x1(nn.Module):
    """ddpm_conv1x1 (models/layers.py:88-97); weight kept [Cin, Cout]."""

    def __init__(self, in_ch, out_ch, init_scale=1.0):
        super().__init__()
        torch.nn.Conv2d(in_ch, out_ch, 1)
        w = default_init_((out_ch, in_ch, 1, 1), init_scale)
        self.weight = nn.Parameter(w[:, :, 0, 0].t().contiguous())
        self.bias = nn.Parameter(torch.zeros(out_ch))

    def forward(self, x, res=None, alpha=1.0, beta=1.0):
        return ops.conv1x1(x, self.weight, self.bias, None, res, alpha, beta, pack=self.__dict__.pop("h16_pack", None))


class Dense(nn.Module):
    """nn.Linear with default_init weight, zero bias (layerspp.py:239-241, ncsnpp.py:97-102)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        torch.nn.Linear(in_dim, out_dim)
        self.weight = nn.Parameter(default_init_((out_dim, in_dim)))
        self.bias = nn.Parameter(torch.zeros(out_dim))

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class NIN(nn.Module):
    """models/layers.py:546-556: W is [in, out] (already GEMM-native)."""

    def __init__(self, in_dim, num_units, init_scale=0.1):
        super().__init__()
        self.W = nn.Parameter(default_init_((in_dim, num_units), init_scale))
        self.b = nn.Parameter(torch.zeros(num_units))


class GroupNorm(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(ch))
        self.bias = nn.Parameter(torch.zeros(ch))

    def forward(self, x, silu=False, drop_p=0.0, seed=0):
        return ops.group_norm(x, self.weight, self.bias, 1e-6, silu, drop_p, seed)


class GaussianFourierProjection(nn.Module):
    """models/layerspp.py:45-54 (W is not trainable)."""

    def __init__(self, embedding_size=256, scale=1.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)

    def forward(self, x):
        xp = x[:, None] * self.W[None, :] * 2 * np.pi
        return torch.cat([torch.sin(xp), torch.cos(xp)], dim=-1)


def get_timestep_embedding(timesteps, embedding_dim, max_positions=10000):
    """models/layers.py:515-529 ([B] scalars -> [B, dim]; B*dim floats of host-side-sized glue)."""
    half = embedding_dim // 2
    emb = math.log(max_positions) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32, device=timesteps.device) * -emb)
    emb = timesteps.float()[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb


def _setup_kernel(k):
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k /= np.sum(k)
    return k


class _Fir:
    """FIR resampling on NHWC via the upfirdn2d kernel (models/up_or_down_sampling.py:181-257)."""

    def __init__(self, fir_kernel):
        self.k = _setup_kernel(fir_kernel)
        self._cache = {}

    def _dev(self, scale, device):
        key = (scale, str(device))
        if key not in self._cache:
            self._cache[key] = torch.tensor(self.k * scale, dtype=torch.float32, device=device)
        return self._cache[key]

    def upsample(self, x, factor=2):
        p = self.k.shape[0] - factor
        return ops.upfirdn2d(x, self._dev(factor ** 2, x.device), up=factor,
                             pad=((p + 1) // 2 + factor - 1, p // 2))

    def downsample(self, x, factor=2):
        p = self.k.shape[0] - factor
        return ops.upfirdn2d(x, self._dev(1, x.device), down=factor, pad=((p + 1) // 2, p // 2))

    def pre_conv_down(self, x, conv_w=3, factor=2):
        p = (self.k.shape[0] - factor) + (conv_w - 1)
        return ops.upfirdn2d(x, self._dev(1, x.device), pad=((p + 1) // 2, p // 2))


class ResnetBlockBigGANpp(nn.Module):
    """models/layerspp.py:225-287"""

    def __init__(self, in_ch, out_ch=None, temb_dim=None, up=False, down=False, dropout=0.1, fir=False,
                 fir_kernel=(1, 3, 3, 1), skip_rescale=True, init_scale=0.0):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        self.GroupNorm_0 = GroupNorm(in_ch)
        self.up, self.down, self.fir = up, down, fir
        self.firop = _Fir(fir_kernel) if fir else None
        self.Conv_0 = Conv3x3(in_ch, out_ch)
        if temb_dim is not None:
            self.Dense_0 = Dense(temb_dim, out_ch)
        self.GroupNorm_1 = GroupNorm(out_ch)
        self.dropout = dropout
        self.Conv_1 = Conv3x3(out_ch, out_ch, init_scale=init_scale)
        if in_ch != out_ch or up or down:
            self.Conv_2 = Conv1x1(in_ch, out_ch)
        self.skip_rescale, self.in_ch, self.out_ch = skip_rescale, in_ch, out_ch

    def forward(self, x, temb_act=None, rng=None):
        """x NHWC; temb_act = SiLU(temb) (shared by all blocks, ncsnpp.py passes temb and each block
        applies act -- the activation is hoisted because it is identical for every block)."""
        shortcut = self.in_ch != self.out_ch or self.up or self.down
        a = ops.INV_SQRT2 if self.skip_rescale else 1.0
        fused_skip = x.dtype == torch.bfloat16   # H16: both gradients of x come back through the GroupNorm backward kernel
        if fused_skip:
            g = self.GroupNorm_0
            h, x = ops.group_norm_res(x, g.weight, g.bias, 1e-6, True, 0.0, 0, 1.0 if shortcut else a)
        else:
            h = self.GroupNorm_0(x, silu=True)
        if self.up:
            if self.fir:
                h, x = self.firop.upsample(h), self.firop.upsample(x)
            else:
                h, x = ops.upsample2x(h), ops.upsample2x(x)
        elif self.down:
            if self.fir:
                h, x = self.firop.downsample(h), self.firop.downsample(x)
            else:
                h, x = ops.downsample2x(h), ops.downsample2x(x)
        tb = self.__dict__.pop("tb_next", None)     # precomputed by NCSNpp.forward for all blocks at once
        if tb is None and temb_act is not None:
            tb = self.Dense_0(temb_act)
        h = self.Conv_0(h, bias2=tb)
        drop = self.dropout if self.training else 0.0
        seed = 0
        if drop > 0:    # never a constant mask: a missing noise source falls back to the process-wide advancing one
            if rng is None:
                from .rng import default_rng
                rng = default_rng(h.device)
            seed = rng.dropout_seed()
            if seed is None:
                raise RuntimeError("training with dropout > 0 on a replayed noise source: the parity fixtures are "
                                   "recorded with model.dropout = 0")
        h = self.GroupNorm_1(h, silu=True, drop_p=drop, seed=seed)
        if shortcut:
            x = self.Conv_2(x)
        # (fused_skip without a shortcut: x IS the GroupNorm node's skip output, which applies the factor `a` itself)
        return self.Conv_1(h, res=x, alpha=a, beta=a, res_grad_unscaled=fused_skip and not shortcut)


class AttnBlockpp(nn.Module):
    """models/layerspp.py:75-104 (single head, head dim = C, tokens = H*W)."""

    def __init__(self, channels, skip_rescale=False, init_scale=0.0):
        super().__init__()
        self.GroupNorm_0 = GroupNorm(channels)
        self.NIN_0 = NIN(channels, channels)
        self.NIN_1 = NIN(channels, channels)
        self.NIN_2 = NIN(channels, channels)
        self.NIN_3 = NIN(channels, channels, init_scale=init_scale)
        self.skip_rescale = skip_rescale

    def forward(self, x):
        if x.dtype == torch.bfloat16:   # H16 storage (round 6)
            B, H, W, C = x.shape
            pk = self.__dict__.pop("h16_pack", None)
            if pk is not None and ops.attn_h16_covered(H * W, C):
                # GroupNorm, ONE QKV GEMM, the fused bf16 attention kernel, the output projection with (x + h) / sqrt(2)
                a = ops.INV_SQRT2 if self.skip_rescale else 1.0
                nins = [(m.W, m.b) for m in (self.NIN_0, self.NIN_1, self.NIN_2, self.NIN_3)]
                return ops.attn_core_h16(self.GroupNorm_0(x), x, nins, pk, float(int(C) ** (-0.5)), a)
            # geometries the bf16 kernels do not cover (4 x 4 maps: 16 tokens): an fp32-storage island between two converters
            return ops.to_bf16(self.forward(ops.to_f32(x)))
        B, H, W, C = x.shape
        T = H * W
        h = self.GroupNorm_0(x).reshape(B * T, C)
        q = ops.gemm(h, self.NIN_0.W, self.NIN_0.b, weight_b=True).reshape(B, T, C)
        k = ops.gemm(h, self.NIN_1.W, self.NIN_1.b, weight_b=True).reshape(B, T, C)
        v = ops.gemm(h, self.NIN_2.W, self.NIN_2.b, weight_b=True).reshape(B, T, C)
        h = ops.attention(q, k, v, float(int(C) ** (-0.5))).reshape(B * T, C)   # fused: no [B,T,T] logits in HBM
        a = ops.INV_SQRT2 if self.skip_rescale else 1.0
        out = ops.gemm(h, self.NIN_3.W, self.NIN_3.b, res=x.reshape(B * T, C), alpha=a, beta=a, weight_b=True)
        return out.reshape(B, H, W, C)


class PyramidDownsample(nn.Module):
    """layerspp.Downsample(with_conv=True) (models/layerspp.py:142-176).  fir=True: FIR-pad then a 3x3 stride-2
    conv (up_or_down_sampling.Conv2d(down=True), :23-56, conv_downsample_2d :144-178)."""

    def __init__(self, in_ch, out_ch, fir, fir_kernel):
        super().__init__()
        self.fir = fir
        inner = nn.Module()
        torch.nn.Conv2d(in_ch, out_ch, 3) if not fir else None
        inner.weight = nn.Parameter(default_init_((out_ch, in_ch, 3, 3)).permute(2, 3, 1, 0).contiguous())
        inner.bias = nn.Parameter(torch.zeros(out_ch))
        if fir:
            self.Conv2d_0 = inner
            self.firop = _Fir(fir_kernel)
        else:
            self.Conv_0 = inner

    def forward(self, x, res=None, alpha=1.0, beta=1.0):
        c = self.Conv2d_0 if self.fir else self.Conv_0
        if self.fir:
            x = self.firop.pre_conv_down(x)
        else:
            x = torch.nn.functional.pad(x, (0, 0, 0, 1, 0, 1))  # NHWC: pad W then H by (0,1)
        pack = c.__dict__.pop("h16_pack", None)
        if res is not None and res.dtype == torch.bfloat16 and x.dtype != torch.bfloat16:
            # the first pyramid level convolves the fp32 3-channel image: computed in fp32, joined to the bf16 trunk afterwards
            return ops.axpby_mixed(ops.conv2d(x, c.weight, c.bias, None, None, 2, 0, alpha, 1.0), res, 1.0, beta)
        return ops.conv2d(x, c.weight, c.bias, None, res, 2, 0, alpha, beta, pack=pack)


class NCSNpp(nn.Module):
    """models/ncsnpp.py:34-414 for resblock_type='biggan', progressive='none',
    progressive_input in {'none','residual'}, conditional=True (every shipped config)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        m = config.model
        if m.resblock_type.lower() != "biggan" or m.progressive.lower() != "none" or not m.conditional:
            raise NotImplementedError("only the configurations shipped with the reference are supported")
        if m.progressive_input.lower() not in ("none", "residual"):
            raise NotImplementedError(m.progressive_input)
        sig = np.exp(np.linspace(np.log(m.sigma_max), np.log(m.sigma_min), m.num_scales))  # models/utils.py:50-60
        self.register_buffer("sigmas", torch.tensor(sig))
        self.nf = nf = m.nf
        ch_mult = m.ch_mult
        self.num_res_blocks = nrb = m.num_res_blocks
        self.attn_resolutions = m.attn_resolutions
        self.num_resolutions = nres = len(ch_mult)
        self.all_resolutions = [config.data.image_size // (2 ** i) for i in range(nres)]
        self.embedding_type = m.embedding_type.lower()
        self.progressive_input = m.progressive_input.lower()
        self.skip_rescale = m.skip_rescale
        self.aux = m.auxiliary_resblock
        self.attention = m.attention
        fir, fk, init_scale, dropout = m.fir, m.fir_kernel, m.init_scale, m.dropout

        def RB(**kw):
            return ResnetBlockBigGANpp(temb_dim=nf * 4, dropout=dropout, fir=fir, fir_kernel=fk,
                                       init_scale=init_scale, skip_rescale=self.skip_rescale, **kw)

        def AT(ch):
            return AttnBlockpp(ch, skip_rescale=self.skip_rescale, init_scale=init_scale)

        modules = []
        if self.embedding_type == "fourier":
            modules.append(GaussianFourierProjection(embedding_size=nf, scale=m.fourier_scale))
            embed_dim = 2 * nf
        else:
            embed_dim = nf
        modules.append(Dense(embed_dim, nf * 4))
        modules.append(Dense(nf * 4, nf * 4))
        channels = config.data.num_channels
        input_pyramid_ch = channels
        modules.append(Conv3x3(channels, nf))
        hs_c = [nf]
        in_ch = nf
        for lvl in range(nres):
            for _ in range(nrb):
                out_ch = nf * ch_mult[lvl]
                modules.append(RB(in_ch=in_ch, out_ch=out_ch))
                in_ch = out_ch
                if self.all_resolutions[lvl] in self.attn_resolutions and self.attention:
                    modules.append(AT(in_ch))
                hs_c.append(in_ch)
            if lvl != nres - 1:
                if self.aux:
                    modules.append(RB(down=True, in_ch=in_ch))
                if self.progressive_input == "residual":
                    modules.append(PyramidDownsample(input_pyramid_ch, in_ch, fir, fk))
                    input_pyramid_ch = in_ch
                if self.aux:
                    hs_c.append(in_ch)
        in_ch = hs_c[-1]
        if not self.aux:
            hs_c.pop()
        modules.append(RB(in_ch=in_ch))
        modules.append(AT(in_ch))
        modules.append(RB(in_ch=in_ch))
        nup = nrb + 1 if self.aux else nrb
        for lvl in reversed(range(nres)):
            for _ in range(nup):
                out_ch = nf * ch_mult[lvl]
                modules.append(RB(in_ch=in_ch + hs_c.pop(), out_ch=out_ch))
                in_ch = out_ch
            if self.all_resolutions[lvl] in self.attn_resolutions and self.attention:
                modules.append(AT(in_ch))
            if lvl != 0 and self.aux:
                modules.append(RB(in_ch=in_ch, up=True))
        assert not hs_c
        modules.append(GroupNorm(in_ch))
        modules.append(Conv3x3(in_ch, channels, init_scale=init_scale))
        self.all_modules = nn.ModuleList(modules)

    # ------------------------------------------------------------------ forward
    def forward(self, x, time_cond, rng=None, nhwc=False):
        """x: [B,C,H,W] (or NHWC when nhwc=True; the return layout follows the input)."""
        mods = self.all_modules
        i = 0
        if self.embedding_type == "fourier":
            used_sigmas = time_cond
            temb = mods[i](torch.log(used_sigmas)); i += 1
        else:
            used_sigmas = self.sigmas[time_cond.long()]
            temb = get_timestep_embedding(time_cond, self.nf)
        temb = mods[i](temb); i += 1
        temb = mods[i](ops.silu(temb)); i += 1
        temb_act = ops.silu(temb)
        # Dense_0(act(temb)) of ALL residual blocks (layerspp.py:262-263) as ONE [B,512] x [512, sum C] GEMM instead of
        # 44 short-M launches forward and 3 x 44 backward; every block takes its (contiguous) column slice as bias2
        blocks = [m for m in mods if isinstance(m, ResnetBlockBigGANpp) and hasattr(m, "Dense_0")]
        if blocks:
            tb_all = ops.linear(temb_act, torch.cat([b.Dense_0.weight for b in blocks], dim=0),
                                torch.cat([b.Dense_0.bias for b in blocks], dim=0))
            for b, t in zip(blocks, ops.split_cols(tb_all, [b.out_ch for b in blocks])):
                b.tb_next = t
        h0 = x if nhwc else ops.to_nhwc(x)
        if not self.config.data.centered:
            h0 = ops.axpby(h0, None, 2.0, 0.0, -1.0)  # 2x-1 (ncsnpp.py:278-280)
        input_pyramid = h0 if self.progressive_input != "none" else None
        h16 = hip.h16() and x.is_cuda     # bf16 compute mode: the trunk's activations are stored bf16 (csrc/unet_h16*.hip)
        if h16:
            self._h16_pack_weights()
        first = mods[i](h0); i += 1      # 3 -> nf channels: the thin fp32 kernel either way
        hs = [ops.to_bf16(first) if h16 else first]
        a = ops.INV_SQRT2 if self.skip_rescale else 1.0
        for lvl in range(self.num_resolutions):
            for _ in range(self.num_res_blocks):
                h = mods[i](hs[-1], temb_act, rng); i += 1
                if h.shape[2] in self.attn_resolutions and self.attention:
                    h = mods[i](h); i += 1
                hs.append(h)
            if lvl != self.num_resolutions - 1:
                if self.aux:
                    h = mods[i](hs[-1], temb_act, rng); i += 1
                if self.progressive_input == "residual":
                    input_pyramid = mods[i](input_pyramid, res=h, alpha=a, beta=a); i += 1
                    h = input_pyramid
                if self.aux:
                    hs.append(h)
        h = hs[-1]
        if not self.aux:
            hs.pop()
        h = mods[i](h, temb_act, rng); i += 1
        h = mods[i](h); i += 1
        h = mods[i](h, temb_act, rng); i += 1
        nup = self.num_res_blocks + 1 if self.aux else self.num_res_blocks
        for lvl in reversed(range(self.num_resolutions)):
            for _ in range(nup):
                h = mods[i](torch.cat([h, hs.pop()], dim=3), temb_act, rng); i += 1
            if h.shape[2] in self.attn_resolutions and self.attention:
                h = mods[i](h); i += 1
            if lvl != 0 and self.aux:
                h = mods[i](h, temb_act, rng); i += 1
        assert not hs
        h = mods[i](h, silu=True); i += 1
        h = mods[i](ops.to_f32(h)); i += 1   # nf -> 3 channels: the skinny-output fp32 path either way
        assert i == len(mods)
        if self.config.model.scale_by_sigma:
            h = ops.scale_rows(h, 1.0 / used_sigmas.float())
        return h if nhwc else ops.to_nchw(h)

    def _h16_pack_weights(self):
        """bf16 packs of every convolution weight the H16 kernels can take, ONE launch per forward call (hip.H16Packer);
        each module finds its (fwd, dg) pair in `h16_pack` and hands it to ops.conv2d."""
        pk = self.__dict__.get("_h16_packer")
        if pk is None or not pk[0].valid_for(pk[2]):
            mods, ws, groups, attn = [], [], [], []
            for m in self.modules():
                if isinstance(m, AttnBlockpp):
                    C = m.NIN_0.W.shape[0]
                    if C % 128 == 0:
                        first = len(ws)
                        ws += [m.NIN_0.W, m.NIN_1.W, m.NIN_2.W, m.NIN_3.W]
                        groups.append((first, 3))
                        attn.append((m, first))
                    continue
                c = m
                if isinstance(m, PyramidDownsample):
                    c = m.Conv2d_0 if m.fir else m.Conv_0
                elif not isinstance(m, (Conv3x3, Conv1x1)):
                    continue
                w = c.weight
                ci, co = (w.shape[2], w.shape[3]) if w.dim() == 4 else (w.shape[0], w.shape[1])
                if ci % 32 == 0 and co % 2 == 0:
                    mods.append((c, len(ws)))
                    ws.append(w)
            if not ws:
                self.__dict__["_h16_packer"] = None
                return
            pk = (hip.H16Packer([w.detach() for w in ws], groups), (mods, attn), ws)
            self.__dict__["_h16_packer"] = pk
        views = pk[0].pack()
        mods, attn = pk[1]
        for c, i in mods:
            c.__dict__["h16_pack"] = views[i]
        for m, first in attn:
            m.__dict__["h16_pack"] = pk[0].group_views[first] + views[first + 3]

    # ------------------------------------------------------------------ checkpoint boundary
    def _layout(self, key, t, to_reference):
        if t.dim() == 4:  # conv weights: RSCK <-> OIHW
            return t.permute(3, 2, 0, 1).contiguous() if to_reference else t.permute(2, 3, 1, 0).contiguous()
        if key.endswith("Conv_2.weight"):  # 1x1: [I,O] <-> [O,I,1,1]
            return t.t().contiguous()[:, :, None, None] if to_reference else t[:, :, 0, 0].t().contiguous()
        return t

    def to_reference_layout(self, key, t):
        """Convert a tensor shaped like parameter `key` (e.g. its gradient / EMA shadow) to the reference layout."""
        return self._layout(key, t.detach(), True)

    def from_reference_layout(self, key, t):
        """Inverse of to_reference_layout (OIHW -> RSCK, [O,I,1,1] -> [I,O])."""
        t = torch.as_tensor(t)
        if t.dim() == 4 and key.endswith("Conv_2.weight"):
            return t[:, :, 0, 0].t().contiguous()
        if t.dim() == 4:
            return t.permute(2, 3, 1, 0).contiguous()
        return t

    def reference_state_dict(self, prefix="module."):
        """State dict in the reference's key/layout convention (models/utils.py:93 wraps in DataParallel)."""
        return {prefix + k: self._layout(k, v.detach(), True) for k, v in self.state_dict().items()}

    def load_reference_state_dict(self, sd):
        own = self.state_dict()
        new = {}
        for k, v in sd.items():
            k2 = k[len("module."):] if k.startswith("module.") else k
            if k2 not in own:
                raise KeyError(k2)
            v = torch.as_tensor(v)
            if v.dim() == 4 and k2.endswith("Conv_2.weight"):
                v = v[:, :, 0, 0].t().contiguous()
            elif v.dim() == 4:
                v = v.permute(2, 3, 1, 0).contiguous()
            new[k2] = v.to(own[k2].dtype)
        missing = set(own) - set(new)
        if missing:
            raise KeyError(f"missing keys: {sorted(missing)[:5]}")
        self.load_state_dict(new)
