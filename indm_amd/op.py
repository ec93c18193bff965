"""Drop-in mirror of the reference's `op` package (op/__init__.py:1-2): `upfirdn2d`, `fused_leaky_relu`,
`FusedLeakyReLU` with the reference's NCHW call conventions and gradient relations, on the HIP kernels.

    from indm_amd.op import upfirdn2d, FusedLeakyReLU, fused_leaky_relu

reference: op/upfirdn2d.py:19-156 (UpFirDn2d / UpFirDn2dBackward, backward = same op with up<->down, flipped
kernel and g_pad; double backward = the forward op) and op/fused_act.py:20-97.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import hip


class UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        gx = hip.upfirdn2d(grad_output.contiguous(), grad_kernel, down[0], down[1], up[0], up[1], g_pad[0], g_pad[1],
                           g_pad[2], g_pad[3], 0)
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad, in_size, out_size)
        return gx.view(in_size)

    @staticmethod
    def backward(ctx, gradgrad_input):
        (kernel,) = ctx.saved_tensors
        up, down, pad, in_size, out_size = ctx.cfg
        ggo = hip.upfirdn2d(gradgrad_input.contiguous(), kernel, up[0], up[1], down[0], down[1], pad[0], pad[1], pad[2],
                            pad[3], 0)
        return ggo.view(in_size[0], in_size[1], out_size[0], out_size[1]), None, None, None, None, None, None, None, None


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        px0, px1, py0, py1 = pad
        kh, kw = kernel.shape
        _, _, in_h, in_w = input.shape
        ctx.in_size = input.shape
        out = hip.upfirdn2d(input.contiguous(), kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1, 0)
        out_h, out_w = out.shape[2], out.shape[3]
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]).contiguous())
        ctx.cfg = (up, down, pad, (out_h, out_w))
        ctx.g_pad = (kw - px0 - 1, in_w * up_x - out_w * down_x + px0 - up_x + 1,
                     kh - py0 - 1, in_h * up_y - out_h * down_y + py0 - up_y + 1)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        up, down, pad, out_size = ctx.cfg
        gx = UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, up, down, pad, ctx.g_pad, ctx.in_size, out_size)
        return gx, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """op/upfirdn2d.py:145-156 (always the native kernel: there is no CPU branch here)."""
    return UpFirDn2d.apply(input, kernel.to(input.device, torch.float32).contiguous(), (up, up), (down, down),
                           (pad[0], pad[1], pad[0], pad[1]))


class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.cfg = (negative_slope, scale)
        gi = hip.fused_bias_act(grad_output.contiguous(), None, out, 3, 1, negative_slope, scale)
        dim = [0] + list(range(2, gi.ndim))
        return gi, gi.sum(dim).detach()

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        (out,) = ctx.saved_tensors
        ns, scale = ctx.cfg
        return hip.fused_bias_act(gradgrad_input.contiguous(), gradgrad_bias, out, 3, 1, ns, scale), None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = hip.fused_bias_act(input.contiguous(), bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.cfg = (negative_slope, scale)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        gi, gb = FusedLeakyReLUFunctionBackward.apply(grad_output, out, *ctx.cfg)
        return gi, gb, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    """op/fused_act.py:86-97 (GPU branch: honours negative_slope like the CUDA kernel, fused_bias_act_kernel.cu:42)."""
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    """op/fused_act.py:74-84"""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
