"""conv + Sin: the fused launch (indm_conv3x3_thin_act_*) against the two-step form, per geometry (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip

def t(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (N, H, C) in [(128, 32, 3), (128, 16, 12), (32, 32, 12), (16, 32, 3), (16, 16, 12)]:
    x = torch.randn(N, H, H, C, device="cuda")
    w = torch.randn(9 * C, 512, device="cuda") * 0.3
    b = torch.randn(512, device="cuda")
    conv = t(lambda: hip.conv3x3_thin(x, w, -1, 1.0, b))
    p = hip.conv3x3_thin(x, w, -1, 1.0, b)
    act = t(lambda: hip.sincos_fwd(p))
    fused = t(lambda: hip.conv3x3_thin_act(x, w, -1, b))
    print(f"{N:4d} x {H}x{H} x {C:2d} -> 512: conv {conv:7.1f} us + Sin {act:7.1f} us = {conv + act:7.1f} us   fused {fused:7.1f} us", flush=True)
