"""The flow's Sin activation inside the GEMM / thin-convolution epilogues (indm_gemm_sin_f32, indm_conv3x3_thin_act_f32) and as its own
pass, per launch at batch 128 -- run once per library (INDM_HIP_LIB=... INDM_HIP_LIB_PARTIAL=1 for an older build) to A/B the
branch-free indm_sincos of round 6 against libm's sincosf.  Prints us per launch and the GEMM's MFMA fraction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip


def t(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("library:", os.environ.get("INDM_HIP_LIB", "in-tree"))
for M in (131072, 32768):
    a = torch.randn(M, 512, device="cuda") * 0.2
    w = torch.randn(512, 512, device="cuda") * 0.05
    b = torch.randn(512, device="cuda") * 0.1
    plain = t(lambda: hip.gemm(a, w, trans_b=True, bias=b))
    fused = t(lambda: hip.gemm_sin(a, w, b))
    p = hip.gemm(a, w, trans_b=True, bias=b)
    act = t(lambda: hip.sincos_fwd(p))
    fl = 2.0 * M * 512 * 512
    print(f"GEMM [{M}, 512] x [512, 512]^T: plain {plain:7.1f} us ({fl / plain / 1e6 / 157.3:4.2f})   + Sin epilogue {fused:7.1f} us "
          f"({fl / fused / 1e6 / 157.3:4.2f} of the fp32 MFMA peak)   Sin as its own pass {act:7.1f} us", flush=True)
for (N, H, C) in [(128, 32, 3), (128, 16, 12)]:
    x = torch.randn(N, H, H, C, device="cuda")
    w = torch.randn(9 * C, 512, device="cuda") * 0.3
    b = torch.randn(512, device="cuda")
    conv = t(lambda: hip.conv3x3_thin(x, w, -1, 1.0, b))
    fused = t(lambda: hip.conv3x3_thin_act(x, w, -1, b))
    nbytes = 4.0 * N * H * H * 512 * 2
    print(f"thin conv {N} x {H}x{H} x {C:2d} -> 512: plain {conv:7.1f} us   + Sin epilogue {fused:7.1f} us = {nbytes / fused / 1e6:4.2f} TB/s of (a, d) written",
          flush=True)
