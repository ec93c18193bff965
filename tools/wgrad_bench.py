"""Weight-grad dispatch experiments (INDM_EXP bit 1: conv wgrad big tile from K >= 4096; bit 2: 256x128 big tile for
N <= 256 wgrad GEMMs).  usage: INDM_EXP=0|1|2|3 python tools/wgrad_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip
dev = "cuda"
r = lambda *s: torch.randn(*s, device=dev)
def timeit(fn, flops, name, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:44s} {ms*1000:8.1f} us {flops/ms/1e9:7.1f} TF  {hip.lib().indm_igemm_last_variant().decode()}", flush=True)
B = 128
for (H, C, Co) in [(16, 256, 256), (8, 256, 256), (8, 512, 256), (4, 256, 256), (4, 512, 256), (16, 512, 256), (32, 128, 128), (32, 256, 128)]:
    x, dy = r(B, H, H, C), r(B, H, H, Co)
    timeit(lambda: hip.conv2d_wgrad(x, dy, 3, 3), 2 * B * H * H * 9 * C * Co, f"conv3x3 wgrad {H}x{H} {C}->{Co}")
for (M, K, N) in [(32768, 256, 256), (8192, 256, 256), (2048, 256, 256), (131072, 128, 128), (32768, 512, 256), (131072, 512, 512)]:
    a, b = r(M, K), r(M, N)
    timeit(lambda: hip.gemm(a, b, trans_a=True, splitk=True), 2.0 * M * K * N, f"gemm TN [{M},{K}]^T x [{M},{N}]")
