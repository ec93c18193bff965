"""Fused attention vs the batched-GEMM + softmax composition at the benchmark's shape (B=128, 16x16 tokens, C=256).
usage (GPU box): python tools/attn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip, ops  # noqa: E402

dev = "cuda"
B, T, C = 128, 256, 256
q, k, v, dh = (torch.randn(B, T, C, device=dev) for _ in range(4))
scale = C ** -0.5


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def unfused_fwd():
    w = hip.softmax_fwd(hip.gemm(q, k, trans_b=True), scale)
    return hip.gemm(w, v), w


def unfused_bwd(w):
    dv = hip.gemm(w, dh, trans_a=True)
    dw = hip.softmax_bwd(hip.gemm(dh, v, trans_b=True), w, scale)
    return hip.gemm(dw, k), hip.gemm(dw, q, trans_a=True), dv


h, lse = hip.attn_fwd(q, k, v, scale)
_, w = unfused_fwd()
fl = 4.0 * B * T * T * C
print(f"fwd: fused {timeit(lambda: hip.attn_fwd(q, k, v, scale)):.0f} us   unfused {timeit(unfused_fwd):.0f} us   "
      f"({fl / 1e9:.1f} GFLOP)")
print(f"bwd: fused {timeit(lambda: hip.attn_bwd(q, k, v, h, dh, lse, scale)):.0f} us   "
      f"unfused {timeit(lambda: unfused_bwd(w)):.0f} us   ({2.5 * fl / 1e9:.1f} GFLOP)")
