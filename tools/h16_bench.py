"""Time the H16 (bf16 storage) convolution kernels against the bf16-compute / fp32-storage kernels of rounds 2-5 at the score
network's shapes.  usage: python tools/h16_bench.py [batch]   (env INDM_H16_* tune the tile / split rules)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda")
SHAPES = [(32, 128, 128, 3), (32, 256, 128, 3), (32, 384, 128, 3), (16, 128, 256, 3), (16, 256, 256, 3), (16, 512, 256, 3),
          (8, 256, 256, 3), (8, 512, 256, 3), (4, 256, 256, 3), (4, 512, 256, 3), (16, 256, 256, 1), (32, 256, 128, 1)]


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


print(f"batch {B}: us per launch   [H16 | fp32-storage bf16-MFMA]   (TF = H16)")
hip.set_compute_dtype("bf16")
for S, Ci, Co, K in SHAPES:
    pad = K // 2
    x32 = torch.randn(B, S, S, Ci, device=dev)
    dy32 = torch.randn(B, S, S, Co, device=dev)
    w = torch.randn(K, K, Ci, Co, device=dev) * 0.05
    x16, dy16 = x32.to(torch.bfloat16), dy32.to(torch.bfloat16)
    (fwd, dg), = hip.h16_pack_weights([w])
    fl = 2.0 * B * S * S * Ci * Co * K * K
    t_f = timeit(lambda: hip.h16_conv2d(x16, fwd, K, K, 1, pad))
    t_d = timeit(lambda: hip.h16_conv2d(dy16, dg, K, K, 1, K - 1 - pad))
    t_w = timeit(lambda: hip.h16_conv2d_wgrad(x16, dy16, K, K, 1, pad))
    if K == 3:
        o_f = timeit(lambda: hip.conv2d(x32, w, 1, pad))
        o_d = timeit(lambda: hip.conv2d_dgrad(dy32, w, (S, S), 1, pad))
        o_w = timeit(lambda: hip.conv2d_wgrad(x32, dy32, K, K, 1, pad))
    else:
        w2 = w.reshape(Ci, Co)
        o_f = timeit(lambda: hip.gemm(x32.view(-1, Ci), w2))
        o_d = timeit(lambda: hip.gemm(dy32.view(-1, Co), w2, trans_b=True))
        o_w = timeit(lambda: hip.gemm(x32.view(-1, Ci), dy32.view(-1, Co), trans_a=True, splitk=True))
    print(f"{S:3d}x{S:<3d} {Ci:4d}->{Co:<4d} k{K}: fwd {t_f:7.1f} | {o_f:7.1f} ({fl / t_f / 1e6:6.0f} TF)   dgrad {t_d:7.1f} | {o_d:7.1f} "
          f"({fl / t_d / 1e6:6.0f} TF)   wgrad {t_w:7.1f} | {o_w:7.1f} ({fl / t_w / 1e6:6.0f} TF)", flush=True)
