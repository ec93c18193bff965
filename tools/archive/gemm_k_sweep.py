"""Plain / gated GEMM time vs K at M = 131072, N = 512 (256 x 128 direct-to-LDS tiles, 2048 tiles = 4 rounds of 512):
per-slab cost vs per-tile fixed cost (workgroup launch, prologue, first copy, store epilogue).  usage (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip  # noqa: E402

dev = "cuda"
M, N = 131072, 512
for gated in (False, True):
    res = []
    for K in (64, 128, 256, 512, 1024, 2048):
        a, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        g = torch.randn(M, K, device=dev) if gated else None
        fn = (lambda: hip.gemm(a, w, trans_b=True, gate=g))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res.append((K, us))
        print(f"{'gated' if gated else 'plain'} NT K={K:5d}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF   ({hip.lib().indm_igemm_last_variant().decode()})")
    (k0, t0), (k1, t1) = res[-2], res[-1]
    slope = (t1 - t0) / (k1 - k0)
    print(f"   per 16-deep slab {slope * 16:.2f} us over the grid; fixed per launch (K -> 0) {t0 - slope * k0:.1f} us; "
          f"pure-MFMA slab time at 154.7 TF: {2.0 * M * N * 16 / 154.7e6:.2f} us")
