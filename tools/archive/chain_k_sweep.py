"""Chain kernel time vs K (slabs per tile) at M = 131072: separates the per-tile fixed cost (prologue + gate loads +
stage 2 + reduction + store) from the per-slab cost of the main loop.  usage (GPU box): python tools/chain_k_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip  # noqa: E402

dev = "cuda"
M, J = 131072, 27
d1 = torch.randn(M, 512, device=dev)
w1p = torch.zeros(512, 32, device=dev)
w1p[:, :J] = torch.randn(512, J, device=dev) / 22.6
res = []
for K in (64, 128, 256, 512, 1024):
    t3, d2 = torch.randn(M, K, device=dev), torch.randn(M, K, device=dev)
    w2t = torch.randn(K, 512, device=dev) / 22.6
    for _ in range(5):
        hip.ires_chain(t3, d2, w2t, d1, w1p, J)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        hip.ires_chain(t3, d2, w2t, d1, w1p, J)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    res.append((K, us))
    print(f"K={K:5d}: {us:7.1f} us   {2.0 * M * 512 * (K + J) / us / 1e6:6.1f} TF")
(k0, t0), (k1, t1) = res[-2], res[-1]
slope = (t1 - t0) / (k1 - k0)
print(f"per 16-deep slab: {slope * 16:.2f} us over the whole grid; fixed per launch (K -> 0): {t0 - slope * k0:.1f} us; "
      f"pure-MFMA time of K=512 at 154.7 TF: {2.0 * M * 512 * 512 / 154.7e6:.1f} us")
