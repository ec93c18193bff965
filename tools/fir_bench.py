"""upfirdn2d (NHWC) at the VE config's batch-128 shapes: achieved HBM bandwidth (read x once + write y once).
usage (GPU box): python tools/fir_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip
dev = "cuda"
k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
k = (torch.outer(k1, k1) / 64).to(dev)
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (H, C, up, down, p0, p1) in [(16, 256, 2, 1, 2, 1), (32, 128, 1, 2, 1, 1), (32, 3, 1, 2, 1, 1), (8, 256, 2, 1, 2, 1), (16, 256, 1, 2, 1, 1),
                                 (32, 128, 1, 1, 2, 1)]:
    x = torch.randn(128, H, H, C, device=dev)
    y = hip.upfirdn2d(x, k * up * up, up, up, down, down, p0, p1, p0, p1, 1)
    us = timeit(lambda: hip.upfirdn2d(x, k * up * up, up, up, down, down, p0, p1, p0, p1, 1))
    nb = 4 * (x.numel() + y.numel())
    print(f"H={H} C={C} up={up} down={down}: out {tuple(y.shape)}  {us:7.1f} us  {nb / us / 1e6:6.2f} TB/s", flush=True)
