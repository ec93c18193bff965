import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip
def timeit(fn, flops, name, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:58s} {ms*1000:9.1f} us  {flops/ms/1e9:7.2f} TF/s", flush=True)
r = lambda *s: torch.randn(*s, device="cuda")
for M in (131072, 32768):
    a, dc, g = r(M, 512), r(M, 512), r(M, 512)
    fl = 2 * M * 512 * 512
    timeit(lambda: hip.gemm(a, dc, True, False, splitk=True), fl, f"wgrad 1x1 K={M}")
    timeit(lambda: hip.gemm(a, dc, True, False, splitk=True, gate=g), fl, f"wgrad 1x1 gated K={M}")

    timeit(lambda: hip.gemm(a, dc[:512].contiguous(), False, True), fl, f"fwd NT M={M}")
