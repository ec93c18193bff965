import os, sys
sys.path.insert(0, "/root/repo")
import torch
from indm_amd import hip
dev = "cuda"
r = lambda *s: torch.randn(*s, device=dev)
def timeit(fn, name, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1)/iters*1000:8.1f} us  {hip.lib().indm_igemm_last_variant().decode()}", flush=True)
hip.set_compute_dtype("bf16")
for M in (131072, 32768):
    a, g, b = r(M, 512), r(M, 512), r(M, 512)
    timeit(lambda: hip.gemm(a, b, trans_a=True, splitk=True), f"TN plain M={M}")
    timeit(lambda: hip.gemm(a, b, trans_a=True, splitk=True, gate=g), f"TN gated M={M}")
M = 131072
a, b = r(M, 512), r(M, 512)
timeit(lambda: hip.gemm(a, b, trans_a=True, splitk=True, gate=a), "TN gated, gate aliases A (no extra bytes)")
