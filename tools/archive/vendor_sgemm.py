"""Reference point only (not on the product path): what the vendor fp32 GEMM (torch.matmul -> rocBLAS/hipBLASLt)
reaches on the shapes of the dominant igemm classes, next to indm_gemm_f32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip
torch.backends.cuda.matmul.allow_tf32 = False


def bench(f, flops, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return flops * n / (time.time() - t0) / 1e12


for (M, N, K) in ((131072, 512, 512), (32768, 512, 512), (4096, 4096, 4096), (131072, 128, 1152)):
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(K, N, device="cuda")
    bt = b.t().contiguous()
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}:  vendor NN {bench(lambda: torch.matmul(a, b), fl):6.1f}  vendor NT "
          f"{bench(lambda: torch.matmul(a, bt.t()), fl):6.1f}  igemm NN {bench(lambda: hip.gemm(a, b), fl):6.1f}  "
          f"igemm NT {bench(lambda: hip.gemm(a, bt, trans_b=True), fl):6.1f}  TF/s", flush=True)
