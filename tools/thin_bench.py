"""Micro-benchmark of the thin (HBM-streaming) GEMM shapes of the flow's 3-/12-channel edges (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip

dev = "cuda"
def timeit(fn, nbytes, flops, name, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:64s} {ms*1000:8.1f} us  {nbytes/ms/1e9:6.2f} TB/s  {flops/ms/1e9:7.2f} TF/s", flush=True)

r = lambda *s: torch.randn(*s, device=dev)
for (M, C) in [(131072, 3), (32768, 12)]:
    K9 = 9 * C
    wide, gate = r(M, 512), r(M, 512)
    thin = r(M, K9)
    w_kn, w_nk = r(K9, 512), r(K9, 512)
    cb, bias = r(M // 1024 if C == 3 else M // 256, 512), r(512)
    fl = 2.0 * M * K9 * 512
    WB = 4.0 * M * 512
    timeit(lambda: hip.gemm(thin, w_kn), WB, fl, f"thin-K  [{M},{K9}]x[{K9},512]")
    timeit(lambda: hip.gemm(thin, w_kn, bias=bias, bias2=cb, rows_per_group=M // cb.shape[0]), WB, fl,
           f"thin-K  [{M},{K9}]x[{K9},512] + bias + bias2")
    timeit(lambda: hip.gemm(wide, w_nk, trans_b=True), WB, fl, f"thin-N  [{M},512]x[{K9},512]^T")
    timeit(lambda: hip.gemm(wide, w_nk, trans_b=True, gate=gate), 2 * WB, fl, f"thin-N  gated")
    w_kn2 = r(512, K9)
    timeit(lambda: hip.gemm(wide, w_kn2), WB, fl, f"thin-N  [{M},512]x[512,{K9}]")
    timeit(lambda: hip.gemm(thin, wide, trans_a=True, splitk=True), WB, fl, f"wgrad   [{K9},{M}]x[{M},512] (thin^T wide)")
    timeit(lambda: hip.gemm(wide, thin, trans_a=True, splitk=True), WB, fl, f"wgrad   [512,{M}]x[{M},{K9}] (wide^T thin)")
    timeit(lambda: hip.gemm(wide, thin, trans_a=True, splitk=True, gate=gate), 2 * WB, fl, f"wgrad   (wide.gate)^T thin")
a, b = r(32768, 256), r(32768, 256)
timeit(lambda: hip.gemm(a, b, trans_a=True, splitk=True), 8.0 * 32768 * 256, 2.0 * 32768 * 256 * 256, "wgrad   [256,32768]x[32768,256] (attention NIN)")
g3 = r(128, 1024, 512)
ones = torch.ones(1, 1024, device=dev)
timeit(lambda: hip.gemm(ones, g3), 4.0 * g3.numel(), 2.0 * g3.numel(), "dbias2  ones[1,1024] x [128,1024,512] (batched)")
x = r(131072, 512)
timeit(lambda: x.fill_(1.0), 4.0 * x.numel(), 1, "torch fill_ [131072,512] (pure write)")
timeit(lambda: x.zero_(), 4.0 * x.numel(), 1, "torch zero_ [131072,512] (pure write)")
y = torch.empty_like(x)
timeit(lambda: y.copy_(x), 8.0 * x.numel(), 1, "torch copy_ (1 read 1 write)")
timeit(lambda: hip.axpby(x, None, 2.0, 0.0), 8.0 * x.numel(), 1, "axpby scale (1 read 1 write)")
timeit(lambda: hip.colreduce(x, None, 0), 4.0 * x.numel(), 1, "colreduce [131072,512]")
for (M, N, K) in [(128, 512, 256), (128, 256, 512), (128, 512, 128)]:
    a, b = r(M, K), r(K, N)
    timeit(lambda: hip.gemm(a, b), 4.0 * (M * K + K * N + M * N), 2.0 * M * N * K, f"small   [{M},{K}]x[{K},{N}]")
x, y1, y2 = r(131072, 512), r(131072, 512), r(131072, 512)
timeit(lambda: hip.mul(x, y1), 12.0 * x.numel(), 1, "mul (2 reads 1 write)")
timeit(lambda: hip.mul2(x, y1, y2), 20.0 * x.numel(), 1, "mul2 (3 reads 2 writes)")
timeit(lambda: hip.sincos_fwd(x), 12.0 * x.numel(), 1, "sincos fwd (1 read 2 writes)")
timeit(lambda: hip.axpby(x, y1, 1.0, 1.0), 12.0 * x.numel(), 1, "axpby (2 reads 1 write)")
