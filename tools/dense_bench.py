"""Micro-benchmark of the igemm engine on the shapes of the INDM step (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip

dev = "cuda"
if os.environ.get("INDM_DENSE_BENCH_DTYPE"):   # "bf16": the same shapes with bf16 MFMA inputs / fp32 accumulate
    hip.set_compute_dtype(os.environ["INDM_DENSE_BENCH_DTYPE"])
def timeit(fn, flops, name, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:58s} {ms*1000:9.1f} us  {flops/ms/1e9:7.2f} TF/s", flush=True)

B = 128
r = lambda *s: torch.randn(*s, device=dev)
M = B * 1024
a, w = r(M, 512), r(512, 512)
timeit(lambda: hip.gemm(a, w), 2*M*512*512, "gemm NN  [131072,512]x[512,512] (flow 1x1 fwd)")
timeit(lambda: hip.gemm(a, w, trans_b=True), 2*M*512*512, "gemm NT  (flow 1x1 transposed / dgrad)")
ga = r(M, 512)
timeit(lambda: hip.gemm(a, w, trans_b=True, gate=ga), 2*M*512*512, "gemm NT gated (VJP chain)")
timeit(lambda: hip.gemm(a, w, gate=ga), 2*M*512*512, "gemm NN gated")
timeit(lambda: hip.gemm(a, a, trans_a=True, splitk=True), 2*M*512*512, "gemm TN  split-K (1x1 wgrad)")
a4, w4 = r(4096, 4096), r(4096, 4096)
timeit(lambda: hip.gemm(a4, w4), 2*4096**3, "gemm NN  4096^3 (reference point: guide's untuned kernel = 122 TF)")
timeit(lambda: hip.gemm(a4, w4, trans_b=True), 2*4096**3, "gemm NT  4096^3")
M1 = B * 256
a1 = r(M1, 512)
timeit(lambda: hip.gemm(a1, w), 2*M1*512*512, "gemm NN  [32768,512]x[512,512] (flow scale 1)")
for (H, C, Co) in [(32, 128, 128), (16, 256, 256), (16, 512, 256), (8, 256, 256), (8, 512, 256), (4, 256, 256), (32, 256, 128), (32, 384, 128)]:
    x, wt = r(B, H, H, C), r(3, 3, C, Co)
    fl = 2 * B * H * H * 9 * C * Co
    timeit(lambda: hip.conv2d(x, wt), fl, f"conv3x3 fwd   {H}x{H} {C}->{Co}")
    dy = r(B, H, H, Co)
    timeit(lambda: hip.conv2d_dgrad(dy, wt, (H, H)), fl, f"conv3x3 dgrad {H}x{H} {C}->{Co}")
    timeit(lambda: hip.conv2d_wgrad(x, dy, 3, 3), fl, f"conv3x3 wgrad {H}x{H} {C}->{Co}")
x3, w3 = r(B, 32, 32, 3), r(3, 3, 3, 512)
timeit(lambda: hip.conv2d(x3, w3), 2*M*27*512, "conv3x3 fwd   32x32 3->512 (spread+gemm)")
x5, w5 = r(B, 32, 32, 512), r(3, 3, 512, 3)
timeit(lambda: hip.conv2d(x5, w5), 2*M*27*512, "conv3x3 fwd   32x32 512->3 (gemm+gather)")
d3 = r(B, 32, 32, 3)
timeit(lambda: hip.conv2d_dgrad(d3, w5, (32, 32)), 2*M*27*512, "conv3x3 dgrad 32x32 (3 -> 512) (spread+gemm)")
timeit(lambda: hip.conv2d_dgrad(x5, w3, (32, 32)), 2*M*27*512, "conv3x3 dgrad 32x32 (512 -> 3) (gemm+gather)")
timeit(lambda: hip.conv2d_wgrad(x5, d3, 3, 3), 2*M*27*512, "conv3x3 wgrad 32x32 512->3")
timeit(lambda: hip.conv2d_wgrad(x3, x5, 3, 3), 2*M*27*512, "conv3x3 wgrad 32x32 3->512")
q, k = r(B, 256, 256), r(B, 256, 256)
timeit(lambda: hip.gemm(q, k, trans_b=True), 2*B*256*256*256, "attention QK^T batched [128,256,256]")
g = r(B, 32, 32, 128); gm, bt = r(128), r(128)
timeit(lambda: hip.groupnorm_fwd(g, gm, bt, 32, 1e-6, 1), 1, "groupnorm+silu fwd [128,32,32,128] (64 MB r + 64 MB w)")
p = r(M, 512)
timeit(lambda: hip.sincos_fwd(p), 1, "sincos fwd [131072,512] (256 MB r, 512 MB w)")
timeit(lambda: hip.mul(p, p), 1, "mul [131072,512] (512 MB r, 256 MB w)")
timeit(lambda: hip.colreduce(p, None, 0), 1, "colreduce [131072,512]")
print("--- HBM-bound helpers at the flow's scale-1 size [131072, 512] (268 MB per tensor)")
def bw(fn, nbytes, name, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:58s} {ms*1000:9.1f} us  {nbytes/ms/1e9:7.2f} TB/s", flush=True)
T = 4 * M * 512
q = r(M, 512)
bw(lambda: hip.sincos_fwd(p, True), 3 * T, "sincos fwd (1 read, 2 writes)")
bw(lambda: hip.sincos_bwd(p, q, p, q), 5 * T, "sincos bwd (4 reads, 1 write)")
bw(lambda: hip.mul(p, q), 3 * T, "mul")
bw(lambda: hip.axpby(p, q, 1.0, -1.0), 3 * T, "axpby")
bw(lambda: hip.colreduce(p, None, 0), T, "colreduce")
v3 = r(B, 32, 32, 3)
bw(lambda: hip.tap_spread(v3, -1), 4 * M * 30, "tap_spread 3 -> 27 ch")
P27 = r(B, 32, 32, 27)
bw(lambda: hip.tap_gather(P27, None, 1), 4 * M * 30, "tap_gather 27 -> 3 ch")
bw(lambda: hip.rowreduce(v3, v3, 0, 1.0), 2 * 4 * M * 3, "rowreduce [128, 3072]")
g5 = r(B, 32, 32, 128); gm, bt = r(128), r(128)
bw(lambda: hip.groupnorm_fwd(g5, gm, bt, 32, 1e-6, 1), 3 * g5.numel() * 4, "groupnorm+silu fwd [128,32,32,128]")
