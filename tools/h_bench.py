"""bf16-storage kernels at the batch-128 production shapes: achieved HBM bandwidth.  usage (GPU box): python tools/h_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip
dev = "cuda"
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 131072
h = lambda *s: torch.randn(*s, device=dev).bfloat16()
a, g, d1 = h(M, 512), h(M, 512), h(M, 512)
w = (torch.randn(512, 512, device=dev) / 22.6).bfloat16()
w1p = torch.zeros(512, 32, device=dev); w1p[:, :27] = torch.randn(512, 27, device=dev) / 22.6
for name, fn, nb in [
    ("hgemm plain  -> bf16", lambda: hip.hgemm(a, w), 2 * M * 512 * 2),
    ("hgemm gated  -> bf16", lambda: hip.hgemm(a, w, gate=g), 2 * M * 512 * 3),
    ("hgemm plain  -> fp32 N=27", lambda: hip.hgemm(a, w[:27].contiguous(), out_bf16=False), 2 * M * 512 + 4 * M * 27),
    ("hchain", lambda: hip.hchain(a, g, w, d1, w1p, 27), 2 * M * 512 * 3 + 4 * M * 27),
    ("hchain + t2", lambda: hip.hchain(a, g, w, d1, w1p, 27, want_t2=True), 2 * M * 512 * 4 + 4 * M * 27),
    ("sincos_h", lambda: hip.sincos_fwd_h(a), 2 * M * 512 * 3),
    ("ires_bwd_act_h (dt, seg)", lambda: hip.ires_bwd_act_h(a, g, d1, a, g, True, 1024), 2 * M * 512 * 7),
]:
    us = timeit(fn)
    print(f"{name:30s} {us:8.1f} us  {nb / us / 1e6:6.2f} TB/s", flush=True)
