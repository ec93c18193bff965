"""Back-to-back VJP-chain kernel vs the two separate launches it replaces, at the batch-128 production shapes.
usage (GPU box): python tools/chain_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip  # noqa: E402

dev = "cuda"


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


for M, J in ((131072, 27), (32768, 108)):
    t3, d2, d1 = (torch.randn(M, 512, device=dev) for _ in range(3))
    w2t = torch.randn(512, 512, device=dev) / 22.6
    JP = 32 if J <= 32 else 128
    w1p = torch.zeros(512, JP, device=dev)
    w1p[:, :J] = torch.randn(512, J, device=dev) / 22.6
    w1 = w1p[:, :J].t().contiguous()                       # [J, 512]: the trans_b form of the separate path

    def separate():
        t2 = hip.gemm(t3, w2t, gate=d2)
        return hip.gemm(t2, w1, trans_b=True, gate=d1)

    def fused():
        return hip.ires_chain(t3, d2, w2t, d1, w1p, J)

    def fused_t2():
        return hip.ires_chain(t3, d2, w2t, d1, w1p, J, want_t2=True)

    a, b = separate(), fused()
    err = (a - b).abs().max().item() / a.abs().max().item()
    us_s, us_f = timeit(separate), timeit(fused)
    print(f"   fused + t2 store: {timeit(fused_t2):.0f} us")
    fl = 2.0 * M * 512 * (512 + J)
    print(f"M={M} J={J}: separate {us_s:.0f} us ({fl / us_s / 1e6:.1f} TF)  fused {us_f:.0f} us ({fl / us_f / 1e6:.1f} TF)  "
          f"rel diff {err:.2e}")
