"""Per-kernel timings of the bf16-storage (F16) flow kernels at the production shapes of BASELINE config #2's batch 128
(scale 0: M = 131072 rows, C = 3, J = 27; scale 1: M = 32768, C = 12, J = 108) with the bytes each launch has to move.
usage: python tools/f16_bench.py [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda"


def timeit(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, C in ((32, 3), (16, 12)):
    M, J = B * H * H, 9 * C
    f16 = lambda: (torch.randn(M * 512, device=dev) * 0.5).to(torch.bfloat16)  # noqa: E731
    X16, G16, Y16 = f16(), f16(), f16()
    x = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(9 * C, 512, device=dev) * 0.1
    b = torch.randn(512, device=dev)
    b2 = torch.randn(B, 512, device=dev)
    Wp = hip.pack_weight_bf16(torch.randn(512, 512, device=dev) * 0.05)
    Pp = hip.pack_weight_bf16(torch.randn(512, 32 if J <= 32 else 128, device=dev) * 0.1)
    thin = torch.randn(M, J, device=dev)
    mb = M * 512 * 2 / 1e6   # one F16 tensor in MB
    rows = [
        ("thin conv plain      (w 1)", lambda: hip.f16_thin_conv(x, w, -1, hip.F16_PLAIN, bias=b), 1),
        ("thin conv act        (w 2)", lambda: hip.f16_thin_conv(x, w, -1, hip.F16_ACT, bias=b), 2),
        ("thin conv gate       (r 1 w 1)", lambda: hip.f16_thin_conv(x, w, 1, hip.F16_GATE, gate=G16), 2),
        ("thin conv gate + raw (r 1 w 2)", lambda: hip.f16_thin_conv(x, w, 1, hip.F16_GATE, gate=G16, want_raw=True), 3),
        ("rows gemm            (r 1 w 1)", lambda: hip.f16_rows(hip.F16_GEMM, X16, M, Wp=Wp), 2),
        ("rows gemm + act      (r 1 w 2)", lambda: hip.f16_rows(hip.F16_GEMM_ACT, X16, M, Wp=Wp, bias=b, bias2=b2, rows_per_group=H * H), 3),
        ("rows chain           (r 2)", lambda: hip.f16_rows(hip.F16_CHAIN, X16, M, Wp=Wp, D1=G16, Pp=Pp, J=J), 2),
        ("rows chain + t2      (r 2 w 1)", lambda: hip.f16_rows(hip.F16_CHAIN, X16, M, Wp=Wp, D1=G16, Pp=Pp, J=J, want_t2=True), 3),
        ("rows thin-N          (r 1)", lambda: hip.f16_rows(hip.F16_THINN, X16, M, Pp=Pp, J=J), 1),
        ("bwd_act + seg        (r 5 w 1)", lambda: hip.f16_bwd_act(X16, G16, Y16, X16, G16, M, H * H), 6),
        ("thin wgrad           (r 1)", lambda: hip.f16_thin_wgrad(thin, X16, M), 1),
        ("thin wgrad gated     (r 2)", lambda: hip.f16_thin_wgrad(thin, X16, M, gate=G16), 2),
        ("wgrad gemm           (r 2)", lambda: hip.f16_wgrad_gemm(X16, Y16, M), 2),
        ("wgrad gemm gated     (r 3)", lambda: hip.f16_wgrad_gemm(X16, Y16, M, gate=G16), 3),
    ]
    print(f"--- batch {B}, {H}x{H}, C = {C}: M = {M}, one F16 [M, 512] tensor = {mb:.0f} MB")
    for name, fn, ntens in rows:
        us = timeit(fn)
        print(f"  {name:34s} {us:8.1f} us   {ntens * mb / us / 1e3 * 1e3:6.2f} TB/s of the {ntens} wide tensor(s)")
