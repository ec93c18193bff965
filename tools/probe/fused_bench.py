"""The round-6 fused F16 launches against the launches they replace (M = 131072, C = 3 and M = 32768, C = 12): us per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from indm_amd import hip
dev = "cuda"
def timeit(fn, n=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, H, C) in ((128, 32, 3), (128, 16, 12)):
    M, J = N * H * H, 9 * C
    x = torch.randn(N, H, H, C, device=dev)
    w = torch.randn(J, 512, device=dev) * 0.2
    f16 = lambda: (torch.randn(M * 512, device=dev) * 0.5).to(torch.bfloat16)  # noqa: E731
    d2, d1 = f16(), f16()
    Wp = hip.pack_weight_bf16(torch.randn(512, 512, device=dev) * 0.05)
    Pp = hip.pack_weight_bf16(torch.randn(512, 32 if J <= 32 else 128, device=dev) * 0.1)
    Wc = hip.pack_thin_conv_bf16(w)
    u2 = hip.f16_thin_conv(x, w, 1, hip.F16_GATE, gate=d2)[0]
    rows = [
        ("thin conv gate", lambda: hip.f16_thin_conv(x, w, 1, hip.F16_GATE, gate=d2)),
        ("chain", lambda: hip.f16_rows(hip.F16_CHAIN, u2, M, Wp=Wp, D1=d1, Pp=Pp, J=J)),
        ("chain_conv (fused)", lambda: hip.f16_chain_conv(x, Wc, d2, Wp, d1, Pp)),
        ("thin conv gate + raw", lambda: hip.f16_thin_conv(x, w, 1, hip.F16_GATE, gate=d2, want_raw=True)),
        ("chain + t2", lambda: hip.f16_rows(hip.F16_CHAIN, u2, M, Wp=Wp, D1=d1, Pp=Pp, J=J, want_t2=True)),
        ("chain_conv + t3, t2 (fused)", lambda: hip.f16_chain_conv(x, Wc, d2, Wp, d1, Pp, want_t=True)),
        ("gemm", lambda: hip.f16_rows(hip.F16_GEMM, u2, M, Wp=Wp)),
        ("conv_gemm (fused: raw + gated + out)", lambda: hip.f16_conv_gemm(x, Wc, d1, Wp)),
    ]
    print(f"--- M = {M}, C = {C}")
    import statistics
    res = {n: [] for n, _ in rows}
    for rep in range(3):
        for n, fn in rows:
            res[n].append(timeit(fn))
    for n, _ in rows:
        print(f"  {n:40s} {statistics.median(res[n]):8.1f} us")
