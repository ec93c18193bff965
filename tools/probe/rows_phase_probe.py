"""Which phase bounds f16_rows_kernel's GEMM mode at M = 131072: debug bits passed through the (unused) J argument of a PROBE build
(bit 0: no stores, bit 1: weight fragments fetched once, bit 2: no A copies, bit 3: no LDS reads).  Not part of the library's contract."""
import os
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from indm_amd import hip
M = 131072
dev = "cuda"
X = (torch.randn(M * 512, device=dev) * 0.5).to(torch.bfloat16)
Wp = hip.pack_weight_bf16(torch.randn(512, 512, device=dev) * 0.05)
def timeit(fn, n=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
import statistics
CASES = (0, 1, 2, 4, 8, 7, 15, 14, 13, 11)
res = {b: [] for b in CASES}
for rep in range(5):          # interleaved repetitions: the same launch varies by +- 10 % between batches of 30
    for bits in CASES:
        res[bits].append(timeit(lambda: hip.f16_rows(hip.F16_GEMM, X, M, Wp=Wp, J=bits)))
for bits in CASES:
    us = statistics.median(res[bits])
    print(f"no-stores {bits & 1}  weights-once {(bits >> 1) & 1}  no-copies {(bits >> 2) & 1}  no-lds-reads {(bits >> 3) & 1}: {us:7.1f} us")
