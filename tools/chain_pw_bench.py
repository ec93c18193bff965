"""bf16 chain kernel: LDS-image weights vs pre-packed register weights (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip
hip.set_compute_dtype("bf16")
def timeit(fn, name, nbytes, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:60s} {ms*1000:8.1f} us  {nbytes/ms/1e9:6.2f} TB/s", flush=True)
r = lambda *s: torch.randn(*s, device="cuda")
for M, J in [(131072, 27), (32768, 108)]:
    JP = 32 if J <= 32 else 128
    t3, d2, d1 = r(M, 512), r(M, 512), r(M, 512)
    w2t = r(512, 512)
    w1p = torch.zeros(512, JP, device="cuda"); w1p[:, :J] = r(512, J)
    nb = 12.0 * M * 512
    for t2 in (False, True):
        hip._w2t_packed.clear()
        timeit(lambda: hip.ires_chain(t3, d2, w2t, d1, w1p, J, want_t2=t2), f"chain bf16 LDS weights   M={M} J={J} t2={t2}", nb + (4.0 * M * 512 if t2 else 0))
        hip.pack_w2t_bf16([w2t])
        timeit(lambda: hip.ires_chain(t3, d2, w2t, d1, w1p, J, want_t2=t2), f"chain bf16 packed weights M={M} J={J} t2={t2}", nb + (4.0 * M * 512 if t2 else 0))
