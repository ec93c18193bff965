"""bf16 chain kernel: fp32 LDS image of W2T vs the pre-packed bf16 fragment-order copy (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip
def timeit(fn, name, flops, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:64s} {ms*1000:8.1f} us  {flops/ms/1e9:7.1f} TF/s", flush=True)
r = lambda *s: torch.randn(*s, device="cuda")
for dtype in ("bf16",):
    hip.set_compute_dtype(dtype)
    for M, J in [(131072, 27), (32768, 108)]:
        JP = 32 if J <= 32 else 128
        t3, d2, d1 = r(M, 512), r(M, 512), r(M, 512)
        w2t = r(512, 512)
        w1p = torch.zeros(512, JP, device="cuda"); w1p[:, :J] = r(512, J)
        fl = 2.0 * M * 512 * (512 + J)
        for t2 in (False, True):
            hip._w2t_packed.clear()
            timeit(lambda: hip.ires_chain(t3, d2, w2t, d1, w1p, J, want_t2=t2), f"chain {dtype} plain weights     M={M} J={J} t2={t2}", fl)
            hip.pack_w2t_bf16([w2t])
            timeit(lambda: hip.ires_chain(t3, d2, w2t, d1, w1p, J, want_t2=t2), f"chain {dtype} arranged weights  M={M} J={J} t2={t2}", fl)
hip.set_compute_dtype("f32")
