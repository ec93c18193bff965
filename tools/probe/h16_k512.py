import sys, torch
sys.path.insert(0, "/root/repo")
from indm_amd import hip
hip.set_compute_dtype("bf16")
dev = torch.device("cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for S in (32, 16):
    x = torch.randn(128, S, S, 512, device=dev).bfloat16()
    w = torch.randn(1, 1, 512, 512, device=dev) * 0.05
    (fwd, dg), = hip.h16_pack_weights([w])
    t = timeit(lambda: hip.h16_conv2d(x, fwd, 1, 1, 1, 0))
    fl = 2.0 * 128 * S * S * 512 * 512
    print(f"h16 1x1 GEMM [{128*S*S}, 512] x [512, 512]: {t:.1f} us = {fl / t / 1e6:.0f} TF")
