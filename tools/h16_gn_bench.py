"""GroupNorm (+SiLU) forward / backward on bf16 activations vs the fp32 kernels at the score network's shapes (batch 128):
us per launch and effective TB/s (forward: read + write; backward: 2 reads + 1 write).  INDM_H16_GNB_CH / _BS tune the backward."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip  # noqa: E402

DROP = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda")


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


for S, C in [(32, 128), (32, 256), (32, 384), (16, 256), (16, 512), (8, 256), (8, 512), (4, 256)]:
    G = min(C // 4, 32)
    x = torch.randn(B, S, S, C, device=dev)
    dy = torch.randn(B, S, S, C, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    x16, dy16 = x.bfloat16(), dy.bfloat16()
    y, m, r = hip.h16_groupnorm_fwd(x16, gamma, beta, G, 1e-6, 1, DROP, 77)
    tf = timeit(lambda: hip.h16_groupnorm_fwd(x16, gamma, beta, G, 1e-6, 1, DROP, 77))
    tb = timeit(lambda: hip.h16_groupnorm_bwd(dy16, x16, gamma, beta, m, r, G, 1, DROP, 77))
    y32, m32, r32 = hip.groupnorm_fwd(x, gamma, beta, G, 1e-6, 1, DROP, 77)
    of = timeit(lambda: hip.groupnorm_fwd(x, gamma, beta, G, 1e-6, 1, DROP, 77))
    ob = timeit(lambda: hip.groupnorm_bwd(dy, x, gamma, beta, m32, r32, G, 1, DROP, 77))
    n = x.numel()
    print(f"{S:3d}x{S:<3d} C={C:4d}: H16 fwd {tf:6.1f} us ({4 * n / tf / 1e6:5.2f} TB/s)  bwd {tb:6.1f} us ({6 * n / tb / 1e6:5.2f} TB/s)   |   "
          f"fp32 fwd {of:6.1f} us ({8 * n / of / 1e6:5.2f} TB/s)  bwd {ob:6.1f} us ({12 * n / ob / 1e6:5.2f} TB/s)", flush=True)
