"""GroupNorm(+SiLU+dropout) forward / backward at the score net's batch-128 shapes (run on the GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (N, H, C) in [(128, 32, 128), (128, 32, 256), (128, 32, 384), (128, 16, 256), (128, 16, 512), (128, 8, 256), (128, 8, 512),
                  (128, 4, 256), (128, 4, 512), (16, 32, 128), (16, 16, 256)]:
    x = torch.randn(N, H, H, C, device="cuda"); g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    dy = torch.randn_like(x)
    G = min(C // 4, 32)
    y, m, r = hip.groupnorm_fwd(x, g, b, G, 1e-6, 1, 0.1, 123)
    tf = t(lambda: hip.groupnorm_fwd(x, g, b, G, 1e-6, 1, 0.1, 123))
    tb = t(lambda: hip.groupnorm_bwd(dy, x, g, b, m, r, G, 1, 0.1, 123))
    nb = x.numel() * 4
    print(f"GN {N}x{H}x{H}x{C}: fwd {tf:7.1f} us ({2 * nb / tf / 1e6:5.2f} TB/s at 8 B/elt)  bwd {tb:7.1f} us "
          f"({3 * nb / tb / 1e6:5.2f} TB/s at 12 B/elt)")
