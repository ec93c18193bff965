"""Background load for race hunting: loops GEMM / conv launches until killed."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip
t0, lim = time.time(), float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
a, w = torch.randn(8192, 512, device="cuda"), torch.randn(512, 512, device="cuda")
x, wc = torch.randn(8, 32, 32, 128, device="cuda"), torch.randn(3, 3, 128, 128, device="cuda")
while time.time() - t0 < lim:
    for _ in range(50):
        hip.gemm(a, w); hip.conv2d(x, wc)
    torch.cuda.synchronize()
    time.sleep(0.002)
