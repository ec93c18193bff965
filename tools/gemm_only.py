"""Run a few launches of one igemm shape (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip
mode = sys.argv[1] if len(sys.argv) > 1 else "nn"
M = 128 * 1024
a, w = torch.randn(M, 512, device="cuda"), torch.randn(512, 512, device="cuda")
x, wt = torch.randn(128, 32, 32, 128, device="cuda"), torch.randn(3, 3, 128, 128, device="cuda")
for _ in range(5):
    if mode == "nn":
        hip.gemm(a, w)
    elif mode == "nt":
        hip.gemm(a, w, trans_b=True)
    elif mode == "conv":
        hip.conv2d(x, wt)
torch.cuda.synchronize()
