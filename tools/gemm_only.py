"""Run a few launches of one contraction shape (for rocprofv3 --pmc).  usage: gemm_only.py MODE [M N K]
MODE: nn | nt | ntg | nng (igemm; g = gated), vnn | vnt (vendor torch.matmul, reference point only), conv,
chain (the back-to-back VJP-chain kernel at M = 131072, J = 27), chain_bf16 (its bf16 instantiation on the fp32 LDS
image of W2T), chain_bf16w (bf16 with the pre-packed fragment-order weights)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip
mode = sys.argv[1] if len(sys.argv) > 1 else "nn"
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (128 * 1024, 512, 512)
a, w = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
wt = w.t().contiguous()
g_ = torch.randn(M, K, device="cuda")
x, cw = torch.randn(128, 32, 32, 128, device="cuda"), torch.randn(3, 3, 128, 128, device="cuda")
d1 = torch.randn(M, 512, device="cuda") if mode.startswith("chain") else None
if mode.startswith("chain_bf16"):
    hip.set_compute_dtype("bf16")
    if mode == "chain_bf16w":
        hip.pack_w2t_bf16([w])
w1p = torch.zeros(512, 32, device="cuda")
w1p[:, :27] = torch.randn(512, 27, device="cuda") / 22.6
for _ in range(5):
    if mode.startswith("chain"):
        hip.ires_chain(a, g_, w, d1, w1p, 27)
    elif mode == "nn":
        hip.gemm(a, w)
    elif mode == "nt":
        hip.gemm(a, wt, trans_b=True)
    elif mode == "ntg":
        hip.gemm(a, wt, trans_b=True, gate=g_)
    elif mode == "nng":
        hip.gemm(a, w, gate=g_)
    elif mode == "vnn":
        torch.matmul(a, w)
    elif mode == "vnt":
        torch.matmul(a, wt.t())
    elif mode == "conv":
        hip.conv2d(x, cw)
torch.cuda.synchronize()
