"""One thin GEMM shape, a few launches (PMC target).  usage: thin_one.py {k27|k108|n27g|n108g|wg27|wg108} """
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip
which = sys.argv[1] if len(sys.argv) > 1 else "k27"
r = lambda *s: torch.randn(*s, device="cuda")
M, C = (131072, 3) if which.endswith("27") or which.endswith("27g") else (32768, 12)
K9 = 9 * C
wide, gate, thin = r(M, 512), r(M, 512), r(M, K9)
if which.startswith("k"):
    w = r(K9, 512); fn = lambda: hip.gemm(thin, w)
elif which.startswith("n"):
    w = r(K9, 512); fn = lambda: hip.gemm(wide, w, trans_b=True, gate=gate)
else:
    fn = lambda: hip.gemm(wide, thin, trans_a=True, splitk=True, gate=gate)
for _ in range(5):
    fn()
torch.cuda.synchronize()
