"""Run every dense kernel family N times on the same inputs and compare the results bit for bit (run on the GPU box).
usage: determinism_probe.py [f32|bf16] [repeats] [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
hip.set_compute_dtype(mode)
dev = "cuda"
r = lambda *s: torch.randn(*s, device=dev)


def check(name, fn):
    ref = fn()
    ref = [t.clone() for t in (ref if isinstance(ref, (tuple, list)) else [ref]) if t is not None]
    bad = 0
    for _ in range(reps):
        out = fn()
        out = [t for t in (out if isinstance(out, (tuple, list)) else [out]) if t is not None]
        if not all(torch.equal(a, b) for a, b in zip(ref, out)):
            bad += 1
    print(f"{name:60s} {'OK' if bad == 0 else f'NONDETERMINISTIC in {bad}/{reps} runs'}  ({hip.lib().indm_igemm_last_variant().decode()})", flush=True)


M = B * 1024
a, g, w = r(M, 512), r(M, 512), r(512, 512)
wt = w.t().contiguous()
check("gemm NT plain", lambda: hip.gemm(a, wt, trans_b=True))
check("gemm NN gated", lambda: hip.gemm(a, w, gate=g))
check("gemm TN wgrad (big)", lambda: hip.gemm(a, g, trans_a=True, splitk=True))
check("gemm TN wgrad gated", lambda: hip.gemm(a, g, trans_a=True, splitk=True, gate=a))
d1 = r(M, 512)
w1p = torch.zeros(512, 32, device=dev); w1p[:, :27] = r(512, 27) / 22.6
check("chain (plain weights)", lambda: hip.ires_chain(a, g, w, d1, w1p, 27))
if mode == "bf16":
    hip.pack_w2t_bf16([w])
    check("chain (packed weights)", lambda: hip.ires_chain(a, g, w, d1, w1p, 27))
    check("chain (packed weights, t2)", lambda: hip.ires_chain(a, g, w, d1, w1p, 27, want_t2=True))
M1 = B * 256
a1, g1, d11 = r(M1, 512), r(M1, 512), r(M1, 512)
w1p4 = torch.zeros(512, 128, device=dev); w1p4[:, :108] = r(512, 108) / 22.6
check("chain scale 1", lambda: hip.ires_chain(a1, g1, w, d11, w1p4, 108))
for (H, C, Co) in [(32, 128, 128), (16, 256, 256), (8, 256, 256), (4, 256, 256), (32, 384, 128)]:
    x, wc, dy = r(B, H, H, C), r(3, 3, C, Co), r(B, H, H, Co)
    check(f"conv fwd {H} {C}->{Co}", lambda: hip.conv2d(x, wc))
    check(f"conv dgrad {H} {C}->{Co}", lambda: hip.conv2d_dgrad(dy, wc, (H, H)))
    check(f"conv wgrad {H} {C}->{Co}", lambda: hip.conv2d_wgrad(x, dy, 3, 3))
x3, w3, x5 = r(B, 32, 32, 3), r(3, 3, 3, 512), r(B, 32, 32, 512)
check("thin conv 3->512", lambda: hip.conv2d(x3, w3))
check("thin wgrad 512->3", lambda: hip.conv2d_wgrad(x5, x3, 3, 3))
check("thin wgrad 3->512", lambda: hip.conv2d_wgrad(x3, x5, 3, 3))
check("thin wgrad gated", lambda: hip.conv2d_wgrad(x3, x5, 3, 3, dy_gate=x5))
xg, gm, bt = r(B, 32, 32, 128), r(128), r(128)
y, mn, rs = hip.groupnorm_fwd(xg, gm, bt, 32, 1e-6, 1, 0.1, 7)
check("groupnorm fwd", lambda: hip.groupnorm_fwd(xg, gm, bt, 32, 1e-6, 1, 0.1, 7))
check("groupnorm bwd", lambda: hip.groupnorm_bwd(xg, xg, gm, bt, mn, rs, 32, 1, 0.1, 7))
