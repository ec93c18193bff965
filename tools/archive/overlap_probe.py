"""Do an MFMA-bound GEMM and an HBM-bound elementwise pass overlap when issued on two streams? (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip
dev = "cuda"
M = 131072
r = lambda *s: torch.randn(*s, device=dev)
a, w = r(M, 512), r(512, 512)
da, du, d, aa, t = r(M, 512), r(M, 512), r(M, 512), r(M, 512), r(M, 512)
a2, b2 = r(M, 512), r(M, 512)
side = torch.cuda.Stream()
hip._ws(1 << 24, torch.device(dev))

def gemm(): return hip.gemm(a, w)
def wgrad(): return hip.gemm(a2, b2, trans_a=True, splitk=True)
def act(): return hip.ires_bwd_act(da, du, d, aa, t, True, 1024)

def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def both(main_fn, side_fn):
    def f():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            side_fn()
        main_fn()
        torch.cuda.current_stream().wait_stream(side)
    return f

print(f"gemm alone {timed(gemm):7.1f} us | wgrad alone {timed(wgrad):7.1f} us | act alone {timed(act):7.1f} us")
print(f"gemm || act   {timed(both(gemm, act)):7.1f} us   (serial {timed(lambda: (gemm(), act())):7.1f})")
print(f"act || wgrad  {timed(both(act, wgrad)):7.1f} us   (serial {timed(lambda: (act(), wgrad())):7.1f})")
print(f"gemm || wgrad {timed(both(gemm, wgrad)):7.1f} us   (serial {timed(lambda: (gemm(), wgrad())):7.1f})")
hi = torch.cuda.Stream(priority=-1)
side = hi
print(f"gemm || act (act on a high-priority stream) {timed(both(gemm, act)):7.1f} us")
