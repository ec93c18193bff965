"""eager vs graphed score call on the full-size network: which parameter gradients differ after ONE backward?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from indm_amd import configs, graphed, hip, utils, optim
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = configs.get_config("vp/CIFAR10/indm_nll")
cfg.model.dropout = 0.0
cfg.device = torch.device("cuda")
torch.manual_seed(0)
net = utils.create_model(cfg, cfg.device).train()
with torch.no_grad():
    for p in net.parameters():
        if p.dim() > 1 and p.abs().max() < 1e-6:
            p.copy_(torch.randn_like(p) * 0.02)
flat = optim.FlatParams(net)
hip.set_compute_dtype(dtype)
x = torch.randn(B, 32, 32, 3, device="cuda", requires_grad=True)
t = torch.rand(B, device="cuda") * 900 + 1
gout = torch.randn(B, 32, 32, 3, device="cuda")
res = {}
for mode in ("eager", "graph", "graph2"):
    flat.zero_grad()
    x.grad = None
    if mode == "eager":
        out = net(x, t, nhwc=True)
    else:
        out = graphed.score_call(net, flat, x, t, None)
    out.backward(gout)
    flat.gather_grads()
    torch.cuda.synchronize()
    res[mode] = (out.detach().clone(), x.grad.clone(), flat.g.clone())
for mode in ("graph", "graph2"):
    o, gx, g = res[mode]
    eo, egx, eg = res["eager"]
    print(mode, "out equal", torch.equal(o, eo), "gx equal", torch.equal(gx, egx), "flat grads equal", torch.equal(g, eg))
    if not torch.equal(g, eg):
        off = 0
        for n, p in flat.named:
            k = p.numel()
            a, b = g[off:off + k], eg[off:off + k]
            if not torch.equal(a, b):
                print("   differs:", n, tuple(p.shape), float((a - b).abs().max()), float(b.abs().max()))
            off += (k + 3) // 4 * 4
hip.set_compute_dtype("f32")
