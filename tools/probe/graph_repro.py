import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, "tests")
import torch
import golden_utils as gu
from indm_amd import graphed, hip, ncsnpp, optim

name, S, attn = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1"
cfg = gu.mini_config(name)
cfg.data.image_size = S
cfg.model.attention = attn
cfg.device = torch.device("cuda")
net = ncsnpp.NCSNpp(cfg).cuda().train()
flat = optim.FlatParams(net)
x = torch.randn(2, S, S, 3, device="cuda", requires_grad=True)
t = torch.rand(2, device="cuda") * 900 + 1
out = graphed.score_call(net, flat, x, t, None)
out.square().sum().backward()
torch.cuda.synchronize()
print("ok", name, S, attn, float(out.abs().sum()))
