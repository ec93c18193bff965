import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, "tests")
import torch
import golden_utils as gu
from indm_amd import graphed, hip, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng

name = sys.argv[1]
cfg = gu.mini_config(name)
cfg.device = torch.device("cuda")
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    tgt = cfg
    ks = k.split(".")
    for q in ks[:-1]:
        tgt = getattr(tgt, q)
    setattr(tgt, ks[-1], eval(v))
score = utils.create_model(cfg, cfg.device)
fl = create_flow_model(cfg).to(cfg.device)
state, fstate, _ = utils.build_states(cfg, score, fl, cfg.device)
step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
score.train(); fl.train()
S = cfg.data.image_size
for it in range(3):
    x = torch.rand(2, 3, S, S, device="cuda") * 2 - 1
    out = step_fn(state, fstate, x, rng=DeviceRng(cfg.device, seed=it, poisson_seed=5))
torch.cuda.synchronize()
print("ok", name, sys.argv[2:], [float(v.mean()) for v in out])
