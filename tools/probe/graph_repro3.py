import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, "tests")
import torch
import parity_cases as pc
import golden_utils as gu
from indm_amd import graphed, hip, losses, sde_lib, utils, likelihood
from indm_amd import flow as iflow
from indm_amd.rng import ReplayRng, DeviceRng

mode = sys.argv[1]
device = "cuda"
d, cfg, score, fl = pc.build("mini_celeba", device)
sde = sde_lib.get_sde(cfg)
x = torch.from_numpy(d["x"]).to(device)
if "score" in mode:
    score.eval()
    out = score(x, (torch.from_numpy(d["score/t"]) * 999).to(device))
if "flow" in mode:
    for m in ("train", "eval"):
        fl.load_reference_state_dict(gu.state_dict(d, "flow"))
        fl.train() if m == "train" else fl.eval()
        rng = ReplayRng(gu.events(d, f"flow_{m}"), device)
        with torch.no_grad():
            z, ld = iflow.flow_forward(cfg, fl, x.clone(), reverse=False, rng=rng)
if "rev" in mode:
    fl.load_reference_state_dict(gu.state_dict(d, "flow"))
    fl.eval()
    rng = ReplayRng(gu.events(d, "flow_reverse"), device)
    with torch.no_grad():
        xr, _ = iflow.flow_forward(cfg, fl, torch.from_numpy(d["flow_eval/z"]).to(device), log_det=None, reverse=True, rng=rng)
if "nelbo" in mode:
    fn = likelihood.get_elbo_fn(cfg, sde, inverse_scaler=utils.get_data_inverse_scaler(cfg))
    rng = ReplayRng(gu.events(d, "nelbo"), device)
    b1, b2 = fn(score, fl, x, rng=rng)
state, flow_state, _ = utils.build_states(cfg, score, fl, device)
step_fn = losses.get_step_fn(cfg, sde, train=True)
score.train(); fl.train()
rng = ReplayRng(gu.events(d, "step0"), device)
outs = step_fn(state, flow_state, torch.from_numpy(d["step0/x"]).to(device), rng=rng)
torch.cuda.synchronize()
print("ok", mode)
