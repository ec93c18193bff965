"""40 optimisation steps on a FIXED synthetic batch: the NLL objective must go down and stay finite (run on the GPU box).
usage: train_sanity.py [f32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import configs, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng
from bench import synthetic_batch

dev = torch.device("cuda")
cfg = configs.get_config("vp/CIFAR10/indm_nll")
cfg.device = dev
cfg.compute_dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
torch.manual_seed(0)
score = utils.create_model(cfg, dev)
flow = create_flow_model(cfg).to(dev)
state, fstate, _ = utils.build_states(cfg, score, flow, dev)
step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
score.train(); flow.train()
x = synthetic_batch(64, 32, 1, dev)
rng = DeviceRng(dev, seed=3)
t0 = time.time()
for it in range(40):
    l, ls, lf, lp = step_fn(state, fstate, x, rng=rng)
    assert torch.isfinite(l).all()
    if it % 5 == 0 or it == 39:
        print(f"step {it:3d}  loss {l.mean():9.4f}  score {ls.mean():8.4f}  flow {lf.mean():8.4f}  logp {lp.mean():8.4f}", flush=True)
torch.cuda.synchronize()
print(f"40 steps in {time.time() - t0:.1f} s, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
