"""Wall-clock of the phases of one training step (synchronised between phases; run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import configs, losses, sde_lib, utils
from indm_amd.flow import create_flow_model, flow_forward
from indm_amd.rng import DeviceRng
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch

dev = torch.device("cuda")
cfg = configs.get_config("vp/CIFAR10/indm_nll")
cfg.device = dev
torch.manual_seed(0)
score = utils.create_model(cfg, dev)
flow = create_flow_model(cfg).to(dev)
state, fstate, flat_g = utils.build_states(cfg, score, flow, dev)
sde = sde_lib.get_sde(cfg)
loss_fn = losses.get_sde_loss_fn(cfg, sde, train=True)
rng = DeviceRng(dev, seed=3)
score.train(); flow.train()
x = synthetic_batch(128, 32, 1, dev)


def sync():
    torch.cuda.synchronize(); return time.time()


for it in range(4):
    state["optimizer"].zero_grad(); fstate["optimizer"].zero_grad()
    t0 = sync()
    z, lf = flow_forward(cfg, flow, x, reverse=False, rng=rng)
    t1 = sync()
    ls = loss_fn(score, z, st=False, rng=rng)
    t2 = sync()
    (ls.mean()).backward(retain_graph=True, inputs=list(score.parameters()) + [z])
    t3 = sync()
    # flow backward from dL/dz (score path) + the flow loss
    gz = z.grad if z.grad is not None else torch.zeros_like(z)
    torch.autograd.backward([z, (-lf / 3072).mean()], [gz, None])
    t4 = sync()
    state["optimizer"].step(grad_clip=1.0, ema=state["ema"]); fstate["optimizer"].step(grad_clip=1.0, ema=fstate["ema"])
    t5 = sync()
    if it:
        print(f"flow fwd+logdet {1e3*(t1-t0):7.1f} ms | score fwd {1e3*(t2-t1):6.1f} | score bwd {1e3*(t3-t2):6.1f} | "
              f"flow bwd {1e3*(t4-t3):7.1f} | optim {1e3*(t5-t4):5.1f} | total {1e3*(t5-t0):7.1f}", flush=True)
