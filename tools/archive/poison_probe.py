"""Uninitialised-read detector: run three training steps with the caching allocator's free blocks pre-filled with a
poison value, and compare with a run on differently poisoned memory.  usage: poison_probe.py [f32|bf16] [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip, configs, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda")


def poison(val):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    big = torch.full((6 << 30,), val, device=dev, dtype=torch.float32) if B > 16 else torch.full((2 << 30,), val, device=dev)
    small = [torch.full((1 << k,), val, device=dev) for k in range(4, 18) for _ in range(256)]
    torch.cuda.synchronize()
    del big, small


def run(val):
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.device = dev
    cfg.training.batch_size = B
    cfg.compute_dtype = dtype
    torch.manual_seed(0)
    score = utils.create_model(cfg, dev)
    flow = create_flow_model(cfg).to(dev)
    state, fstate, _ = utils.build_states(cfg, score, flow, dev)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); flow.train()
    rng = DeviceRng(dev, seed=21, poisson_seed=22)
    g = torch.Generator().manual_seed(23)
    outs = []
    for _ in range(3):
        x = (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).to(dev)
        poison(val)
        outs.append(torch.stack(step_fn(state, fstate, x, rng=rng)))
    torch.cuda.synchronize()
    return torch.stack(outs), state["optimizer"].flat.p.clone(), fstate["optimizer"].flat.p.clone()


a = run(0.0)
for val in (float("nan"), 1e30, -3.0):
    b = run(val)
    same = all(torch.equal(x, y) for x, y in zip(a, b))
    print(f"{dtype} batch {B} poison {val}: {'identical' if same else 'DIFFERENT'}  loss diff "
          f"{(a[0] - b[0]).abs().max().item():.3e}  finite {bool(torch.isfinite(b[0]).all())}", flush=True)
