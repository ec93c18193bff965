"""Repeat the flow's training forward + backward on the same inputs / seeds and compare z, log-det and the parameter
gradients bit for bit.  usage: flow_determinism.py [f32|bf16] [repeats] [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip, configs
from indm_amd.flow import create_flow_model
from indm_amd.losses import flow_forward
from indm_amd.rng import DeviceRng

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda")
cfg = configs.get_config("vp/CIFAR10/indm_nll")
cfg.device = dev
cfg.compute_dtype = dtype
hip.set_compute_dtype(dtype)
torch.manual_seed(0)
flow = create_flow_model(cfg).to(dev)
flow.train()
x = (torch.rand(B, 3, 32, 32, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(dev)
params = [p for p in flow.parameters() if p.requires_grad]


def once():
    rng = DeviceRng(dev, seed=21, poisson_seed=22)
    for p in params:
        p.grad = None
    z, ld = flow_forward(cfg, flow, x, reverse=False, rng=rng)
    (z.square().mean() - ld.mean()).backward()
    return [z.detach().clone(), ld.detach().clone()] + [p.grad.clone() for p in params if p.grad is not None]


ref = once()
names = ["z", "logdet"] + [n for n, p in flow.named_parameters() if p.requires_grad and p.grad is not None]
bad = 0
for r in range(reps):
    out = once()
    diff = [n for n, a, b in zip(names, ref, out) if not torch.equal(a, b)]
    if diff:
        bad += 1
        print(f"rep {r}: {len(diff)} tensors differ; first: {diff[:6]}", flush=True)
print(f"flow fwd+bwd {dtype} batch {B}: {bad}/{reps} repeats differ", flush=True)
