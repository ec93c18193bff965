"""How far the host runs ahead of the GPU inside one training step (run on the GPU box).
At a few points of the step the host time and a HIP event are recorded; lag = GPU arrival - host arrival.
lag ~ 0 means the GPU drained its queue and is waiting for the host there (host-bound stretch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import configs, losses, sde_lib, utils, optim, flow as iflow
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng
from bench import synthetic_batch

dev = torch.device("cuda")
cfg = configs.get_config("vp/CIFAR10/indm_nll")
cfg.device = dev
cfg.training.batch_size = 128
torch.manual_seed(42)
score = utils.create_model(cfg, dev)
flow = create_flow_model(cfg).to(dev)
state, fstate, flat_g = utils.build_states(cfg, score, flow, dev)
sde = sde_lib.get_sde(cfg)
step_fn = losses.get_step_fn(cfg, sde, train=True)
score.train(); flow.train()
rng = DeviceRng(dev, seed=1234)
marks = []

def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, time.time(), e))

def wrap(obj, attr, name, before=True):
    f = getattr(obj, attr)
    def g(*a, **k):
        if before: mark(name)
        r = f(*a, **k)
        if not before: mark(name)
        return r
    setattr(obj, attr, g)

wrap(optim.FlatParams, "gather_grads", "backward returned (before gather_grads)")
wrap(optim.FusedAdamW, "step", "before optimiser step")
wrap(iflow.ResidualFlow, "prepare", "flow.prepare done", before=False)
wrap(iflow.ResidualFlow, "fwdpass", "flow fwdpass enqueued", before=False)

for it in range(5):
    x = synthetic_batch(128, 32, 100 + it, dev)
    torch.cuda.synchronize()
    marks.clear()
    mark("step start")
    t0 = time.time()
    out = step_fn(state, fstate, x, rng=rng)
    t1 = time.time()
    mark("step returned (losses on the CPU)")
    torch.cuda.synchronize()
    if it >= 2:
        h0, e0 = marks[0][1], marks[0][2]
        print(f"--- step {it}: {1e3 * (t1 - t0):.1f} ms")
        seen = set()
        for name, h, e in marks:
            if name in seen: continue
            seen.add(name)
            print(f"  {name:45s} host {1e3 * (h - h0):8.2f} ms   gpu {e0.elapsed_time(e):8.2f} ms   lag {e0.elapsed_time(e) - 1e3 * (h - h0):8.2f} ms")
