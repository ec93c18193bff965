"""Flake hunting, third tool: WHICH elements differ.  Repeats one full-size forward + backward of the training loss (batch 4,
fixed weights / inputs / noise seeds, no optimiser step: every repetition is the same computation) and compares EVERY module's
forward output and arriving gradient elementwise with the first repetition's, on the GPU, one host read per repetition.  On
a mismatch it reports the module, how many elements differ, where (index ranges per dimension, distinct images / channel
blocks / 128-byte lines), by how much, and saves both tensors.  Meant to run next to other GPU processes (the whole suite
under `pytest -n 3`), the only setting in which the bit-reproducibility test has ever failed.

usage: flake_pattern.py [seconds] [tag] [f32|bf16]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import configs, hip, losses, sde_lib, utils  # noqa: E402
from indm_amd.flow import create_flow_model  # noqa: E402
from indm_amd.rng import DeviceRng  # noqa: E402

limit = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
tag = sys.argv[2] if len(sys.argv) > 2 else "p0"
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
dev = torch.device("cuda")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(OUT, exist_ok=True)

cfg = configs.get_config("vp/CIFAR10/indm_nll")
cfg.device = dev
cfg.training.batch_size = 4
cfg.compute_dtype = dtype
torch.manual_seed(0)
score = utils.create_model(cfg, dev)
flow = create_flow_model(cfg).to(dev)
state, fstate, _ = utils.build_states(cfg, score, flow, dev)
sde = sde_lib.get_sde(cfg)
hip.set_compute_dtype(dtype)
loss_fn = losses.get_sde_loss_fn(cfg, sde, train=True)
score.train(); flow.train()
g = torch.Generator().manual_seed(23)
x = (torch.rand(4, 3, 32, 32, generator=g) * 2 - 1).to(dev)

ref, names, cur = [], [], []
recording = [True]
mism = []   # (slot, device scalar count)


def see(name, t):
    t = t.detach()
    slot = len(cur)
    cur.append(t)
    if recording[0]:
        ref.append(t.clone())
        names.append(name)
    else:
        r = ref[slot]
        if r.shape != t.shape:
            mism.append((slot, torch.tensor(-1, device=dev)))
        else:
            mism.append((slot, (r.view(torch.int32) != t.contiguous().view(torch.int32)).sum()))


def fwd_hook(name):
    def hook(_m, _inp, out):
        outs = [o for o in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(o) and o.is_floating_point()]
        for j, o in enumerate(outs):
            see(f"fwd {name}[{j}]", o.contiguous())
            if o.requires_grad:
                o.register_hook(lambda gr, nm=f"bwd {name}[{j}]": see(nm, gr.contiguous()))
    return hook


for pre, mod in (("score", score), ("flow", flow)):
    for n, m in mod.named_modules():
        if n:
            m.register_forward_hook(fwd_hook(f"{pre}.{n}"))


def one():
    from indm_amd.flow import flow_forward
    del cur[:], mism[:]
    for p in list(score.parameters()) + list(flow.parameters()):
        p.grad = None
    rng = DeviceRng(dev, seed=21, poisson_seed=22)
    z, lf = flow_forward(cfg, flow, x, reverse=False, rng=rng)
    ls = loss_fn(score, z, rng=rng)
    see("loss_score", ls)
    see("loss_flow", lf)
    (ls - lf / 3072.0).mean().backward()
    grads = [p.grad for p in list(score.parameters()) + list(flow.parameters()) if p.grad is not None]
    see("all grads", torch.cat([gr.reshape(-1) for gr in grads]))


def describe(name, a, b):
    d = (a.view(torch.int32) != b.view(torch.int32))
    idx = d.nonzero()
    msg = [f"{name}: shape {tuple(a.shape)}, {idx.shape[0]} of {a.numel()} elements differ, max |d| {(a - b).abs().max().item():.3e}, "
           f"max |ref| {a.abs().max().item():.3e}"]
    for dim in range(idx.shape[1]):
        col = idx[:, dim]
        msg.append(f"  dim {dim}: min {int(col.min())} max {int(col.max())} distinct {int(col.unique().numel())}")
    flat = d.reshape(-1).nonzero().flatten()
    lines = (flat // 32).unique()
    msg.append(f"  distinct 128-byte lines (fp32): {int(lines.numel())}; first flat indices {flat[:12].tolist()}; "
               f"ulps of the first: {[(int(a.reshape(-1)[i].view(torch.int32)) - int(b.reshape(-1)[i].view(torch.int32))) for i in flat[:12].tolist()]}")
    return "\n".join(msg)


one()
recording[0] = False
torch.cuda.synchronize()
print(f"[{tag}] reference recorded: {len(ref)} tensors, {sum(r.numel() for r in ref) * 4 / 1e6:.0f} MB", flush=True)
t0, it, nbad = time.time(), 0, 0
while time.time() - t0 < limit:
    one()
    it += 1
    assert len(cur) == len(ref), (len(cur), len(ref))
    counts = torch.stack([c for _, c in mism]).tolist()
    bad = [(slot, c) for (slot, _), c in zip(mism, counts) if c != 0]
    if bad:
        nbad += 1
        slot, c = bad[0]
        print(f"[{tag}] repetition {it}: {len(bad)} of {len(ref)} tensors differ; FIRST (execution order): slot {slot}", flush=True)
        print(describe(names[slot], ref[slot], cur[slot]), flush=True)
        if slot > 0:
            print(f"  previous tensor (identical): {names[slot - 1]} shape {tuple(ref[slot - 1].shape)}", flush=True)
        torch.save({"name": names[slot], "ref": ref[slot].cpu(), "got": cur[slot].cpu(),
                    "prev_name": names[slot - 1] if slot else None, "prev": ref[slot - 1].cpu() if slot else None},
                   os.path.join(OUT, f"pattern_{tag}_{nbad}.pt"))
        if nbad >= 6:
            break
print(f"[{tag}] {it} repetitions in {time.time() - t0:.0f} s, {nbad} with a difference", flush=True)
hip.set_compute_dtype("f32")
