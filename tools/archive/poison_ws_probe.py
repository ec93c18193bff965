"""Stale-scratch detector: three training steps with the shared scratch buffer NaN-filled before every op that asks
for it, compared with a plain run.  usage: poison_ws_probe.py [f32|bf16] [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip, configs, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda")


def run(poison):
    hip.POISON_WS["on"] = poison
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
    rec = []
    for _ in range(3):
        x = (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).to(dev)
        out = torch.stack(step_fn(state, fstate, x, rng=rng))
        torch.cuda.synchronize()
        rec.append((out, state["optimizer"].flat.g.clone(), fstate["optimizer"].flat.g.clone()))
    names = []
    for st in (state, fstate):
        o, lst = 0, []
        for n, p in st["optimizer"].flat.named:
            lst.append((n, o, p.numel()))
            o += (p.numel() + 3) // 4 * 4
        names.append(lst)
    hip.POISON_WS["on"] = False
    return rec, names


try:
    a, names = run(False)
    b, _ = run(True)
    for s in range(3):
        for k, what in enumerate(("loss", "score grad", "flow grad")):
            x, y = a[s][k], b[s][k]
            if torch.equal(x, y):
                continue
            msg = f"step {s} {what}: finite {bool(torch.isfinite(y).all())} max|d| {(x - y).abs().nan_to_num(9e9).max().item():.3e}"
            if k:
                d = ~((x == y) | (x.isnan() & y.isnan()))
                bad = [n for n, o, c in names[k - 1] if d[o:o + c].any()]
                nan = [n for n, o, c in names[k - 1] if y[o:o + c].isnan().any()]
                msg += f"; {len(bad)} params differ, {len(nan)} have NaN: {nan[:6]} ... {nan[-6:]}"
            print(msg, flush=True)
    print(f"{dtype} batch {B}: done", flush=True)
finally:
    hip.set_compute_dtype("f32")
