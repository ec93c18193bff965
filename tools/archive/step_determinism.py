"""Race hunting: run the 3-step training twice from the same seeds, keep losses / flat gradients / parameters of EVERY
step, and report the first tensor (and parameter names) that differs.  usage: step_determinism.py [f32|bf16] [tries] [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indm_amd import hip, configs, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda")


def run():
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
        rec.append(dict(loss=out, sg=state["optimizer"].flat.g.clone(), fg=fstate["optimizer"].flat.g.clone(),
                        sp=state["optimizer"].flat.p.clone(), fp=fstate["optimizer"].flat.p.clone()))
    names = {}
    for key, st in (("sg", state), ("fg", fstate)):
        o, lst = 0, []
        for n, p in st["optimizer"].flat.named:
            lst.append((n, o, p.numel()))
            o += (p.numel() + 3) // 4 * 4
        names[key] = lst
    return rec, names


try:
    a, names = run()
    nbad = 0
    for t in range(tries):
        b, _ = run()
        found = False
        for s in range(3):
            for key in ("loss", "sg", "fg", "sp", "fp"):
                x, y = a[s][key], b[s][key]
                if not torch.equal(x, y):
                    found = True
                    msg = f"try {t}: first difference at step {s} in {key}: max |d| {(x - y).abs().max().item():.3e}"
                    if key == "loss":
                        msg += f" rows {sorted(set((x != y).nonzero()[:, 0].tolist()))}"
                    if key in names:
                        d = (x != y)
                        hit = [(n, int(d[o:o + k].sum())) for n, o, k in names[key] if d[o:o + k].any()]
                        msg += f"; {len(hit)} parameters differ: {hit[:12]}"
                    print(msg, flush=True)
                    break
            if found:
                break
        nbad += found
    print(f"{dtype} batch {B}: {nbad}/{tries} runs differ from the first", flush=True)
finally:
    hip.set_compute_dtype("f32")
