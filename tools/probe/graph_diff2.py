"""two training steps, eager vs graphed score call, same init and noise: where do the runs part?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from indm_amd import configs, graphed, hip, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
out = {}
for mode in ("eager", "graph"):
    graphed.enabled(mode == "graph"); graphed.clear()
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.model.dropout = 0.0
    cfg.training.batch_size = B
    cfg.device = torch.device("cuda")
    cfg.compute_dtype = dtype
    torch.manual_seed(0)
    score = utils.create_model(cfg, cfg.device)
    fl = create_flow_model(cfg).to(cfg.device)
    with torch.no_grad():
        for p in score.parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    state, fstate, flat_g = utils.build_states(cfg, score, fl, cfg.device)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); fl.train()
    rec = []
    for it in range(2):
        g = torch.Generator().manual_seed(50 + it)
        x = (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).cuda()
        ls = step_fn(state, fstate, x, rng=DeviceRng(cfg.device, seed=9 + it, poisson_seed=3 + it))
        torch.cuda.synchronize()
        rec.append(([v.clone() for v in ls], state["optimizer"].flat.p.clone(), fstate["optimizer"].flat.p.clone(),
                    state["optimizer"].flat.g.clone(), fstate["optimizer"].flat.g.clone()))
    out[mode] = rec
    hip.set_compute_dtype("f32")
for it in range(2):
    e, g = out["eager"][it], out["graph"][it]
    print(f"step {it}: losses equal {[torch.equal(a, b) for a, b in zip(e[0], g[0])]}  score params equal {torch.equal(e[1], g[1])} "
          f"flow params equal {torch.equal(e[2], g[2])}  score grads equal {torch.equal(e[3], g[3])} ({float((e[3]-g[3]).abs().max()):.3e} of {float(e[3].abs().max()):.3e}) "
          f"flow grads equal {torch.equal(e[4], g[4])} ({float((e[4]-g[4]).abs().max()):.3e} of {float(e[4].abs().max()):.3e})")
    print("   loss eager", e[0][0].tolist(), "graph", g[0][0].tolist())
