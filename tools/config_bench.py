"""Training-step time of every BASELINE.json configuration at its PER-RANK shard on one MI355X (run on the GPU box).
#2 vp/CIFAR10/indm_nll b128 fp32 | #3 ve/CIFAR10/indm b512/8 bf16 | #4 vp/CELEBA/indm_fid b256/8 | #5 nrb=8 b256/8"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import configs, hip, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng
from bench import synthetic_batch

dev = torch.device("cuda")
CASES = [("#2 vp/CIFAR10/indm_nll", "vp/CIFAR10/indm_nll", 128, "f32", {}),
         ("#3 ve/CIFAR10/indm (512/8)", "ve/CIFAR10/indm", 64, "bf16", {}),
         ("#3 ve/CIFAR10/indm (512/8)", "ve/CIFAR10/indm", 64, "f32", {}),
         ("#4 vp/CELEBA/indm_fid (256/8)", "vp/CELEBA/indm_fid", 32, "f32", {}),
         ("#4 vp/CELEBA/indm_fid (256/8)", "vp/CELEBA/indm_fid", 32, "bf16", {}),
         ("#5 vp/CIFAR10/indm_nll nrb=8 (256/8)", "vp/CIFAR10/indm_nll", 32, "f32", {"num_res_blocks": 8}),
         ("#5 vp/CIFAR10/indm_nll nrb=8 (256/8)", "vp/CIFAR10/indm_nll", 32, "bf16", {"num_res_blocks": 8})]
only = sys.argv[1:] 
for name, cname, B, dtype, model_kw in CASES:
    if only and not any(o in name for o in only):
        continue
    if os.environ.get("INDM_CB_DTYPE", dtype) != dtype:
        continue
    cfg = configs.get_config(cname)
    cfg.device = dev
    cfg.training.batch_size = B
    cfg.model.update(**model_kw)
    cfg.compute_dtype = dtype
    torch.manual_seed(42)
    score = utils.create_model(cfg, dev)
    flow = create_flow_model(cfg).to(dev)
    state, fstate, _ = utils.build_states(cfg, score, flow, dev)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); flow.train()
    rng = DeviceRng(dev, seed=1234, poisson_seed=4321)
    S = cfg.data.image_size
    xs = [synthetic_batch(B, S, i, dev) for i in range(9)]
    for i in range(3):
        step_fn(state, fstate, xs[i], rng=rng)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(3, 9):
        step_fn(state, fstate, xs[i], rng=rng)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 6 * 1e3
    print(f"{name:40s} {dtype:5s} batch {B:4d} {S}x{S}: {ms:8.1f} ms/step  {B / ms * 1e3:8.1f} img/s per GPU   "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:5.1f} GiB", flush=True)
    del score, flow, state, fstate, step_fn, xs
    hip.set_compute_dtype("f32")
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
