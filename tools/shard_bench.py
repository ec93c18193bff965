"""Training-step time of config #2 at the per-rank shards of a strong-scaling run (16 / 32 / 64 / 128 images) on one
MI355X, with the host-return time next to the GPU-done time (run on the GPU box).  usage: shard_bench.py [f32|bf16] [batches...]
INDM_BLOCK_DRIVERS=0 runs the op-level composition for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import configs, hip, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng
from bench import synthetic_batch

dev = torch.device("cuda")
args = sys.argv[1:]
dtype = args.pop(0) if args and args[0] in ("f32", "bf16") else "f32"
batches = [int(v) for v in args] or [16, 32, 64, 128]
cfg = configs.get_config("vp/CIFAR10/indm_nll")
cfg.device = dev
cfg.compute_dtype = dtype
torch.manual_seed(42)
score = utils.create_model(cfg, dev)
flow = create_flow_model(cfg).to(dev)
state, fstate, _ = utils.build_states(cfg, score, flow, dev)
step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
score.train(); flow.train()
for B in batches:
    rng = DeviceRng(dev, seed=1234, poisson_seed=4321)
    xs = [synthetic_batch(B, 32, i, dev) for i in range(11)]
    for i in range(3):
        step_fn(state, fstate, xs[i], rng=rng)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(3, 11):
        step_fn(state, fstate, xs[i], rng=rng)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print(f"batch {B:4d}: {(t2 - t0) / 8 * 1e3:7.1f} ms/step ({B * 8 / (t2 - t0):7.1f} img/s)   host returns after "
          f"{(t1 - t0) / 8 * 1e3:7.1f} ms/step   drivers={'on' if hip.block_drivers() else 'off'} {dtype} storage={hip.storage_dtype()}", flush=True)
hip.set_compute_dtype("f32")
