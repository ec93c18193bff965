"""Where the HOST time of one training step goes (cProfile; run on the GPU box).  At small per-rank batches the step is
launch-bound: the host's ~5 000 launches per step cost more than the GPU work.  usage: host_profile.py [batch] [config] [f32|bf16] [num_res_blocks]
The profiled steps run autograd single-threaded so that the backward's Python code shows up in the profile."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import configs, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng
from bench import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
cfg = configs.get_config(sys.argv[2] if len(sys.argv) > 2 else "vp/CIFAR10/indm_nll")
cfg.device = dev
cfg.training.batch_size = B
cfg.compute_dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
if len(sys.argv) > 4:
    cfg.model.num_res_blocks = int(sys.argv[4])
torch.manual_seed(42)
score = utils.create_model(cfg, dev)
flow = create_flow_model(cfg).to(dev)
state, fstate, flat_g = utils.build_states(cfg, score, flow, dev)
step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
score.train(); flow.train()
rng = DeviceRng(dev, seed=1234, poisson_seed=4321)
xs = [synthetic_batch(B, cfg.data.image_size, i, dev) for i in range(12)]
for i in range(4):
    step_fn(state, fstate, xs[i], rng=rng)
torch.cuda.synchronize()
t0 = time.time()
for i in range(4, 8):
    step_fn(state, fstate, xs[i], rng=rng)
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print(f"batch {B}: host returns after {(t1 - t0) / 4 * 1e3:.1f} ms per step; GPU done {(t2 - t0) / 4 * 1e3:.1f} ms per step")
import gc
gc.collect(); gc.disable()
t0 = time.time()
for i in range(4, 8):
    step_fn(state, fstate, xs[i], rng=rng)
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
gc.enable()
print(f"gc disabled: host returns after {(t1 - t0) / 4 * 1e3:.1f} ms per step; GPU done {(t2 - t0) / 4 * 1e3:.1f} ms per step; "
      f"gc counts {gc.get_count()} thresholds {gc.get_threshold()}")
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):
    step_fn(state, fstate, xs[7], rng=rng)
    pr.enable()
    for i in range(8, 12):
        step_fn(state, fstate, xs[i], rng=rng)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(60)
