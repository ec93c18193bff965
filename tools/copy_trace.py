"""Where do the device-to-device copies (and other short torch-side kernels) of one STEADY-STATE training step come from?  Runs
the bench workload (batch 128 unless INDM_TRACE_BATCH), profiles ONE step with torch.profiler and prints (i) the Python-level
call sites of copy_ / clone / contiguous / to / zero_ / fill_ on GPU tensors, (ii) the profiler's totals for copies and
memsets, (iii) count and GPU time of every aten op whose kernels are all shorter than 10 us.  Round 5's finding: 13 Memcpy
DtoD and ~ 300 short aten kernels (1.3 ms) per step -- rocprofv3's 490 copyBuffer "per step" were start-up copies of a
4-step profile.  (run on the GPU box)"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from indm_amd import configs, hip, losses, parallel, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng

dev = "cuda:0"
batch = int(os.environ.get("INDM_TRACE_BATCH", "128"))
cfg = configs.get_config("vp/CIFAR10/indm_nll")
cfg.training.batch_size = batch
cfg.device = dev
cfg.compute_dtype = os.environ.get("INDM_TRACE_DTYPE", "f32")
torch.manual_seed(cfg.seed)
score = utils.create_model(cfg, dev)
flow = create_flow_model(cfg).to(dev)
state, flow_state, flat_g = utils.build_states(cfg, score, flow, dev)
sde = sde_lib.get_sde(cfg)
step_fn = losses.get_step_fn(cfg, sde, train=True, world_size=1, allreduce=parallel.make_allreduce(flat_g, 1))
score.train(); flow.train()
rng = DeviceRng(dev, seed=1234, poisson_seed=4321)
bs = [bench.synthetic_batch(batch, cfg.data.image_size, 1234 + 1000 * i, dev) for i in range(4)]
for i in range(3):
    step_fn(state, flow_state, bs[i], rng=rng)
torch.cuda.synchronize()
# Python-level call sites of the copying tensor methods (the autograd engine's own clones are C++ and do not show here)
import traceback
sites = collections.Counter()
def _wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        if self.is_cuda:
            fr = [x for x in traceback.extract_stack()[:-1] if "indm_amd" in x.filename or "bench.py" in x.filename][-2:]
            sites[(name, " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in reversed(fr)))] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
for n in ("copy_", "clone", "contiguous", "to", "zero_", "fill_"):
    _wrap(n)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step_fn(state, flow_state, bs[3], rng=rng)
    torch.cuda.synchronize()
print("python-level call sites of copy_/clone/contiguous/to/zero_/fill_ on GPU tensors in the step:")
for k, v in sites.most_common(40):
    print(f"  {v:5d}  {k[0]:12s} {k[1]}")
print("profiler totals:")
for a in prof.key_averages():
    if any(t in a.key for t in ("copy", "Memcpy", "clone", "contiguous", "Memset", "zero_", "fill_")):
        print(f"  {a.count:6d} calls  cpu {a.cpu_time_total / 1e3:8.3f} ms  gpu {a.device_time_total / 1e3:8.3f} ms  {a.key}")
print("short aten kernels (< 10 us each) and copies of the step, per op:")
tot = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if all(k.duration < 10 for k in e.kernels) and e.name.startswith("aten::"):
        tot[e.name][0] += len(e.kernels); tot[e.name][1] += sum(k.duration for k in e.kernels)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {v[0]:5d} launches {v[1] / 1e3:8.3f} ms  {k}")
