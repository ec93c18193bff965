"""Which lines of indm_amd/ still call ATen ops that launch a device kernel inside one training step.
Without a GPU the kernel layer is replaced by tests/_emu.py exactly like the CPU host-logic tests, a
TorchDispatchMode records every ATen op with the Python stack it came from, and ops issued from inside the emulated
kernel bodies are dropped.  Launch counts do not depend on the batch size, so batch 2 of the full-size config gives
the per-step launch count of the bench workload."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import _emu
from indm_amd import hip, configs, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng

GPU = torch.cuda.is_available()   # on the GPU box: the real kernel layer, so host->device copies show up as well
if not GPU:
    for name in dir(_emu):
        if not name.startswith("_") and callable(getattr(_emu, name)) and hasattr(hip, name):
            setattr(hip, name, getattr(_emu, name))

VIEWS = {"view", "_unsafe_view", "reshape", "t", "transpose", "permute", "slice", "select", "expand", "as_strided",
         "detach", "alias", "unsqueeze", "squeeze", "split", "split_with_sizes", "unbind", "narrow", "empty",
         "empty_like", "empty_strided", "new_empty", "_local_scalar_dense", "is_same_size", "sym_size", "stride",
         "size", "chunk", "unfold", "view_as", "flatten", "lift_fresh", "is_pinned", "_to_copy_view", "resize_",
         "new_empty_strided", "set_", "real", "result_type", "can_cast", "is_nonzero", "item", "_has_compatible_shallow_copy_type"}


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in VIEWS:
            return out
        tens = [a for a in list(args) + [out] if torch.is_tensor(a)]
        if tens and max(t.numel() for t in tens) == 0:
            return out
        if GPU and tens and not any(t.is_cuda for t in tens):
            return out   # host arithmetic of the schedule: no launch
        site, emu = None, False
        for fr in reversed(traceback.extract_stack()[:-1]):
            fn = fr.filename
            if fn.endswith("_emu.py"):
                emu = True
                break
            if site is None and "/indm_amd/" in fn:
                site = f"{os.path.relpath(fn, ROOT)}:{fr.lineno} {fr.name}"
        if not emu:
            if site is None or "backward" in site or "step_fn" in site:   # the autograd engine's own ops: tell them apart by result shape
                name += str(tuple(out.shape)) if torch.is_tensor(out) else ""
            self.sites[(site or "autograd engine", name)] += 1
        return out


cfg = configs.get_config("vp/CIFAR10/indm_nll")
cfg.device = torch.device("cuda" if GPU else "cpu")
cfg.training.batch_size = 2
if not GPU:
    cfg.model.dropout = 0.0   # the Philox mask lives inside the GroupNorm kernel: no launches of its own
torch.manual_seed(0)
score = utils.create_model(cfg, cfg.device)
flow = create_flow_model(cfg).to(cfg.device)
state, fstate, _ = utils.build_states(cfg, score, flow, cfg.device)
sde = sde_lib.get_sde(cfg)
step_fn = losses.get_step_fn(cfg, sde, train=True)
score.train(); flow.train()
rng = DeviceRng(cfg.device, seed=1)
x = torch.rand(2, 3, 32, 32).to(cfg.device)
step_fn(state, fstate, x, rng=rng)
with Sites() as m:
    step_fn(state, fstate, x, rng=rng)
tot = sum(m.sites.values())
print(f"{tot} ATen ops outside the kernel layer in one step")
by_site = collections.defaultdict(collections.Counter)
for (site, name), n in m.sites.items():
    by_site[site][name] += n
for site, ops in sorted(by_site.items(), key=lambda kv: -sum(kv[1].values()))[:80]:
    print(f"{sum(ops.values()):5d}  {site:58s} {dict(ops.most_common(int(os.environ.get('TOP', 5))))}")
