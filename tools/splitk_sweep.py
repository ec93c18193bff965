"""Split-K slice count vs time for every dense launch of one training step (run on the GPU box).
One step of config #2 at the given per-rank batch is run with recording wrappers around hip.gemm / conv2d / conv2d_dgrad /
conv2d_wgrad; every distinct (operation, shapes) whose kernel variant splits K is then replayed on random data with the
slice count forced through INDM_SPLITK (the library's tuning hook) next to the rule's own choice.

usage: INDM_SPLITK_TUNE=1 python3 tools/splitk_sweep.py [f32|bf16] [batch] [min share of step in %]"""
import collections
import os
import sys

os.environ["INDM_SPLITK_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from indm_amd import configs, hip, losses, sde_lib, utils  # noqa: E402
from indm_amd.flow import create_flow_model  # noqa: E402
from indm_amd.rng import DeviceRng  # noqa: E402
from bench import synthetic_batch  # noqa: E402

args = sys.argv[1:]
dtype = args.pop(0) if args and args[0] in ("f32", "bf16") else "f32"
B = int(args.pop(0)) if args else 16
dev = torch.device("cuda")
cfg = configs.get_config("vp/CIFAR10/indm_nll")
cfg.device = dev
cfg.training.batch_size = B
cfg.compute_dtype = dtype
torch.manual_seed(42)
score = utils.create_model(cfg, dev)
flow = create_flow_model(cfg).to(dev)
state, fstate, _ = utils.build_states(cfg, score, flow, dev)
step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
score.train(); flow.train()
rng = DeviceRng(dev, seed=1234, poisson_seed=4321)
step_fn(state, fstate, synthetic_batch(B, 32, 0, dev), rng=rng)   # packs weights, sizes workspaces

seen = collections.OrderedDict()   # key -> [count, replay closure factory]


def sig(v):
    if torch.is_tensor(v):
        return ("T",) + tuple(v.shape)
    if isinstance(v, (tuple, list)):
        return tuple(sig(u) for u in v)
    return v


def like(v):
    if torch.is_tensor(v):
        return torch.randn(v.shape, device=dev, dtype=v.dtype) if v.is_floating_point() else v.clone()
    return v


def recorder(name, fn):
    def wrapped(*a, **k):
        key = (name, sig(a), tuple(sorted((kk, sig(vv)) for kk, vv in k.items())))
        if key not in seen:
            a2, k2 = [like(v) for v in a], {kk: like(vv) for kk, vv in k.items()}
            seen[key] = [0, (lambda a2=a2, k2=k2: fn(*a2, **k2))]
        seen[key][0] += 1
        return fn(*a, **k)
    return wrapped


orig = {n: getattr(hip, n) for n in ("gemm", "conv2d", "conv2d_dgrad", "conv2d_wgrad")}
for n, f in orig.items():
    setattr(hip, n, recorder(n, f))
step_fn(state, fstate, synthetic_batch(B, 32, 1, dev), rng=rng)
for n, f in orig.items():
    setattr(hip, n, f)
print(f"{dtype} batch {B}: {len(seen)} distinct dense calls from Python (the block drivers' own launches are not in this list)", flush=True)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def short(key):
    name, a, k = key
    shp = " ".join("x".join(str(d) for d in s[1:]) for s in a if isinstance(s, tuple) and s and s[0] == "T")
    kw = " ".join(f"{kk}" for kk, vv in k if isinstance(vv, tuple) and vv and vv[0] == "T")
    extra = " ".join(f"{kk}={vv}" for kk, vv in k if not isinstance(vv, tuple) and kk in ("stride", "trans_a", "trans_b"))
    return f"{name} {shp} {extra} {('+' + kw) if kw else ''}"


S = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48)
tot_def = tot_best = 0.0
rows = []
for key, (cnt, fn) in seen.items():
    os.environ["INDM_SPLITK"] = "0"
    t_def = timeit(fn)
    variant = hip.lib().indm_igemm_last_variant().decode()
    res = {}
    if "splitk" in variant or "big" in variant or variant.startswith("128x128") or variant.startswith("32x128") or variant.startswith("glds"):
        for s in S:
            os.environ["INDM_SPLITK"] = str(s)
            res[s] = (timeit(fn, 10), hip.lib().indm_igemm_last_variant().decode())
    os.environ["INDM_SPLITK"] = "0"
    best = min([(t_def, 0)] + [(t, s) for s, (t, _) in res.items()])
    rows.append((cnt * t_def, cnt, t_def, variant, best, res, key))
    tot_def += cnt * t_def
    tot_best += cnt * best[0]
rows.sort(key=lambda r: -r[0])
for tot, cnt, t_def, variant, best, res, key in rows:
    line = f"{tot / 1e3:7.2f} ms = {cnt:3d} x {t_def:7.1f} us [{variant:28s}] best S={best[1]:2d} {best[0]:7.1f} us | {short(key)}"
    print(line)
    if res and best[0] < 0.93 * t_def:
        print("          " + "  ".join(f"S{s}:{t:.0f}{'' if 'splitk' in v else '*'}" for s, (t, v) in res.items()))
print(f"sum over the step: rule {tot_def / 1e3:.2f} ms, best-per-shape {tot_best / 1e3:.2f} ms")
hip.set_compute_dtype("f32")
