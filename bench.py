"""Headline benchmark: INDM (VP, NLL) training images/s on synthetic CIFAR-10-shaped batches.

    python bench.py --gpus N --steps K --warmup W
    N > 1 without a torchrun environment: this process starts the N ranks itself (fresh child processes through
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, before anything here
    touches the GPU), relays rank 0's JSON line and exits with the job's code.  Under torchrun (RANK / WORLD_SIZE set,
    the driver's N > 1 launch) it is one rank; WORLD_SIZE must equal --gpus.

A "step" is one call of the reference-API train_step_fn (flow forward + log-det, score loss, backward, one
flat-gradient all-reduce when N > 1, fused clip+AdamW+EMA for both networks) on configs/vp/CIFAR10/indm_nll
at batch 128 PER GPU (weak scaling; `--global-batch G` shards a fixed global batch instead: strong scaling, the
partitioning of SURVEY 8e / BASELINE configs #3-#5), fp32, dropout 0.1, random-init weights of the full architecture
(61.8 M + 13.2 M parameters).  Rank 0 prints ONE JSON line; see DESIGN.md "Measurement".
Extra keys at N = 1: `cpu_baseline` (+ `cpu_baseline_8_threads`), `nelbo`, `sampling` (BASELINE config #5: 1000-step
predictor-only PC sampling throughput of the deep num_res_blocks = 8 NCSN++ at batch 256, hipGraph-captured step).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak of the same guide
# what "bf16" means in this build: which tensors live in HBM as bf16, which only enter the MFMAs as bf16
BF16_TENSORS = {
    "stored_bf16": "the flow's [pixels, 512] tensors a1, d1, a2, d2, t3, t2 and the temporaries of its backward, at the scales the "
                   "F16 kernels cover (C <= 12 channels, >= 1024 pixels: both CIFAR scales, CelebA's first); round 6: the score "
                   "network's trunk -- NHWC activations and their gradients of every ResnetBlockBigGANpp, the skip concatenations, "
                   "the resampling -- and a per-call bf16 pack of its convolution weights (csrc/unet_h16*.hip)",
    "stored_fp32_rounded_to_bf16_at_the_mfma": "the score network's attention blocks (6 of 50 blocks: an fp32-storage island), its "
                                               "first 3 -> nf and last nf -> 3 convolutions, the time embedding; the flow's 3- / "
                                               "12-channel side, CelebA's 48-channel scale",
    "fp32_always": "accumulators, GroupNorm / BatchNorm statistics, reductions, log-dets, losses, optimiser state, gradients",
    "step_roofline_peak": "2.5 PFLOP/s dense bf16 for the WHOLE step, although part of it is fp32-stored: read `frac` with this list",
}


def step_gflop_per_image(n_bar):
    """Algorithmic work of one vp/CIFAR10/indm_nll training step per image (BASELINE.md section 2 / SURVEY App. C, 2*MAC
    convention): the score net's forward + backward (65.08 GFLOP) plus the flow's log-det forward + backward with n
    Poisson series terms per block, 12.73 (n + 8) GFLOP -- evaluated at the REALISED mean series length of the timed
    steps (E[n] = 2 gives the 192.4 GFLOP of BASELINE.md)."""
    return 12.73 * (n_bar + 8.0) + 65.08


class _CountingPoisson:
    """wraps a noise source's poisson(): the realised series lengths of the timed region (BASELINE.md section 3 asks for
    the roofline at the realised n-bar, not at its expectation)"""

    def __init__(self, rng):
        self.rng, self.draws, self._orig = rng, [], rng.poisson
        rng.poisson = self

    def __call__(self, lam):
        n = self._orig(lam)
        self.draws.append(n)
        return n

    def reset(self):
        self.draws = []

    def mean(self):
        return sum(self.draws) / len(self.draws) if self.draws else None


def step_roofline(n_bar, images_per_s, dtype, gflop_per_image=None):
    """achieved = (algorithmic GFLOP per image at the realised n-bar) x images/s, vs the MFMA
    peak of the arithmetic type"""
    g = step_gflop_per_image(n_bar) if gflop_per_image is None else gflop_per_image
    peak = PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    ach = g * images_per_s / 1e3
    return {"gflop_per_image": g, "achieved_tflops": ach, "peak_tflops": peak, "frac": ach / peak, "n_bar": n_bar}


def device_identity(dev):
    """what this rank's GPU is: enough to tell N ranks on N GPUs from N ranks on one"""
    pr = torch.cuda.get_device_properties(dev)
    out = {"index": dev.index, "name": pr.name}
    for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        if hasattr(pr, k):
            out[k] = str(getattr(pr, k))
    return out


def synthetic_batch(B, size, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, size, size, generator=g)
    x = (255.0 * x + torch.rand(x.shape, generator=g)) / 256.0   # run_lib.py:86 uniform dequantisation
    return (x * 2.0 - 1.0).to(device)                             # data.centered scaler (datasets.py:58-60)


def _usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # honour a cgroup CPU quota (otherwise torch oversubscribes the cores it may not use)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, 32))


class _RecordingRng:
    """Oracle noise source that records every draw so the GPU path can be replayed on the same noise."""

    def __init__(self, base):
        self.base, self.events = base, []

    def _rec(self, kind, v):
        import numpy as np
        self.events.append((kind, np.asarray(v.numpy() if torch.is_tensor(v) else v).copy()))
        return v

    def randn(self, *shape):
        return self._rec("randn", self.base.randn(*shape))

    def randn_like(self, x):
        return self._rec("randn", self.base.randn_like(x))

    def rand(self, *shape):
        return self._rec("rand", self.base.rand(*shape))

    def rademacher_like(self, x):
        return self._rec("rademacher", self.base.rademacher_like(x))

    def poisson(self, lam):
        return int(self._rec("poisson", self.base.poisson(lam)))


def cpu_baseline_child(path, batch, threads=0):
    """Child process: one full-size oracle training step on the host cores (bounded sample), plus the oracle's NELBO
    bits/dim of two synthetic images with the noise it drew (the parent replays that noise on the GPU).
    threads > 0: only the training step, on that many threads (the 8-thread figure comparable with SURVEY App. C)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import indm_oracle as O
    from indm_amd import configs
    torch.set_num_threads(threads if threads > 0 else _usable_cores())
    blob = torch.load(path)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    torch.manual_seed(4242)
    if threads <= 0:
        rec = _RecordingRng(O.Rng())
        xe = synthetic_batch(2, cfg.data.image_size, 4242, "cpu")
        t0 = time.time()
        nb, nbr = O.nelbo_fn(blob["score"], blob["flow"], cfg, O.get_sde(cfg), xe, rec)
        torch.save({"events": rec.events, "bpd": nb, "bpd_residual": nbr, "seconds": time.time() - t0}, path + ".nelbo")
    st = O.OracleState(blob["score"], cfg.optim.lr, cfg.model.ema_rate, cfg)
    fst = O.OracleState(blob["flow"], cfg.flow.lr, cfg.flow.ema_rate, cfg)
    x = synthetic_batch(batch, cfg.data.image_size, 999, "cpu")
    t0 = time.time()
    O.flow_step_fn_nll(cfg, O.get_sde(cfg), st, fst, x, O.Rng())
    dt = time.time() - t0
    print(json.dumps({"dt": dt, "threads": torch.get_num_threads()}))


def cpu_baseline(cfg, score, flow, batch=16, timeout=420):
    """The CPU oracle (a port of the reference's step, pinned to reference goldens) timed on this box's host
    cores on a bounded sample: ONE full-size training step at a small batch, in a child process with a hard
    timeout so the default bench run stays within minutes."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "weights.pt")
        torch.save({"score": {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()},
                    "flow": {k: v.cpu() for k, v in flow.reference_state_dict(prefix="").items()}}, path)
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")

        def child(threads, limit):
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", path,
                                    "--batch", str(batch), "--cpu-threads", str(threads)], capture_output=True,
                                   text=True, timeout=limit, env=env)
                res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            except Exception as e:  # noqa: BLE001
                return {"value": None, "unit": "images/s", "cores": threads or _usable_cores(), "kind": "port",
                        "sample": f"oracle step at batch {batch} did not finish within {limit}s ({type(e).__name__})"}
            return {"value": batch / res["dt"], "unit": "images/s", "cores": res["threads"], "kind": "port",
                    "sample": f"1 full-size vp/CIFAR10/indm_nll train step at batch {batch} on the CPU oracle "
                              f"(oracle/indm_oracle.py, torch CPU fp32, {res['threads']} threads): {res['dt']:.1f} s"}

        full = child(0, timeout)
        nelbo = torch.load(path + ".nelbo", weights_only=False) if os.path.exists(path + ".nelbo") else None
        eight = child(8, 240) if (full["value"] is not None and _usable_cores() > 8) else None
    return full, nelbo, eight


def nelbo_leg(cfg, sde, score, flow, dev, oracle_nelbo):
    """The metric's second half: NELBO bits/dim of the same weights on the HIP path -- replayed on the
    noise the CPU oracle drew for two synthetic images (|difference| must stay inside the +-0.01 bits/dim north-star
    band) -- and nelbo_fn's own throughput on a batch of 32."""
    from indm_amd import likelihood, utils
    from indm_amd.rng import DeviceRng, ReplayRng
    fn = likelihood.get_elbo_fn(cfg, sde, inverse_scaler=utils.get_data_inverse_scaler(cfg))
    score.eval(); flow.eval()
    out = {}
    if oracle_nelbo is not None:
        xe = synthetic_batch(2, cfg.data.image_size, 4242, dev)
        got, got_r = fn(score, flow, xe, rng=ReplayRng(oracle_nelbo["events"], dev))
        out.update(bpd=[float(v) for v in got], bpd_cpu_oracle=[float(v) for v in oracle_nelbo["bpd"]],
                   max_abs_diff_bpd=float((got.cpu() - oracle_nelbo["bpd"]).abs().max()),
                   max_abs_diff_residual_bpd=float((got_r.cpu() - oracle_nelbo["bpd_residual"]).abs().max()),
                   cpu_oracle_images_per_s=2.0 / oracle_nelbo["seconds"])
        out["within_north_star_band"] = bool(out["max_abs_diff_bpd"] <= 0.01)
        if not out["within_north_star_band"]:
            print("[bench] WARNING: NELBO differs from the CPU oracle by more than 0.01 bits/dim", file=sys.stderr)
    xb = synthetic_batch(32, cfg.data.image_size, 777, dev)
    rng = DeviceRng(dev, seed=5)
    # algorithmic FLOPs of one call: the sum over its dense launches (2 M N K each), counted on the op-level composition
    # (same kernels as the block drivers, which enqueue theirs from C++ where Python cannot see them)
    from indm_amd import hip
    drivers_on = hip.block_drivers()
    hip.block_drivers(False)
    hip.profile_start()
    fn(score, flow, xb, rng=rng)
    prof = hip.profile_stop()
    hip.block_drivers(drivers_on)
    flops_call = sum(q["flops"] for q in prof.values())
    fn(score, flow, xb, rng=rng)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        fn(score, flow, xb, rng=rng)
    torch.cuda.synchronize()
    out["images_per_s"] = 3 * 32 / (time.time() - t0)
    out["gflop_per_image"] = flops_call / 32 / 1e9
    out["achieved_tflops"] = out["gflop_per_image"] * out["images_per_s"] / 1e3
    bf16 = hip.compute_dtype() == "bf16"
    out["peak_tflops"] = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS   # (--dtype bf16: the leg runs in that mode too)
    out["frac"] = out["achieved_tflops"] / out["peak_tflops"]
    out["note"] = "random-init weights after the timed steps; eval-mode flow log-det (n+20 power-series terms)"
    score.train(); flow.train()
    return out


def sampling_leg(cfg0, flow, dev, steps=60, batch=256, nrb=8, dtype="f32"):
    """BASELINE config #5: predictor-only PC sampling (`sampling.method=pc`, Euler-Maruyama, no corrector) of the deep
    NCSN++ (`model.num_res_blocks = 8`, random init) at batch 256.  Times `steps` consecutive sampler steps (the first
    two -- eager warm-up and graph capture -- are excluded), reports the per-step wall time next to the GPU time of
    the same steps measured with HIP events on the launch stream (they agree when the captured hipGraph keeps the GPU
    back-to-back busy), and the images/s of a full 1000-step run incl. the flow's fixed-point inverse."""
    import copy
    from indm_amd import hip, sampling, sde_lib, utils
    from indm_amd.rng import DeviceRng
    cfg = copy.deepcopy(cfg0)
    cfg.model.num_res_blocks = nrb
    cfg.sampling.method, cfg.sampling.predictor, cfg.sampling.corrector = "pc", "euler_maruyama", "none"
    cfg.sampling.batch_size = batch
    torch.manual_seed(cfg.seed + 5)
    score = utils.create_model(cfg, dev).eval()
    flow.eval()
    sde = sde_lib.get_sde(cfg)
    st = sampling._Stepper(cfg, sde, score, DeviceRng(dev, seed=77))
    res = {}
    with torch.no_grad():
        x = st.rng.randn_image(batch, 3, cfg.data.image_size, cfg.data.image_size)
        ts = torch.linspace(sde.T, cfg.sampling.truncation_time, 1000)
        table = sampling.predictor_table(st, "euler_maruyama", ts[:steps + 2], batch).to(dev)
        # algorithmic FLOPs of one sampler step: the sum over the dense launches of an eager fp32 step (2 M N K each; the
        # bf16-storage kernels are not bracketed per launch, the architecture and hence the count are the same)
        hip.set_compute_dtype("f32")
        hip.profile_start()
        sampling._GraphedPredictor(st, x, table, use_graph=False).step(0)
        flops_step = sum(q["flops"] for q in hip.profile_stop().values())
        hip.set_compute_dtype(dtype)
        for mode in ("graph", "eager"):
            gp = sampling._GraphedPredictor(st, x, table, use_graph=(mode == "graph"))
            gp.step(0); gp.step(1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.time()
            e0.record()
            for i in range(2, steps + 2):
                gp.step(i)
            e1.record()
            torch.cuda.synchronize()
            wall = (time.time() - t0) / steps
            res[mode] = {"wall_ms_per_step": 1e3 * wall, "gpu_ms_per_step": e0.elapsed_time(e1) / steps}
        t0 = time.time()
        from indm_amd.flow import flow_forward
        flow_forward(cfg, flow, torch.randn(batch, 3, 32, 32, device=dev), log_det=None, reverse=True,
                     rng=DeviceRng(dev, seed=78))
        torch.cuda.synchronize()
        t_inv = time.time() - t0
    g = res["graph"]
    flow.train()
    del score
    peak_dtype = hip.compute_dtype()
    hip.set_compute_dtype("f32")
    return {"workload": f"pc_sampler step (EM predictor, no corrector), NCSN++ num_res_blocks={nrb}, batch {batch}, "
                        + ("fp32" if dtype == "f32" else "bf16 MFMA inputs / fp32 accumulate, bf16-stored activations (H16)")
                        + f", {steps} timed steps of a 1000-step schedule",
            "hipgraph": g, "eager_launches": res["eager"],
            "sampler_steps_per_s": 1e3 / g["wall_ms_per_step"],
            "flow_inverse_s": t_inv,
            "images_per_s_1000_step_sampling": batch / (1000 * g["wall_ms_per_step"] * 1e-3 + t_inv),
            "nfe_per_image": 1000,
            "gflop_per_image_per_step": flops_step / batch / 1e9,
            "achieved_tflops": flops_step / (g["gpu_ms_per_step"] * 1e-3) / 1e12,
            "peak_tflops": PEAK_BF16_MFMA_TFLOPS if peak_dtype == "bf16" else PEAK_F32_MFMA_TFLOPS,
            "frac": flops_step / (g["gpu_ms_per_step"] * 1e-3) / 1e12
                    / (PEAK_BF16_MFMA_TFLOPS if peak_dtype == "bf16" else PEAK_F32_MFMA_TFLOPS)}


def pmc_traffic(kernel_key):
    """HBM bytes per launch of `kernel_key` from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    in separate runs of this same command, gfx950 FETCH_SIZE correction applied; profiles/*traffic*.json).  PMC
    counters cannot be read from inside the timed process, so this is the newest committed measurement of the same
    kernel instantiation, or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        if d.get("bench_key") == kernel_key:
            return d["hbm_bytes_per_launch_corrected"], os.path.relpath(path, ROOT)
    return None, None


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def per_rank_batch(batch, global_batch, world):
    """-> (images per rank, strong?): `--batch B` keeps B images per GPU (weak scaling); `--global-batch G` shards a fixed
    global batch evenly over the ranks (strong scaling: the partitioning of BASELINE configs #3 - #5) and overrides it."""
    if global_batch > 0:
        if global_batch % world:
            raise SystemExit(f"--global-batch {global_batch} does not divide over {world} ranks")
        return global_batch // world, True
    return batch, False


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: start N rank processes (one per GPU, RCCL over xGMI) as CHILDREN of
    this process -- which has not touched the GPU and never will -- and hand their output through.  Returns the job's
    exit code.  (Never exec-replace: a process image swap after GPU initialisation takes the node down on this pool.)"""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, _usable_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] starting {n} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if proc.returncode == 0 and len(lines) != 1:
        print(f"[bench] expected ONE JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode


def run_training(args, cfg_name, batch, dtype, steps, warmup, rank, world, dev, kernel_roofline=True, exposed=False):
    """Build both networks (random init, the full architecture of `cfg_name`) and time `steps` train_step_fn calls after
    `warmup` untimed ones, bracketed by barrier + synchronize.  Returns the measurements of this rank (elapsed is the MAX
    over ranks) plus the live objects the extra legs of the headline run need."""
    from indm_amd import configs, graphed, hip, losses, parallel, sde_lib, utils
    from indm_amd.flow import create_flow_model
    from indm_amd.rng import DeviceRng

    graphed.clear()     # the captured graphs of an earlier leg's networks (and their memory pools) are not needed any more
    cfg = configs.get_config(cfg_name)
    cfg.training.batch_size = batch * world
    cfg.device = dev
    cfg.compute_dtype = dtype
    torch.manual_seed(cfg.seed)                     # default_cifar10_configs.py:126
    score = utils.create_model(cfg, dev)
    flow = create_flow_model(cfg).to(dev)
    state, flow_state, flat_g = utils.build_states(cfg, score, flow, dev)
    dist_on = torch.distributed.is_initialized()
    if dist_on:
        parallel.broadcast_parameters(state["optimizer"].flat.p)
        parallel.broadcast_parameters(flow_state["optimizer"].flat.p)
    sde = sde_lib.get_sde(cfg)
    reducer = parallel.make_allreduce(flat_g, world)
    step_fn = losses.get_step_fn(cfg, sde, train=True, world_size=world, allreduce=reducer)
    score.train(); flow.train()
    batches = [synthetic_batch(batch, cfg.data.image_size, 1234 + rank + 1000 * i, dev) for i in range(steps + warmup)]
    rng = DeviceRng(dev, seed=1234 + rank, poisson_seed=4321)   # same series lengths on every rank: balanced steps
    counter = _CountingPoisson(rng)

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    dominant, warm_prof = None, {}
    drivers_on = hip.block_drivers()
    for i in range(warmup):
        profiled = kernel_roofline and i == warmup - 1
        if profiled:
            # which dense kernel class owns the most GPU time?  Timed per launch with HIP events on the op-level
            # composition (the block drivers enqueue the same kernels from C++, where Python cannot bracket them)
            hip.block_drivers(False)
            hip.profile_start()
        step_fn(state, flow_state, batches[i], rng=rng)
        if profiled:
            prof = hip.profile_stop()
            hip.block_drivers(drivers_on)
            warm_prof = prof or {}
            if prof:
                dominant = max(prof, key=lambda k: prof[k]["ms"])
                if rank == 0 and os.environ.get("INDM_BENCH_PROFILE"):
                    for k in sorted(prof, key=lambda k: -prof[k]["ms"]):
                        q = prof[k]
                        print(f"[bench] {k:72s} launches {q['launches']:5d}  {q['ms']:8.2f} ms  "
                              f"{q['flops'] / 1e9:10.1f} GFLOP  {q['flops'] / q['ms'] / 1e9:7.2f} TF/s",
                              file=sys.stderr, flush=True)
    barrier()
    # the dominant class inside the timed region: HIP events on the launch stream around ITS launches only -- recorded by
    # the C-side launch timer when the block drivers own those launches (chain-kernel classes), else by hip.py
    chain_timer, variant = None, None
    if dominant:
        variant = dominant[len("igemm<"):-1].split(",", 2)[-1]
        if drivers_on and variant.startswith("ires_chain"):
            # capacity from the launches the profiled warm-up step counted for this class (x4: Poisson series lengths vary)
            per_step = max(64, 4 * warm_prof[dominant]["launches"])
            chain_timer = hip.ChainTimer(per_step * steps)
            chain_timer.reset(variant)
            hip.set_chain_timer(chain_timer)
        hip.profile_start(only=dominant)
    counter.reset()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.time()
    last = None
    marks[0].record()
    for i in range(steps):
        last = step_fn(state, flow_state, batches[warmup + i], rng=rng)
        marks[i + 1].record()    # GPU-side step boundaries (no host sync): the median step time of SURVEY 8(d)
    barrier()
    elapsed = time.time() - t0
    n_bar = counter.mean()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    prof = hip.profile_stop() if dominant else {}
    if chain_timer is not None:
        hip.set_chain_timer(None)
        r = chain_timer.read()
        p0 = prof.get(dominant, dict(launches=0, flops=0.0, bytes=0.0, ms=0.0))
        # algorithmic bytes per launch of a chain class (hip.ires_chain's figure): t3 + d2 + d1 + W2T + P (+ t2 when stored)
        M = batch * (1024 if "x32" in variant else 256)
        J = 27 if "x32" in variant else 108
        if hip.storage_dtype() == "bf16":
            # F16 storage: the chain kernel reads the PRE-GATED operand (t3 .* d2, written by the thin convolution) and d1
            nb = 2.0 * 2 * M * 512 + 2.0 * 512 * 512 + 4.0 * M * J + (2.0 * M * 512 if "t2out" in variant else 0)
        else:
            nb = 4.0 * 3 * M * 512 + 4.0 * 512 * 512 + 4.0 * M * J + (4.0 * M * 512 if "t2out" in variant else 0)
        prof[dominant] = dict(launches=p0["launches"] + r["launches"], flops=p0["flops"] + r["flops"],
                              bytes=p0["bytes"] + nb * r["launches"], ms=p0["ms"] + r["ms"])
    exposed_ms, exposed_info = None, None
    if exposed and dist_on and world > 1:
        # the same steps WITHOUT the collective (replicas drift apart from here on: last thing this function does): what the
        # all-reduce costs on the critical path = step time with it - step time without it.  The flow's cost follows the
        # Poisson series lengths (n + 8 passes per block), so the replay draws them from a FRESH stream with the timed run's
        # seeds and walks the same warm-up + timed batches: every replayed step unrolls exactly the series its timed twin did
        # (a continued stream on other batches made the difference noise of a few ms at 264 ms per step).
        nocoll = losses.get_step_fn(cfg, sde, train=True, world_size=world, allreduce=None)
        rng2 = DeviceRng(dev, seed=1234 + rank, poisson_seed=4321)
        counter2 = _CountingPoisson(rng2)
        for i in range(warmup):
            nocoll(state, flow_state, batches[i], rng=rng2)
        barrier()
        # the replay is the timed run's twin in its profiling state too: the same per-launch HIP-event brackets
        # around the dominant class -- hip.py's for op-level launches, the C-side timer of the block drivers for chain kernels
        timer2 = None
        if dominant:
            if chain_timer is not None:
                timer2 = hip.ChainTimer(max(64, 4 * warm_prof[dominant]["launches"]) * steps)
                timer2.reset(variant)
                hip.set_chain_timer(timer2)
            hip.profile_start(only=dominant)
        counter2.reset()
        m2 = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t1 = time.time()
        m2[0].record()
        for i in range(steps):
            nocoll(state, flow_state, batches[warmup + i], rng=rng2)
            m2[i + 1].record()
        barrier()
        if dominant:
            hip.profile_stop()
            if timer2 is not None:
                hip.set_chain_timer(None)
        same_series = counter2.mean() == n_bar   # the replay unrolled the series lengths of the timed run
        t_nc = torch.tensor([(time.time() - t1) / steps], device=dev)
        torch.distributed.all_reduce(t_nc, op=torch.distributed.ReduceOp.MAX)
        exposed_ms = 1e3 * float(t_nc)
        # per-step differences on this rank (GPU-side step boundaries): their spread says how far to trust the mean
        d = [marks[i].elapsed_time(marks[i + 1]) - m2[i].elapsed_time(m2[i + 1]) for i in range(steps)]
        mean_d = sum(d) / len(d)
        exposed_info = {"per_step_difference_ms_rank0": {"mean": mean_d, "min": min(d), "max": max(d),
                                                         "std": (sum((v - mean_d) ** 2 for v in d) / len(d)) ** 0.5},
                        "replayed_steps": steps, "same_poisson_stream": bool(same_series)}
    if dist_on:
        t = torch.tensor([elapsed], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    roof, measured_in = None, "timed region"
    if dominant and not (dominant in prof and prof[dominant]["ms"] > 0) and warm_prof.get(dominant, {}).get("ms", 0) > 0:
        # a short run whose timed steps did not launch the class (or lost its events): the same per-launch HIP-event
        # timing taken in the profiled warm-up step stands in, and the line says so
        prof[dominant], measured_in = warm_prof[dominant], "profiled warm-up step"
        if rank == 0:
            print(f"[bench] no timed-region launches recorded for {dominant}: roofline from the warm-up step",
                  file=sys.stderr, flush=True)
    if dominant and dominant in prof and prof[dominant]["ms"] > 0:
        q = prof[dominant]
        ach = q["flops"] / (q["ms"] * 1e-3) / 1e12
        traffic, src = pmc_traffic(dominant)
        peak = PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        roof = {"bound": "mfma", "kernel": dominant, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "traffic": traffic, "traffic_source": src, "measured_in": measured_in, "launches": q["launches"],
                "avg_launch_ms": q["ms"] / q["launches"], "algorithmic_gflop_per_launch": q["flops"] / q["launches"] / 1e9,
                "algorithmic_bytes_per_launch": q["bytes"] / q["launches"]}
    if roof is not None and chain_timer is not None and hip.storage_dtype() == "bf16":
        # the class was chosen on the op-level composition of the profiled warm-up step (bf16 MFMA inputs on fp32 tensors: the
        # only form Python can bracket per launch); the launches TIMED above are the block drivers' F16 (bf16-storage) chain
        # links, another kernel: say so
        roof["kernel_class_chosen_on"] = roof["kernel"]
        roof["kernel"] = roof["kernel"][:-1] + ",f16_storage>"
        roof["kernel_symbol"] = ("f16_chain_conv_kernel (csrc/flow_f16.hip, round 6: thin transposed convolution + d2 gate + chain link of one "
                                 "series iteration in ONE launch; its FLOPs include the convolution's), timed by the C-side launch timer of "
                                 "the block drivers" if getattr(hip, "CHAIN_CONV", {}).get("on") else
                                 "f16_rows_kernel<CHAIN, ...> (csrc/flow_f16.hip), timed by the C-side launch timer of the block drivers")
    imgs_per_s = batch * world * steps / elapsed
    if roof is not None and n_bar is not None:
        # BASELINE.md section 3's definition: the WHOLE step at the realised n-bar (the dominant kernel above is ~13 % of it)
        roof["step"] = step_roofline(n_bar, imgs_per_s / world, dtype)
    return dict(cfg=cfg, sde=sde, score=score, flow=flow, elapsed=elapsed, images_per_s=imgs_per_s,
                ms_per_step=1e3 * elapsed / steps, median_ms=median_ms, roofline=roof, n_bar=n_bar,
                last_loss_mean=float(last[0].mean()) if last is not None else None,
                allreduce_exposed_ms=max(0.0, 1e3 * elapsed / steps - exposed_ms) if exposed_ms is not None else None,
                allreduce_exposed_raw_ms=(1e3 * elapsed / steps - exposed_ms) if exposed_ms is not None else None,
                allreduce_exposed_detail=exposed_info,
                ms_per_step_without_collective=exposed_ms,
                collectives_per_step=list(reducer.calls) if reducer is not None else None)


def shard_legs(args, dev):
    """One rank's share of every multi-GPU BASELINE configuration's partitioning on 8 GPUs: config #2's global
    batch 128 strong-scaled (16 images per rank), #5 / #4-sized 32 and #3-sized 64 images per rank, on the headline's
    network, fp32 and bf16; no collective (x 8 minus the all-reduce is the node figure).  Short runs: 2 warm-up + 5 steps."""
    out = {}
    for b in (16, 32, 64):
        for dtype in ("f32", "bf16"):
            r = run_training(args, "vp/CIFAR10/indm_nll", b, dtype, steps=5, warmup=2, rank=0, world=1, dev=dev,
                             kernel_roofline=False)
            out[f"batch{b}_{dtype}"] = {"ms_per_step": r["ms_per_step"], "ms_per_step_median": r["median_ms"],
                                        "images_per_s": r["images_per_s"],
                                        "step_roofline_frac": step_roofline(r["n_bar"], r["images_per_s"], dtype)["frac"],
                                        "n_bar": r["n_bar"]}
            del r
            torch.cuda.empty_cache()
    out["workload"] = ("configs/vp/CIFAR10/indm_nll.py train step on one rank's shard of an 8-GPU job (no collective): "
                       "16 / 32 / 64 images, 2 warm-up + 5 timed steps each (the 16-image bf16 step is within 3 ms of "
                       "the host's launch time: its mean moves with host load, see the median)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: a fixed GLOBAL batch sharded evenly over the ranks (SURVEY 8e: 128 -> 16/rank, "
                         "512 -> 64, 256 -> 32 at 8 GPUs); overrides --batch")
    ap.add_argument("--no-sampling", action="store_true", help="skip the config-#5 sampling leg")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the bf16 and per-rank-shard training legs (N = 1)")
    ap.add_argument("--cpu-threads", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = the headline (exact fp32 MFMA); bf16 = MFMA inputs rounded to bf16, fp32 accumulate, the "
                         "flow's wide tensors stored as bf16 (the dtype of BASELINE configs #3-#5; NOT the headline)")
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child(args.cpu_baseline_child, args.batch, args.cpu_threads)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return sys.exit(launch_ranks(args.gpus, sys.argv[1:]))      # before any GPU call in this process

    from indm_amd import hip, parallel

    # test hook (tests/test_bench_multirank_gpu.py): all ranks share GPU 0 and exchange gradients over gloo, so that
    # the multi-process path of this file can be exercised on a single-GPU box (RCCL refuses two ranks on one device)
    one_gpu = os.environ.get("INDM_BENCH_ONE_GPU") == "1"
    rank, world, local = parallel.init_distributed(backend="gloo" if one_gpu else None)
    if one_gpu:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE = {world} ranks")
    args.batch, strong = per_rank_batch(args.batch, args.global_batch, world)
    # host-side work per step is a few [B]-vector ops: keep every rank's intra-op pool small so N ranks on one host do
    # not oversubscribe its cores
    torch.set_num_threads(max(1, min(8, _usable_cores() // max(world, 1))))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X; there is no CPU fallback for the product path")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    hip.lib()
    dist_on = torch.distributed.is_initialized()
    ident = device_identity(dev)
    idents = [ident]
    if dist_on:
        idents = [None] * world
        torch.distributed.all_gather_object(idents, ident)

    r = run_training(args, "vp/CIFAR10/indm_nll", args.batch, args.dtype, args.steps, args.warmup, rank, world, dev,
                     exposed=True)
    cfg, sde, score, flow, elapsed = r["cfg"], r["sde"], r["score"], r["flow"], r["elapsed"]

    if rank == 0:
        print(f"[bench] {args.steps} steps in {elapsed:.2f}s on {world} GPU(s)", file=sys.stderr, flush=True)
        out = {
            "metric": "training images/sec, CIFAR-10 INDM (VP, NLL)",
            "value": r["images_per_s"], "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "ms_per_step_median": r["median_ms"], "value_at_median_step": args.batch * world / (r["median_ms"] * 1e-3),
            "config": {"workload": "configs/vp/CIFAR10/indm_nll.py train step (flow 16-16 idim 512 + NCSN++ nf128), "
                                   f"batch {args.batch}/GPU, 32x32x3, "
                                   + ("fp32" if args.dtype == "f32" else "bf16 MFMA inputs / fp32 accumulate, bf16 storage of "
                                      "the flow's 512-wide tensors and of the score network's trunk (NOT the fp32 parity "
                                      "configuration)")
                                   + ", dropout 0.1, random-init weights",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "last_loss_mean": r["last_loss_mean"], "n_bar": r["n_bar"]},
            "roofline": r["roofline"],
        }
        # who ran: one entry per rank (device uuid / PCI bus id), so that N ranks on N GPUs can be told from N ranks on one
        keys = [d.get("uuid") or d.get("pci_bus_id") or str(d.get("index")) for d in idents]
        out["ranks_seen"] = len(idents)
        out["distinct_devices"] = len(set(keys))
        out["devices"] = idents
        out["backend"] = torch.distributed.get_backend() if dist_on else None
        if world > 1:
            out["allreduce_exposed_ms"] = r["allreduce_exposed_ms"]          # clamped at 0; the raw difference and its spread:
            out["allreduce_exposed_raw_ms"] = r["allreduce_exposed_raw_ms"]
            out["allreduce_exposed_detail"] = r["allreduce_exposed_detail"]
            out["ms_per_step_without_collective"] = r["ms_per_step_without_collective"]
            out["collective_elements_per_step"] = r["collectives_per_step"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], oracle_nelbo, out["cpu_baseline_8_threads"] = cpu_baseline(cfg, score, flow)
            out["nelbo"] = nelbo_leg(cfg, sde, score, flow, dev, oracle_nelbo)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_sampling:
            try:
                out["sampling"] = sampling_leg(cfg, flow, dev)
                # the per-rank shard of config #5 on 8 GPUs (256 / 8 images): here the captured graph matters
                out["sampling"]["per_rank_shard_of_8"] = sampling_leg(cfg, flow, dev, steps=60, batch=32)
                # config #5 is one of the bf16 configurations: the same sampler on bf16-stored activations
                out["sampling"]["bf16"] = sampling_leg(cfg, flow, dev, steps=60, batch=256, dtype="bf16")
                out["sampling"]["bf16"]["per_rank_shard_of_8"] = sampling_leg(cfg, flow, dev, steps=60, batch=32, dtype="bf16")
            except Exception as e:  # noqa: BLE001  (the headline line must survive a failure of an extra leg)
                out["sampling"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_extra_legs:
            del r, score, flow
            torch.cuda.empty_cache()
            try:
                if args.dtype == "f32":
                    # BASELINE configs #3-#5 compute in bf16: the same workload in that mode, as a sub-leg of the default run
                    # (round 6: as many timed steps as the headline, at least 16 -- the 8-step leg moved with host noise)
                    nb = max(args.steps, 16)
                    b = run_training(args, "vp/CIFAR10/indm_nll", args.batch, "bf16", steps=nb, warmup=3,
                                     rank=0, world=1, dev=dev)
                    out["bf16"] = {"value": b["images_per_s"], "unit": "images/s", "ms_per_step": b["ms_per_step"],
                                   "ms_per_step_median": b["median_ms"],
                                   "value_at_median_step": args.batch / (b["median_ms"] * 1e-3), "steps": nb, "warmup": 3,
                                   "roofline": b["roofline"], "last_loss_mean": b["last_loss_mean"], "n_bar": b["n_bar"],
                                   "dtype": "bf16 MFMA inputs, fp32 accumulate, bf16 storage of the flow's 512-wide tensors and "
                                            "of the score network's trunk",
                                   "bf16_tensors": BF16_TENSORS}
                    del b
                    torch.cuda.empty_cache()
                out["train_per_rank_shard_of_8"] = shard_legs(args, dev)
            except Exception as e:  # noqa: BLE001
                out["extra_legs_error"] = f"{type(e).__name__}: {e}"
            finally:
                hip.set_compute_dtype("f32")
        print(json.dumps(out))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
