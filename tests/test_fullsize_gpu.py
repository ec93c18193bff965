"""Full-size parity: ONE training step of the real configs/vp/CIFAR10/indm_nll architecture (61.8 M + 13.2 M
parameters, 32 iResBlocks of width 512, 55-module NCSN++) at batch 2 on the HIP path vs the CPU oracle with the
same random-init weights and the same noise.  Exercises every production tile shape / channel count that the
mini golden configs cannot (tall 256x128 tiles, 512-wide gates, skinny 3/12 <-> 512 edges, 384/512-ch GroupNorm)."""
import numpy as np
import pytest
import torch

import golden_utils as gu  # noqa: F401  (puts oracle/ on sys.path)
import indm_oracle as O
from indm_amd import configs, likelihood, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import ReplayRng

pytestmark = pytest.mark.gpu


class RecordingRng(O.Rng):
    def __init__(self):
        self.events = []

    def _rec(self, kind, v):
        self.events.append((kind, np.asarray(v.numpy() if torch.is_tensor(v) else v).copy()))
        return v

    def randn(self, *shape):
        return self._rec("randn", torch.randn(*shape))

    def randn_like(self, x):
        return self._rec("randn", torch.randn_like(x))

    def rand(self, *shape):
        return self._rec("rand", torch.rand(*shape))

    def rademacher_like(self, x):
        return self._rec("rademacher", super().rademacher_like(x))

    def poisson(self, lam):
        n = super().poisson(lam)
        self.events.append(("poisson", np.asarray(n)))
        return n


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name", ["vp/CIFAR10/indm_nll", "ve/CIFAR10/indm", "vp/CELEBA/indm_fid"])
def test_full_size_training_step_matches_oracle(name):
    """vp: flow_step_fn_nll; ve: the two-phase flow_step_fn_fid with the FIR / Fourier-embedding NCSN++ (upfirdn2d);
    vp/CELEBA (BASELINE config #4 at PRODUCTION shapes): 64x64 images, flow.squeeze -> residual flow on 12 @ 32^2 and
    48 @ 16^2 with the imagenet/64x64 wolf encoder, 5-level NCSN++ (ch_mult 1,1,2,2,2), two-phase fid step"""
    torch.manual_seed(7)
    np.random.seed(7)
    cfg = configs.get_config(name)
    cfg.model.dropout = 0.0
    cfg.training.batch_size = 2
    dev = torch.device("cuda")
    cfg.device = dev
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    with torch.no_grad():  # init_scale=0 leaves the output convs at 1e-10: make the untrained net non-degenerate
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    x = (torch.rand(2, 3, cfg.data.image_size, cfg.data.image_size) * 2 - 1)

    ost = O.OracleState(sd_s, cfg.optim.lr, cfg.model.ema_rate, cfg)
    ofst = O.OracleState(sd_f, cfg.flow.lr, cfg.flow.ema_rate, cfg)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ostep = O.flow_step_fn_nll if cfg.training.likelihood_weighting else O.flow_step_fn_fid
    state, fstate, _ = utils.build_states(cfg, score, fl, dev)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); fl.train()
    # TWO steps: the second one runs on weights the fused optimiser wrote through the flat buffer (anything that
    # cached a view or a re-laid-out copy of a parameter would show up here)
    for it in range(2):
        rec = RecordingRng()
        want = ostep(cfg, O.get_sde(cfg), ost, ofst, x, rec)
        rng = ReplayRng(rec.events, dev)
        got = step_fn(state, fstate, x.to(dev), rng=rng)
        assert rng.done()
        for name, a, b in zip(("loss", "score", "flow", "logp"), got, want):
            gu.assert_close(a, b, 1e-3 * (1 + 2 * it), 1e-5 * (1 + 2 * it), f"full-size step {it} {name}")
    ss, sf = score.reference_state_dict(prefix=""), fl.reference_state_dict(prefix="")
    for sd, o, lr in ((ss, ost, cfg.optim.lr), (sf, ofst, cfg.flow.lr)):
        for k in list(o.trainable)[:: max(1, len(o.trainable) // 12)]:
            diff = (sd[k].cpu().double() - o.P[k].detach().double()).abs()
            assert diff.max().item() <= 4.4 * lr, (k, diff.max().item())
            # AdamW's first update is ~lr*sign(g): an element whose gradient sits at rounding-noise level may move the
            # other way under a different reduction order, so a few elements (<= 0.5 %, at least 2) may differ by 2*lr
            bad = int((diff > 1e-6 + 1e-4 * o.P[k].detach().abs().double()).sum().item())
            assert bad <= max(2, 5e-3 * diff.numel()), (k, bad, diff.numel())


@pytest.mark.timeout(1500)
def test_full_size_loss_curve_matches_oracle():
    """BASELINE config #2's parity gate ("vs reference loss curve"): EIGHT consecutive optimisation steps of the
    full-size vp/CIFAR10/indm_nll networks on a fresh batch each, HIP path vs CPU oracle on identical noise.  The four
    per-image loss vectors agree to 1e-6 .. 1e-5 relative at every step (measured); the gate is 2e-4."""
    torch.manual_seed(7)
    np.random.seed(7)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.model.dropout = 0.0
    cfg.training.batch_size = 2
    dev = torch.device("cuda")
    cfg.device = dev
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    ost = O.OracleState(sd_s, cfg.optim.lr, cfg.model.ema_rate, cfg)
    ofst = O.OracleState(sd_f, cfg.flow.lr, cfg.flow.ema_rate, cfg)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    state, fstate, _ = utils.build_states(cfg, score, fl, dev)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); fl.train()
    worst = 0.0
    for it in range(8):
        x = torch.rand(2, 3, 32, 32) * 2 - 1
        rec = RecordingRng()
        want = O.flow_step_fn_nll(cfg, O.get_sde(cfg), ost, ofst, x, rec)
        rng = ReplayRng(rec.events, dev)
        got = step_fn(state, fstate, x.to(dev), rng=rng)
        assert rng.done()
        for nm, a, b in zip(("loss", "score", "flow", "logp"), got, want):
            rel = float(((a.double() - b.double()).abs() / (b.double().abs() + 1e-6)).max())
            worst = max(worst, rel)
            assert rel <= 2e-4, (it, nm, rel, a, b)
    assert state["step"] == 8 and fstate["step"] == 8
    print(f"worst relative difference over 8 steps: {worst:.2e}")


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("B", [1, 3, 5, 32])
def test_full_size_step_at_odd_batch_sizes(B):
    """Row counts that are not a multiple of the big tiles (M = B * 1024 and B * 256 at the two flow scales, B * HW in
    the U-Net): ragged tile tails, split-K slices and per-image segment reductions at a batch the goldens never use.
    B = 32 (config #5's per-rank shard) runs the whole step on the production variants (256-row direct-to-LDS tiles,
    big-tile weight grads, chain kernel at M = 32 768 / 8 192) against the oracle: one step, ~40 s of CPU."""
    torch.manual_seed(11 + B)
    np.random.seed(11 + B)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.model.dropout = 0.0
    cfg.training.batch_size = B
    dev = torch.device("cuda")
    cfg.device = dev
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    ost = O.OracleState(sd_s, cfg.optim.lr, cfg.model.ema_rate, cfg)
    ofst = O.OracleState(sd_f, cfg.flow.lr, cfg.flow.ema_rate, cfg)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    state, fstate, _ = utils.build_states(cfg, score, fl, dev)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); fl.train()
    for it in range(2 if B < 32 else 1):
        x = torch.rand(B, 3, 32, 32) * 2 - 1
        rec = RecordingRng()
        want = O.flow_step_fn_nll(cfg, O.get_sde(cfg), ost, ofst, x, rec)
        rng = ReplayRng(rec.events, dev)
        got = step_fn(state, fstate, x.to(dev), rng=rng)
        assert rng.done()
        for nm, a, b in zip(("loss", "score", "flow", "logp"), got, want):
            gu.assert_close(a, b, 2e-4, 1e-5, f"batch {B} step {it} {nm}")


@pytest.mark.timeout(2400)
def test_full_size_step_at_batch_128_matches_oracle():
    """BASELINE config #2's parity gate at ITS OWN batch: one full-size vp/CIFAR10/indm_nll optimisation step at batch
    128 -- the benchmark's exact launch geometry (M = 131072 / 32768 rows in the flow, 2048-workgroup chain kernel, 256-row
    tiles everywhere in the U-Net) -- against the CPU oracle on identical weights and noise.  The oracle evaluates the
    step in 16-image chunks (flow_step_fn_nll_chunked: whole-batch encoder BatchNorm statistics, everything else per
    image; pinned against the reference's goldens in tests/test_oracle_golden.py) because the eager double-backward
    graph of 128 images does not fit in host memory.  Gate: 2e-4 relative on all four per-image loss vectors; sampled
    parameters after the fused optimiser within 2.2 lr (AdamW's first step is lr * sign(g))."""
    B = 128
    torch.manual_seed(128)
    np.random.seed(128)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.model.dropout = 0.0
    cfg.training.batch_size = B
    dev = torch.device("cuda")
    cfg.device = dev
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    ost = O.OracleState(sd_s, cfg.optim.lr, cfg.model.ema_rate, cfg)
    ofst = O.OracleState(sd_f, cfg.flow.lr, cfg.flow.ema_rate, cfg)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    # the draw sequence of one step (kinds and per-image shapes), learnt from a 1-image forward of the oracle, then
    # drawn afresh for the whole batch; Poisson series lengths are per block and shared by all images
    probe = RecordingRng()
    O._nll_losses(cfg, O.get_sde(cfg), O.OracleState(sd_s, cfg.optim.lr, cfg.model.ema_rate, cfg),
                  O.OracleState(sd_f, cfg.flow.lr, cfg.flow.ema_rate, cfg), torch.rand(2, 3, 32, 32) * 2 - 1, probe, None)
    events = []
    for kind, v in probe.events:
        if kind == "poisson":
            events.append((kind, v))
        else:
            shape = (B,) + tuple(v.shape[1:])
            events.append((kind, (torch.rand(shape) if kind == "rand" else torch.randn(shape)).numpy()))
    x = torch.rand(B, 3, 32, 32) * 2 - 1
    state, fstate, _ = utils.build_states(cfg, score, fl, dev)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); fl.train()
    rng = ReplayRng(events, dev)
    got = step_fn(state, fstate, x.to(dev), rng=rng)
    assert rng.done()
    torch.cuda.synchronize()
    want = O.flow_step_fn_nll_chunked(cfg, O.get_sde(cfg), ost, ofst, x, events, chunk=16)
    for nm, a, b in zip(("loss", "score", "flow", "logp"), got, want):
        assert a.shape == (B,)
        gu.assert_close(a, b, 2e-4, 1e-5, f"batch 128 {nm}")
    ss, sf = score.reference_state_dict(prefix=""), fl.reference_state_dict(prefix="")
    for sd, o, lr in ((ss, ost, cfg.optim.lr), (sf, ofst, cfg.flow.lr)):
        for k in list(o.trainable)[:: max(1, len(o.trainable) // 12)]:
            diff = (sd[k].cpu().double() - o.P[k].detach().double()).abs()
            assert diff.max().item() <= 2.2 * lr, (k, diff.max().item())
            bad = int((diff > 1e-6 + 1e-4 * o.P[k].detach().abs().double()).sum().item())
            assert bad <= max(2, 5e-3 * diff.numel()), (k, bad, diff.numel())


@pytest.mark.timeout(3000)
@pytest.mark.parametrize("name,B", [("ve/CIFAR10/indm", 64), ("vp/CELEBA/indm_fid", 32), ("vp/CIFAR10/indm_nll#nrb8", 32)])
def test_full_size_step_at_the_per_rank_shard_matches_oracle(name, B):
    """BASELINE configs #3 / #4 / #5 at THEIR per-rank shards on 8 GPUs (512 / 8 = 64 images of 32x32 on the FIR / Fourier VE
    network; 256 / 8 = 32 images of 64x64 on the squeeze flow + 5-level U-Net, two-phase fid step; round 6: 256 / 8 = 32
    images on the deep score network of config #5, model.num_res_blocks = 8 -- 76 residual blocks): one full-size
    optimisation step on the production launch geometry -- 256-row direct-to-LDS tiles at M = 32 * 4096 and 64 * 1024,
    big-tile weight grads, the three-kernel GroupNorm the 64x64 maps fall back to, the LDS-tiled FIR kernels -- against the
    CPU oracle on identical weights and noise.  The oracle evaluates the step in chunks (flow_step_fn_nll_chunked /
    flow_step_fn_fid_chunked, both pinned to the reference's goldens).  Gate: 2e-4 relative on the per-image losses."""
    from indm_amd import hip
    torch.manual_seed(64)
    np.random.seed(64)
    cfg = configs.get_config(name.split("#")[0])
    if name.endswith("#nrb8"):
        cfg.model.num_res_blocks = 8
    cfg.model.dropout = 0.0
    cfg.training.batch_size = B
    dev = torch.device("cuda")
    cfg.device = dev
    S = cfg.data.image_size
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    fresh = lambda: (O.OracleState(sd_s, cfg.optim.lr, cfg.model.ema_rate, cfg),  # noqa: E731
                     O.OracleState(sd_f, cfg.flow.lr, cfg.flow.ema_rate, cfg))
    ost, ofst = fresh()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    fid = not cfg.training.likelihood_weighting
    # the draw sequence of one step, learnt from a 2-image step of the oracle, then drawn afresh for the whole batch
    probe = RecordingRng()
    p1, p2 = fresh()
    (O.flow_step_fn_fid if fid else O.flow_step_fn_nll)(cfg, O.get_sde(cfg), p1, p2, torch.rand(2, 3, S, S) * 2 - 1, probe)
    del p1, p2
    events = []
    for kind, v in probe.events:
        if kind == "poisson":
            events.append((kind, v))
        else:
            shape = (B,) + tuple(v.shape[1:])
            events.append((kind, (torch.rand(shape) if kind == "rand" else torch.randn(shape)).numpy()))
    x = torch.rand(B, 3, S, S) * 2 - 1
    state, fstate, _ = utils.build_states(cfg, score, fl, dev)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); fl.train()
    rng = ReplayRng(events, dev)
    drivers = hip.block_drivers()
    hip.block_drivers(False)      # op-level composition: every dense launch is visible to the class recorder below
    hip.profile_start()
    try:
        got = step_fn(state, fstate, x.to(dev), rng=rng)
    finally:
        classes = hip.profile_stop()
        hip.block_drivers(drivers)
    assert rng.done()
    # the production variants ran: 256-row direct-to-LDS tiles, the big tile (weight grads), and -- CelebA -- GroupNorm on
    # 64 x 64 maps through the three-kernel passes
    names = " ".join(classes)
    assert "glds256x128" in names and "big256" in names, sorted(classes)
    if S == 64:
        assert hip.lib().indm_groupnorm_fused(64 * 64, 128, 32) == 0 and hip.lib().indm_groupnorm_fused(32 * 32, 128, 32) == 1
    chunked = O.flow_step_fn_fid_chunked if fid else O.flow_step_fn_nll_chunked
    want = chunked(cfg, O.get_sde(cfg), ost, ofst, x, events, chunk=8 if S == 64 else 16)
    for nm, a, b in zip(("loss", "score", "flow", "logp"), got, want):
        assert a.shape == (B,)
        gu.assert_close(a, b, 2e-4, 1e-5, f"{name} batch {B} {nm}")
    ss, sf = score.reference_state_dict(prefix=""), fl.reference_state_dict(prefix="")
    for sd, o, lr in ((ss, ost, cfg.optim.lr), (sf, ofst, cfg.flow.lr)):
        for k in list(o.trainable)[:: max(1, len(o.trainable) // 12)]:
            diff = (sd[k].cpu().double() - o.P[k].detach().double()).abs()
            assert diff.max().item() <= 2.2 * lr, (k, diff.max().item())

    # ---- bf16 arm (BASELINE configs #3 / #4 are "bf16 compute / fp32 accumulate"): the SAME step from the same initial
    # weights and noise with bf16 MFMA inputs and bf16 STORAGE of the flow's 512-wide tensors (block drivers on), against
    # the fp32 oracle result computed above.  Stated tolerance: every per-image loss within 3 % (+1e-2), the batch mean
    # within 1 %.  Which scales run on the F16 kernels is asserted, not assumed: VE / CIFAR both (3 @ 32^2, 12 @ 16^2);
    # CelebA's first scale (12 @ 32^2) does, its second (48 @ 16^2: C > 12) stays on fp32 tensors with bf16 MFMA inputs
    # ("bf16" for config #4 is therefore a MIXED storage path: scale 0 bf16, scale 1 fp32).
    del state, fstate, step_fn, score, fl
    torch.cuda.empty_cache()
    score = utils.create_model(cfg, dev)
    score.load_reference_state_dict(sd_s)
    fl = create_flow_model(cfg).to(dev)
    fl.load_reference_state_dict(sd_f)
    score, fl = score.to(dev), fl.to(dev)
    cfg.compute_dtype = "bf16"
    try:
        state, fstate, _ = utils.build_states(cfg, score, fl, dev)
        step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
        assert hip.compute_dtype() == "bf16" and hip.block_drivers() and hip.storage_dtype() == "bf16"
        score.train(); fl.train()
        hip.f16_stats(reset=True)
        hip.h16_stats(reset=True)
        rng = ReplayRng(events, dev)
        got16 = step_fn(state, fstate, x.to(dev), rng=rng)
        assert rng.done()
        stats = hip.f16_stats(reset=True)
        ustats = hip.h16_stats(reset=True)
    finally:
        hip.set_compute_dtype("f32")
    # round 6: the score network's trunk ran on bf16-STORED activations (csrc/unet_h16*.hip): convolutions / data grads /
    # weight grads on the H16 kernels; only the strided data grads of the VE input pyramid may fall back to fp32 copies
    print(f"{name}: H16 launches {ustats}")
    assert hip.h16() is False and ustats["conv"] >= 150 and ustats["wgrad"] >= 80, ustats
    assert ustats["conv_fallback"] <= (8 if S == 32 else 0) and ustats["wgrad_fallback"] == 0, ustats
    if S == 32:
        assert stats == {(3, 32, 32): [16, 0], (12, 16, 16): [16, 0]}, stats
    else:
        # (round 6: the 48-channel scale -- 9 C = 432 taps -- runs the wide form of the F16 drivers: im2col operand + 64-row GEMMs)
        assert stats == {(12, 32, 32): [16, 0], (48, 16, 16): [16, 0]}, stats
    for nm, a, b in zip(("loss", "score", "flow", "logp"), got16, want):
        a, b = a.double().cpu(), b.double()
        assert torch.isfinite(a).all()
        if nm == "loss":   # the objective: per image, and its batch mean
            err = ((a - b).abs() / (b.abs() + 1e-2 / 3e-2)).max().item()
            print(f"{name} batch {B} bf16 (H16 + F16 storage) vs the fp32 oracle: worst per-image loss deviation {err:.2e}, "
                  f"batch mean {abs(float(a.mean() - b.mean())) / abs(float(b.mean())):.2e}")
            assert err <= 3e-2, (name, "bf16", nm, err, a, b)
            assert abs(float(a.mean() - b.mean())) <= 1e-2 * abs(float(b.mean())), (name, "bf16 mean", a.mean(), b.mean())
        else:              # its three terms (they cancel in part): relative to the largest entry of the vector
            assert (a - b).abs().max().item() <= 3e-2 * b.abs().max().item() + 1e-2, (name, "bf16", nm, a, b)


@pytest.mark.timeout(1800)
def test_bf16_storage_loss_curve_on_both_scales_vs_oracle():
    """The F16 (bf16-storage) kernels of BOTH flow scales against the oracle, end to end: at batch 2 the second scale
    (12 @ 16^2, M = 512) is below the F16 kernels' M >= 1024 and silently stays on fp32 tensors, so the batch-2 bf16 gates
    above only anchor scale 0.  Here batch 4 (M = 4096 / 1024): THREE consecutive optimisation steps of the full-size
    vp/CIFAR10/indm_nll networks in bf16 mode vs the fp32 CPU oracle on the same batches and noise, with the coverage counter
    asserting that all 32 blocks of every step ran on F16 kernels.  Stated tolerance: batch-mean loss within 1 %, every
    per-image loss within 4 % (+1e-2)."""
    from indm_amd import hip
    torch.manual_seed(23)
    np.random.seed(23)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.model.dropout = 0.0
    cfg.training.batch_size = 4
    dev = torch.device("cuda")
    cfg.device = dev
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    ost = O.OracleState(sd_s, cfg.optim.lr, cfg.model.ema_rate, cfg)
    ofst = O.OracleState(sd_f, cfg.flow.lr, cfg.flow.ema_rate, cfg)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg.compute_dtype = "bf16"
    worst_mean, worst_img = 0.0, 0.0
    try:
        state, fstate, _ = utils.build_states(cfg, score, fl, dev)
        step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
        assert hip.compute_dtype() == "bf16" and hip.storage_dtype() == "bf16"
        score.train(); fl.train()
        for it in range(3):
            x = torch.rand(4, 3, 32, 32) * 2 - 1
            rec = RecordingRng()
            want = O.flow_step_fn_nll(cfg, O.get_sde(cfg), ost, ofst, x, rec)
            hip.f16_stats(reset=True)
            hip.h16_stats(reset=True)
            rng = ReplayRng(rec.events, dev)
            got = step_fn(state, fstate, x.to(dev), rng=rng)
            assert rng.done()
            assert hip.f16_stats(reset=True) == {(3, 32, 32): [16, 0], (12, 16, 16): [16, 0]}
            ustats = hip.h16_stats(reset=True)   # ... and the score network's trunk on bf16-stored activations (round 6)
            assert ustats["conv"] >= 150 and ustats["wgrad"] >= 80 and ustats["conv_fallback"] == 0 and ustats["wgrad_fallback"] == 0, ustats
            for nm, a, b in zip(("loss", "score", "flow", "logp"), got, want):
                a, b = a.double().cpu(), b.double()
                assert torch.isfinite(a).all()
                if nm == "loss":
                    rel_mean = abs(float(a.mean() - b.mean())) / abs(float(b.mean()))
                    err_img = float(((a - b).abs() / (b.abs() + 0.25)).max())
                    worst_mean, worst_img = max(worst_mean, rel_mean), max(worst_img, err_img)
                    assert rel_mean <= 1e-2, (it, nm, rel_mean, a, b)
                    assert err_img <= 4e-2, (it, nm, err_img, a, b)
                else:
                    assert (a - b).abs().max().item() <= 4e-2 * b.abs().max().item() + 1e-2, (it, nm, a, b)
    finally:
        hip.set_compute_dtype("f32")
    print(f"bf16 storage, both scales, 3 steps: worst batch-mean deviation {worst_mean:.2e}, worst per-image {worst_img:.2e}")


@pytest.mark.timeout(1800)
def test_config5_full_training_step_matches_oracle():
    """BASELINE config #5's TRAINING step (vp/CIFAR10/indm_nll with model.num_res_blocks = 8: 76 residual blocks, 10
    attention blocks) at full size, batch 2, HIP path vs the CPU oracle on the same weights and noise: the four per-image
    loss vectors (1e-3 relative) and a sample of the updated parameters of both networks."""
    torch.manual_seed(58)
    np.random.seed(58)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.model.num_res_blocks = 8
    cfg.model.dropout = 0.0
    cfg.training.batch_size = 2
    dev = torch.device("cuda")
    cfg.device = dev
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    x = (torch.rand(2, 3, 32, 32) * 2 - 1)
    ost = O.OracleState(sd_s, cfg.optim.lr, cfg.model.ema_rate, cfg)
    ofst = O.OracleState(sd_f, cfg.flow.lr, cfg.flow.ema_rate, cfg)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    rec = RecordingRng()
    want = O.flow_step_fn_nll(cfg, O.get_sde(cfg), ost, ofst, x, rec)
    state, fstate, _ = utils.build_states(cfg, score, fl, dev)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); fl.train()
    rng = ReplayRng(rec.events, dev)
    got = step_fn(state, fstate, x.to(dev), rng=rng)
    assert rng.done()
    for nm, a, b in zip(("loss", "score", "flow", "logp"), got, want):
        gu.assert_close(a, b, 1e-3, 1e-5, f"config #5 training step {nm}")
    ss, sf = score.reference_state_dict(prefix=""), fl.reference_state_dict(prefix="")
    for sd, o, lr in ((ss, ost, cfg.optim.lr), (sf, ofst, cfg.flow.lr)):
        for k in list(o.trainable)[:: max(1, len(o.trainable) // 12)]:
            diff = (sd[k].cpu().double() - o.P[k].detach().double()).abs()
            assert diff.max().item() <= 2.2 * lr, (k, diff.max().item())
            bad = int((diff > 1e-6 + 1e-4 * o.P[k].detach().abs().double()).sum().item())
            assert bad <= max(2, 5e-3 * diff.numel()), (k, bad, diff.numel())


@pytest.mark.timeout(1500)
def test_full_size_nelbo_within_north_star_tolerance():
    """NELBO bits/dim of the full-size model (flow eval power-series log-det with n+20 terms, score loss integrand,
    Hutchinson divergence, prior, residual) on the HIP path vs the CPU oracle on the same noise: the north-star asks
    for +-0.01 bits/dim; the fp32 paths agree to 1e-3."""
    torch.manual_seed(11)
    np.random.seed(11)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    dev = torch.device("cuda")
    cfg.device = dev
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    x = (torch.rand(2, 3, 32, 32) * 2 - 1)
    rec = RecordingRng()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    want, want_r = O.nelbo_fn(sd_s, sd_f, cfg, O.get_sde(cfg), x, rec)
    score.eval(); fl.eval()
    fn = likelihood.get_elbo_fn(cfg, sde_lib.get_sde(cfg), inverse_scaler=utils.get_data_inverse_scaler(cfg))
    rng = ReplayRng(rec.events, dev)
    got, got_r = fn(score, fl, x.to(dev), rng=rng)
    assert rng.done()
    assert (got.cpu() - want).abs().max().item() < 1e-3, (got, want)          # bits/dim
    assert (got_r.cpu() - want_r).abs().max().item() < 1e-3, (got_r, want_r)


@pytest.mark.timeout(1500)
def test_config5_deep_score_net_and_pc_sampler():
    """BASELINE config #5: configs/vp/CIFAR10/indm_nll.py with model.num_res_blocks = 8 and sampling.method = pc
    (Euler-Maruyama predictor, no corrector), at FULL size.  (a) structure: 8 residual blocks per resolution,
    num_res_for_upsampling = 9 (ncsnpp.py:194-198); (b) score network output + input gradient vs the CPU oracle at batch 2;
    (c) three PC-sampler steps + the flow's fixed-point inverse vs the oracle on the same recorded noise (the eager launch
    sequence; the hipGraph-captured step is bit-identical to it, tests/test_parity_gpu.py)."""
    from indm_amd import sampling
    from indm_amd.ncsnpp import AttnBlockpp, ResnetBlockBigGANpp
    torch.manual_seed(5)
    np.random.seed(5)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.model.num_res_blocks = 8
    cfg.model.dropout = 0.0
    cfg.sampling.method, cfg.sampling.predictor, cfg.sampling.corrector, cfg.sampling.num_scales = "pc", "euler_maruyama", "none", 3
    dev = torch.device("cuda")
    cfg.device = dev
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    nrb = sum(isinstance(m, ResnetBlockBigGANpp) for m in score.all_modules)
    # down path 4 x 8 + 3 aux; middle 2; up path 4 x 9 + 3 aux
    assert nrb == 4 * 8 + 3 + 2 + 4 * 9 + 3, nrb
    assert sum(isinstance(m, AttnBlockpp) for m in score.all_modules) == 8 + 1 + 1
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    score.eval(); fl.eval()
    # (b) score function and its input gradient
    x = (torch.rand(2, 3, 32, 32) * 2 - 1)
    t = torch.tensor([0.2, 0.85])
    gy = torch.randn(2, 3, 32, 32)
    xo = x.clone().requires_grad_(True)
    so = O.get_score_fn(sd_s, cfg, O.get_sde(cfg), train=False)(xo, t)
    (gxo,) = torch.autograd.grad(so, xo, gy)
    xg = x.to(dev).requires_grad_(True)
    sg = losses.get_score_fn(cfg, sde_lib.get_sde(cfg), score, train=False, continuous=True)(xg, t)
    (gxg,) = torch.autograd.grad(sg, xg, gy.to(dev))
    gu.assert_close(sg, so.detach(), 2e-4, 2e-5, "nrb=8 score")
    gu.assert_close(gxg, gxo, 5e-4, 5e-5, "nrb=8 score input-grad")
    # (c) PC sampler
    rec = RecordingRng()
    want = O.pc_sampler(sd_s, sd_f, cfg, O.get_sde(cfg), (2, 3, 32, 32), rec, cfg.sampling.truncation_time)
    fn = sampling.get_sampling_fn(cfg, sde_lib.get_sde(cfg), (2, 3, 32, 32), utils.get_data_inverse_scaler(cfg),
                                  cfg.sampling.truncation_time, device=dev)
    rng = ReplayRng(rec.events, dev)
    got = fn(score, fl, rng=rng)
    assert rng.done() and got[2] == want[2]
    gu.assert_close(got[0], want[0], 5e-4, 5e-4, "config #5 pc before-flow latents")
    gu.assert_close(got[1], want[1], 3e-3, 3e-3, "config #5 pc samples")


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name", ["ve/CIFAR10/indm", "vp/CELEBA/indm_fid"])
def test_bf16_mode_full_size_step_within_tolerance(name):
    """BASELINE configs #3 / #4 ask for bf16 compute / fp32 accumulate.  One full-size training step in bf16 mode vs the
    fp32 CPU oracle on the same weights and noise.  Stated tolerance: per-image losses within 3 % relative (+1e-2
    absolute) -- bf16 carries 8 significant bits through ~60 chained contractions; parameters move by at most
    2.2 * lr (AdamW's first step).  fp32 stays the parity gate and the headline."""
    from indm_amd import hip
    torch.manual_seed(9)
    np.random.seed(9)
    cfg = configs.get_config(name)
    cfg.model.dropout = 0.0
    cfg.training.batch_size = 2
    dev = torch.device("cuda")
    cfg.device = dev
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    x = (torch.rand(2, 3, cfg.data.image_size, cfg.data.image_size) * 2 - 1)
    ost = O.OracleState(sd_s, cfg.optim.lr, cfg.model.ema_rate, cfg)
    ofst = O.OracleState(sd_f, cfg.flow.lr, cfg.flow.ema_rate, cfg)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    rec = RecordingRng()
    ostep = O.flow_step_fn_nll if cfg.training.likelihood_weighting else O.flow_step_fn_fid
    want = ostep(cfg, O.get_sde(cfg), ost, ofst, x, rec)
    cfg.compute_dtype = "bf16"
    try:
        state, fstate, _ = utils.build_states(cfg, score, fl, dev)
        step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
        assert hip.compute_dtype() == "bf16"
        score.train(); fl.train()
        rng = ReplayRng(rec.events, dev)
        got = step_fn(state, fstate, x.to(dev), rng=rng)
        assert rng.done()
    finally:
        hip.set_compute_dtype("f32")
    for nm, a, b in zip(("loss", "score", "flow", "logp"), got, want):
        assert torch.isfinite(a).all()
        err = (a - b).abs().max().item()
        assert err <= 3e-2 * b.abs().max().item() + 1e-2, (name, nm, a, b)
    sf = fl.reference_state_dict(prefix="")
    for k in list(ofst.trainable)[:: max(1, len(ofst.trainable) // 8)]:
        assert (sf[k].cpu() - ofst.P[k].detach()).abs().max().item() <= 2.2 * cfg.flow.lr + 1e-6, k


@pytest.mark.timeout(1800)
def test_bf16_mode_loss_curve_tracks_fp32_oracle():
    """Drift over several steps in bf16 compute mode: EIGHT consecutive optimisation steps of the full-size
    vp/CIFAR10/indm_nll networks with the MFMA inputs rounded to bf16, against the fp32 CPU oracle stepping its own fp32
    weights on the same batches and noise.  Stated tolerance: the batch-mean loss of every step within 1 % of the oracle's,
    every per-image loss within 4 % (+1e-2) -- the weights of the two runs differ by then by what 8 AdamW steps of
    bf16-rounded gradients produce, so this bounds the accumulated effect, not a single step's rounding."""
    from indm_amd import hip
    torch.manual_seed(21)
    np.random.seed(21)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.model.dropout = 0.0
    cfg.training.batch_size = 2
    dev = torch.device("cuda")
    cfg.device = dev
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    ost = O.OracleState(sd_s, cfg.optim.lr, cfg.model.ema_rate, cfg)
    ofst = O.OracleState(sd_f, cfg.flow.lr, cfg.flow.ema_rate, cfg)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg.compute_dtype = "bf16"
    worst_mean, worst_img = 0.0, 0.0
    try:
        state, fstate, _ = utils.build_states(cfg, score, fl, dev)
        step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
        assert hip.compute_dtype() == "bf16"
        score.train(); fl.train()
        for it in range(8):
            x = torch.rand(2, 3, 32, 32) * 2 - 1
            rec = RecordingRng()
            want = O.flow_step_fn_nll(cfg, O.get_sde(cfg), ost, ofst, x, rec)
            rng = ReplayRng(rec.events, dev)
            got = step_fn(state, fstate, x.to(dev), rng=rng)
            assert rng.done()
            a, b = got[0].double(), want[0].double()
            assert torch.isfinite(a).all()
            rel_mean = abs(float(a.mean() - b.mean())) / abs(float(b.mean()))
            err_img = float(((a - b).abs() / (b.abs() + 0.25)).max())
            worst_mean, worst_img = max(worst_mean, rel_mean), max(worst_img, err_img)
            assert rel_mean <= 1e-2, (it, rel_mean, a, b)
            assert err_img <= 4e-2, (it, err_img, a, b)
    finally:
        hip.set_compute_dtype("f32")
    print(f"bf16 loss curve over 8 steps: worst batch-mean deviation {worst_mean:.2e}, worst per-image {worst_img:.2e}")


@pytest.mark.timeout(1500)
def test_bf16_mode_nelbo_relative_tolerance():
    """NELBO bits/dim of the full-size VP model in bf16 compute mode vs the fp32 CPU oracle on the same noise.  Stated
    tolerance: 2e-3 RELATIVE (measured 1e-4 ... 9e-4 on random-init weights, whose NELBO is 12 ... 29 bits/dim; at a
    trained model's ~3 bits/dim that is < 0.006 bits/dim, inside the +-0.01 north-star band, which is an fp32 claim and is
    tested in fp32 above)."""
    from indm_amd import hip
    torch.manual_seed(13)
    np.random.seed(13)
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    dev = torch.device("cuda")
    cfg.device = dev
    score = utils.create_model(cfg, dev)
    fl = create_flow_model(cfg).to(dev)
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.dim() > 1 and p.abs().max() < 1e-6:
                p.copy_(torch.randn_like(p) * 0.02)
    sd_s = {k: v.cpu() for k, v in score.reference_state_dict(prefix="").items()}
    sd_f = {k: v.cpu() for k, v in fl.reference_state_dict(prefix="").items()}
    x = (torch.rand(2, 3, 32, 32) * 2 - 1)
    rec = RecordingRng()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    want, _ = O.nelbo_fn(sd_s, sd_f, cfg, O.get_sde(cfg), x, rec)
    score.eval(); fl.eval()
    fn = likelihood.get_elbo_fn(cfg, sde_lib.get_sde(cfg), inverse_scaler=utils.get_data_inverse_scaler(cfg))
    hip.set_compute_dtype("bf16")
    try:
        got, _ = fn(score, fl, x.to(dev), rng=ReplayRng(rec.events, dev))
    finally:
        hip.set_compute_dtype("f32")
    assert ((got.cpu() - want).abs() / want.abs()).max().item() < 2e-3, (got, want)
