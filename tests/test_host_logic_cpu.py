"""Host-logic tests on CPU: the product's module graph / explicit-VJP log-det engine / step functions are run
with the kernel layer replaced by its plain-PyTorch statement (tests/_emu.py via the emulate_hip fixture) and
compared with the REFERENCE's golden outputs.  This validates everything above the C ABI without a GPU; the
`-m gpu` suite repeats the same cases on the real kernels."""
import pytest

import parity_cases as pc


@pytest.mark.parametrize("fx", pc.FIXTURES)
@pytest.mark.parametrize("case", pc.ALL_CASES, ids=lambda f: f.__name__)
def test_case(case, fx, emulate_hip):
    case(fx, "cpu")


def test_pf_ode(emulate_hip):
    pc.case_pf_ode("mini_vp_nll", "cpu")


@pytest.mark.parametrize("case", pc.EXTRA_CASES, ids=lambda f: f.__name__)
def test_extra_case(case, emulate_hip):
    case("cpu")


@pytest.mark.parametrize("case", pc.R2_CASES, ids=lambda f: f.__name__)
def test_r2_case(case, emulate_hip):
    case("cpu")


@pytest.mark.parametrize("tag", ["mb2_nll", "mb2_fid"])
def test_micro_batch_golden(tag, emulate_hip):
    pc.case_micro_batch_step(tag, "cpu")


def test_sampling_lib_cache(emulate_hip, tmp_path):
    """sampling_lib.get_samples: first call samples and writes the reference's npz files, second call resumes from
    them (flow inverse in chunks of 16), pc_denoise resumes from the for-search snapshot."""
    import numpy as np
    import torch
    import golden_utils as gu
    from indm_amd import sampling, sampling_lib, sde_lib, utils
    d, cfg, score, fl = pc.build("mini_ve", "cpu")
    cfg.model.num_scales = 4
    cfg.sampling.method, cfg.sampling.num_scales = "pc", 3
    score.eval(); fl.eval()
    sde = sde_lib.get_sde(cfg)
    inv, scaler = utils.get_data_inverse_scaler(cfg), utils.get_data_scaler(cfg)
    shape = (16, 3, 32, 32)
    fn = sampling.get_sampling_fn(cfg, sde, shape, inv, cfg.sampling.truncation_time, device="cpu")
    sd, tsd = str(tmp_path / "samples"), str(tmp_path / "samples" / "t1")
    torch.manual_seed(0)
    b1, a1 = sampling_lib.get_samples(cfg, score, fl, fn, 3, 0, sd, 1.0, inv, tsd, scaler)
    f = np.load(sd + "/samples_0_before_flow.npz")["samples"]
    g = np.load(tsd + "/samples_0.npz")["samples"]
    s = np.load(sd + "/samples_0_before_flow_for_search.npz")["samples"]
    assert f.shape == g.shape == s.shape == (16, 32, 32, 3) and g.dtype == np.uint8 and f.dtype != np.uint8
    assert np.allclose(f, b1.permute(0, 2, 3, 1).numpy() * 255.0)
    assert np.array_equal(g, np.clip(a1.permute(0, 2, 3, 1).numpy() * 255.0, 0, 255).astype(np.uint8))
    # second call: cached latents -> flow inverse in chunks of 16 -> same images (up to the fixed-point tolerance)
    os_remove = __import__("os").remove
    os_remove(tsd + "/samples_0.npz")
    from indm_amd import rng as irng
    irng.manual_seed(5)     # the flow inverse draws h from the prior with the process-wide (advancing) noise source
    b2, a2 = sampling_lib.get_samples(cfg, score, fl, fn, 3, 0, sd, 1.0, inv, tsd, scaler)
    from indm_amd.flow import flow_forward
    irng.manual_seed(5)
    with torch.no_grad():
        want = inv(flow_forward(cfg, fl, scaler(b2), log_det=None, reverse=True)[0])
    assert torch.allclose(b2, b1, rtol=1e-6, atol=1e-6 * float(b1.abs().max())) and torch.equal(a2, want)
    assert __import__("os").path.exists(tsd + "/samples_0.npz")
    # pc_denoise: resumes from the for-search snapshot with the denoising sampler
    cfg.sampling.pc_denoise, cfg.sampling.need_sample, cfg.sampling.pc_denoise_time = True, False, 0.0
    fn2 = sampling.get_sampling_fn(cfg, sde, shape, inv, cfg.sampling.truncation_time, device="cpu")
    b3, a3 = sampling_lib.get_samples(cfg, score, fl, fn2, 3, 0, sd, 1.0, inv, tsd, scaler)
    assert __import__("os").path.exists(sd + "/samples_0_before_flow_denoise_0.0.npz")
    assert np.load(tsd + "/samples_0_denoise_0.0.npz")["samples"].dtype == np.uint8
    assert b3.shape == a3.shape == shape


@pytest.mark.parametrize("fx", ["mini_vp_nll", "mini_vp_fid"])
def test_micro_batches_and_loss_readback(fx, emulate_hip):
    """optim.num_micro_batch = 2 (losses.py:287-309): every micro-batch's loss vectors are read back (enqueued before its
    backward pass) and concatenated in order; total = score + flow + logp; steps advance once."""
    import torch
    import golden_utils as gu
    from indm_amd import losses, sde_lib, utils
    from indm_amd.rng import DeviceRng
    if fx not in pc.FIXTURES:
        pytest.skip("fixture not present")
    d, cfg, score, fl = pc.build(fx, "cpu")
    cfg.optim.num_micro_batch = 2
    state, flow_state, _ = utils.build_states(cfg, score, fl, "cpu")
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); fl.train()
    x = torch.from_numpy(d["step0/x"])
    out = step_fn(state, flow_state, x, rng=DeviceRng("cpu", seed=7))
    assert len(out) == 4
    for v in out:
        assert v.shape == (x.shape[0],) and not v.is_cuda and torch.isfinite(v).all()
    if cfg.training.likelihood_weighting:
        assert torch.allclose(out[0], out[1] + out[2] + out[3], rtol=1e-5, atol=1e-5)
    assert state["step"] == 1 and flow_state["step"] == 1


def test_default_noise_source_advances(emulate_hip):
    """callers that pass no rng= must not replay a constant seed -- two consecutive sampling_fn /
    nelbo_fn / loss_fn calls draw different noise (the reference uses the advancing global torch generator), and
    rng.manual_seed() makes a run reproducible."""
    import torch
    from indm_amd import likelihood, losses, rng, sampling, sde_lib, utils
    d, cfg, score, fl = pc.build("mini_vp_nll", "cpu")
    cfg.sampling.method, cfg.sampling.num_scales = "pc", 2
    sde = sde_lib.get_sde(cfg)
    score.eval(); fl.eval()
    inv = utils.get_data_inverse_scaler(cfg)
    fn = sampling.get_sampling_fn(cfg, sde, (2, 3, 32, 32), inv, cfg.sampling.truncation_time, device="cpu")
    rng.manual_seed(11)
    b1, _, _ = fn(score, fl)
    b2, _, _ = fn(score, fl)
    assert not torch.equal(b1, b2)
    rng.manual_seed(11)
    b3, _, _ = fn(score, fl)
    assert torch.equal(b1, b3)
    x = torch.from_numpy(d["x"])
    nelbo = likelihood.get_elbo_fn(cfg, sde, inverse_scaler=inv)
    n1, n2 = nelbo(score, fl, x)[0], nelbo(score, fl, x)[0]
    assert not torch.equal(n1, n2)
    loss_fn = losses.get_sde_loss_fn(cfg, sde, train=False)
    with torch.no_grad():
        l1, l2 = loss_fn(score, x), loss_fn(score, x)
    assert not torch.equal(l1, l2)
    # per-(step, rank) streams of the training step do not collide the way step + 7919*rank did
    assert rng.mix64(42, 0, 1) != rng.mix64(42, 7919, 0) and rng.mix64(1, 2) != rng.mix64(2, 1)


def test_warmup_ramps_from_the_configured_lr(emulate_hip):
    """optim.warmup > 0 -- lr_k = config.optim.lr * min(k / warmup, 1) for BOTH optimisers
    (losses.py:51-57: the default lr of optimize_fn is config.optim.lr), not stuck at the 0 written at step 0."""
    import torch
    from indm_amd import losses, sde_lib, utils
    from indm_amd.rng import DeviceRng
    d, cfg, score, fl = pc.build("mini_vp_nll", "cpu")
    cfg.optim.warmup = 2
    state, flow_state, _ = utils.build_states(cfg, score, fl, "cpu")
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
    score.train(); fl.train()
    x = torch.from_numpy(d["step0/x"])
    seen = []
    for k in range(3):
        before = state["optimizer"].flat.p.clone()
        step_fn(state, flow_state, x, rng=DeviceRng("cpu", seed=k))
        seen.append((state["optimizer"].param_groups[0]["lr"], flow_state["optimizer"].param_groups[0]["lr"],
                     float((state["optimizer"].flat.p - before).abs().max())))
    want = [cfg.optim.lr * min(k / 2, 1.0) for k in range(3)]
    for (ls, lf, moved), w in zip(seen, want):
        assert abs(ls - w) < 1e-12 and abs(lf - w) < 1e-12, (seen, want)
    assert seen[0][2] == 0.0 and seen[1][2] > 0.0 and seen[2][2] > 0.0


def test_eval_step_still_updates_ema(emulate_hip):
    """train=False: flow_step_fn_nll still advances step and EMA (losses.py:314-318); flow_step_fn_fid does nothing."""
    import torch
    from indm_amd import losses, sde_lib, utils
    for fx, moved in (("mini_vp_nll", True), ("mini_vp_fid", False)):
        d, cfg, score, fl = pc.build(fx, "cpu")
        state, flow_state, _ = utils.build_states(cfg, score, fl, "cpu")
        step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=False)
        out = step_fn(state, flow_state, torch.from_numpy(d["step0/x"]))
        assert all(float(v.abs().sum()) == 0.0 for v in out)
        assert (state["step"], flow_state["step"]) == ((1, 1) if moved else (0, 0))
        assert state["ema"].num_updates == (1 if moved else 0) and flow_state["ema"].num_updates == (1 if moved else 0)


def test_initialisers_match_the_reference():
    """D5: default_init / variance_scaling (models/layers.py:54-91).  The product's constructors consume the torch
    generator exactly like the reference's, so a seeded NCSN++ is BIT-identical; the flow (different draw order) must
    match in distribution: per-tensor second moment and range of every parameter."""
    import numpy as np
    import torch
    import golden_utils as gu
    from indm_amd import flow as iflow
    from indm_amd import ncsnpp
    d = gu.load("mini_r2")
    torch.manual_seed(99)
    for i, shp in enumerate(d["init/shapes"]):
        shp = tuple(int(v) for v in shp if v > 0)
        for scale in (1.0, 0.1, 0.0):
            w = ncsnpp.default_init_(shp, scale).double()
            got = np.asarray([float(w.mean()), float(w.var(unbiased=False)), float(w.min()), float(w.max())])
            assert np.allclose(got, d[f"init/{i}/scale{scale}"], rtol=1e-12, atol=0), (shp, scale, got)
    cfg = gu.mini_config("mini_vp_nll")
    torch.manual_seed(7)
    score = ncsnpp.NCSNpp(cfg)
    torch.manual_seed(8)
    fl = iflow.create_flow_model(cfg)
    for k, v in score.reference_state_dict(prefix="").items():
        v = v.double()
        got = np.asarray([float(v.sum()), float(v.pow(2).sum()), float(v.min()), float(v.max())])
        assert np.array_equal(got, d[f"init_net/score/{k}"]), k
    for k, v in fl.reference_state_dict(prefix="").items():
        ref = d[f"init_net/flow/{k}"]
        v = v.double()
        n = v.numel()
        if n < 64 or not v.is_floating_point():
            assert abs(float(v.sum()) - ref[0]) <= 1e-6 + 6.0 * (ref[1] + 1e-12) ** 0.5, k    # |sum| <~ 6 sigma sqrt(n)
            continue
        assert abs(float(v.pow(2).sum()) - ref[1]) <= (8.0 / n ** 0.5) * ref[1] + 1e-9, (k, float(v.pow(2).sum()), ref[1])
        assert float(v.min()) >= 2.5 * min(ref[2], 0.0) - 1e-6 and float(v.max()) <= 2.5 * max(ref[3], 0.0) + 1e-6, k


def test_fid_and_inception_score_statistics():
    """closed forms: FID of two Gaussians' exact statistics; IS of one-hot / uniform predictions"""
    import numpy as np
    from indm_amd import evaluation as ev
    rs = np.random.RandomState(0)
    d = 6
    A, B = rs.randn(d, d), rs.randn(d, d)
    s1, s2 = A @ A.T + np.eye(d), B @ B.T + np.eye(d)
    m1, m2 = rs.randn(d), rs.randn(d)
    assert abs(ev.frechet_distance(m1, s1, m1, s1)) < 1e-8
    # commuting covariances: Tr(S1 + S2 - 2 sqrt(S1 S2)) = sum (sqrt(l1) - sqrt(l2))^2
    l1, l2 = rs.rand(d) + 0.5, rs.rand(d) + 0.5
    want = ((m1 - m2) ** 2).sum() + ((np.sqrt(l1) - np.sqrt(l2)) ** 2).sum()
    assert abs(ev.frechet_distance(m1, np.diag(l1), m2, np.diag(l2)) - want) < 1e-9
    assert abs(ev.frechet_distance(m1, s1, m2, s2) - ev.frechet_distance(m2, s2, m1, s1)) < 1e-8   # symmetric
    x = rs.randn(4000, d) @ A.T + m1
    mu, sg = ev.feature_statistics(x)
    assert np.allclose(mu, x.mean(0)) and np.allclose(sg, np.cov(x, rowvar=False))
    assert ev.fid_from_features(x, x) < 1e-8 and ev.fid_from_features(x, x + 3.0) > 9.0 * d - 1e-6
    K = 10
    onehot = np.full((5000, K), -50.0)
    onehot[np.arange(5000), np.arange(5000) % K] = 50.0
    assert abs(ev.inception_score_from_logits(onehot) - K) < 1e-6          # confident and balanced: IS = #classes
    assert abs(ev.inception_score_from_logits(np.zeros((100, K))) - 1.0) < 1e-12
    assert len(ev.inception_scores_10_splits(np.tile(onehot, (10, 1)))) == 10
    with pytest.raises(ValueError):
        ev.inception_scores_10_splits(onehot)


@pytest.mark.parametrize("tag", sorted(pc.R3_TAGS))
def test_reconstruction_loss_golden(tag, emulate_hip):
    """loss_fn(recon_loss=True[, st=True]) against the reference's goldens (mini_r3.npz)"""
    pc.case_reconstruction_loss(tag, "cpu")


@pytest.mark.parametrize("tag", ["st_nll", "st_fid", "st_fid_mb2"])
def test_training_st_golden(tag, emulate_hip):
    """training.st = True steps against the reference's goldens (mini_r3.npz)"""
    pc.case_training_st_step(tag, "cpu")


def test_probability_flow_sampler_golden(emulate_hip):
    """sampling.probability_flow = True against the reference's goldens (mini_r4.npz)"""
    pc.case_probability_flow_sampler("cpu")
