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
    b2, a2 = sampling_lib.get_samples(cfg, score, fl, fn, 3, 0, sd, 1.0, inv, tsd, scaler)
    from indm_amd.flow import flow_forward
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
