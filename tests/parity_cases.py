"""Parity cases shared by the CPU host-logic tests (HIP layer emulated by tests/_emu.py) and the
GPU parity tests (real kernels through the C ABI).  Each case replays the noise the reference drew,
runs the PRODUCT path (indm_amd/*) and compares with the golden outputs of the reference."""
import numpy as np
import torch

import golden_utils as gu
from indm_amd import flow as iflow
from indm_amd import likelihood, losses, ncsnpp, sampling, sde_lib, utils
from indm_amd.rng import ReplayRng

FIXTURES = ["mini_vp_nll", "mini_vp_fid", "mini_ve"]


def build(fx, device):
    d, cfg = gu.load(fx), gu.mini_config(fx)
    cfg.device = torch.device(device)
    score = ncsnpp.NCSNpp(cfg)
    score.load_reference_state_dict(gu.state_dict(d, "score"))
    fl = iflow.create_flow_model(cfg)
    fl.load_reference_state_dict(gu.state_dict(d, "flow"))
    return d, cfg, score.to(device), fl.to(device)


def case_state_dict_roundtrip(fx, device):
    d, cfg, score, fl = build(fx, device)
    for model, pre in ((score, "score"), (fl, "flow")):
        ref = gu.state_dict(d, pre)
        out = model.reference_state_dict(prefix="")
        assert set(out) == set(ref), (sorted(set(out) ^ set(ref))[:5])
        for k in ref:
            assert tuple(out[k].shape) == tuple(ref[k].shape), k
            assert torch.equal(out[k].cpu().to(ref[k].dtype), ref[k]), k
    assert [n for n, p in fl.named_parameters() if p.requires_grad] == list(d["flow_param_order"])
    assert [n for n, p in score.named_parameters() if p.requires_grad] == list(d["score_param_order"])


def case_score_network(fx, device, tol=1.0):
    d, cfg, score, _ = build(fx, device)
    sde = sde_lib.get_sde(cfg)
    x = torch.from_numpy(d["x"]).to(device).requires_grad_(True)
    t = torch.from_numpy(d["score/t"])
    names = [k[len("score/grad/"):] for k in d if k.startswith("score/grad/")]
    params = dict(score.named_parameters())
    s = losses.get_score_fn(cfg, sde, score, train=False, continuous=True)(x, t)
    gu.assert_close(s, d["score/out"], 1e-4 * tol, 1e-5 * tol, "score out")
    grads = torch.autograd.grad(s, [x] + [params[n] for n in names], torch.from_numpy(d["score/gy"]).to(device))
    gu.assert_close(grads[0], d["score/gx"], 1e-4 * tol, 1e-5 * tol, "score gx")
    for n, g in zip(names, grads[1:]):
        gu.assert_close(score.to_reference_layout(n, g), d[f"score/grad/{n}"], 2e-4 * tol, 1e-5 * tol, n)


def case_flow_forward_modes(fx, device, tol=1.0):
    d, cfg, _, fl = build(fx, device)
    x = torch.from_numpy(d["x"]).to(device)
    params = dict(fl.named_parameters())
    for mode in ("train", "eval"):
        fl.load_reference_state_dict(gu.state_dict(d, "flow"))
        fl.train() if mode == "train" else fl.eval()
        rng = ReplayRng(gu.events(d, f"flow_{mode}"), device)
        z, ld = iflow.flow_forward(cfg, fl, x.clone(), reverse=False, rng=rng)
        assert rng.done()
        gu.assert_close(z, d[f"flow_{mode}/z"], 2e-5 * tol, 2e-5 * tol, f"flow {mode} z")
        gu.assert_close(ld, d[f"flow_{mode}/logdet_kl"], 5e-5 * tol, 2e-4 * tol, f"flow {mode} logdet")
        if mode == "train":
            names = [k[len("flow_train/grad/"):] for k in d if k.startswith("flow_train/grad/")]
            gr = torch.autograd.grad([z, ld], [params[n] for n in names],
                                     [torch.from_numpy(d["flow_train/gz"]).to(device),
                                      torch.tensor([1.0, -0.5], device=device)])
            for n, g in zip(names, gr):
                gu.assert_close(fl.to_reference_layout(n, g), d[f"flow_train/grad/{n}"], 5e-4 * tol, 5e-5 * tol,
                                "grad " + n)
            sd = fl.reference_state_dict(prefix="")
            for k, v in gu.sub(d, "flow_train/buffers").items():
                gu.assert_close(sd[k].float(), v.float(), 1e-4 * tol, 1e-5 * tol, "buffer " + k)
    fl.load_reference_state_dict(gu.state_dict(d, "flow"))
    fl.eval()
    rng = ReplayRng(gu.events(d, "flow_nologdet"), device)
    with torch.no_grad():
        z, _ = iflow.flow_forward(cfg, fl, x.clone(), log_det=None, reverse=False, rng=rng)
    gu.assert_close(z, d["flow_nologdet/z"], 2e-5 * tol, 2e-5 * tol, "flow nologdet")
    rng = ReplayRng(gu.events(d, "flow_reverse"), device)
    with torch.no_grad():
        xr, _ = iflow.flow_forward(cfg, fl, torch.from_numpy(d["flow_eval/z"]).to(device), log_det=None,
                                   reverse=True, rng=rng)
    assert rng.done()
    gu.assert_close(xr, d["flow_reverse/x"], 2e-4 * tol, 2e-4 * tol, "flow reverse")


def case_nelbo(fx, device, tol=1.0):
    d, cfg, score, fl = build(fx, device)
    sde = sde_lib.get_sde(cfg)
    score.eval(); fl.eval()
    fn = likelihood.get_elbo_fn(cfg, sde, inverse_scaler=utils.get_data_inverse_scaler(cfg))
    rng = ReplayRng(gu.events(d, "nelbo"), device)
    b1, b2 = fn(score, fl, torch.from_numpy(d["x"]).to(device), rng=rng)
    assert rng.done()
    gu.assert_close(b1, d["nelbo/bpd"], 2e-4 * tol, 2e-4 * tol, "nelbo")
    gu.assert_close(b2, d["nelbo/bpd_residual"], 2e-4 * tol, 2e-4 * tol, "nelbo residual")


def case_pc_sampler(fx, device, tol=1.0):
    d, cfg, score, fl = build(fx, device)
    cfg.sampling.method = "pc"
    sde = sde_lib.get_sde(cfg)
    score.eval(); fl.eval()
    shape = tuple(d["x"].shape)
    fn = sampling.get_sampling_fn(cfg, sde, shape, utils.get_data_inverse_scaler(cfg), cfg.sampling.truncation_time,
                                  device=device)
    rng = ReplayRng(gu.events(d, "pc"), device)
    before, after, nfe = fn(score, fl, rng=rng)
    assert rng.done() and nfe == int(d["pc/nfe"])
    gu.assert_close(before, d["pc/before"], 2e-4 * tol, 2e-4 * tol, "pc before")
    gu.assert_close(after, d["pc/after"], 2e-3 * tol, 2e-3 * tol, "pc after")


def case_pf_ode(fx, device, tol=1.0):
    d, cfg, score, fl = build(fx, device)
    sde = sde_lib.get_sde(cfg)
    score.eval(); fl.eval()
    inv = utils.get_data_inverse_scaler(cfg)
    x = torch.from_numpy(d["x"]).to(device)
    from indm_amd import ops
    from indm_amd.sampling import _Stepper
    st = _Stepper(cfg, sde, score, None)
    eps = torch.from_numpy(d["ode_rhs/eps"]).to(device)
    drift, div = likelihood.get_div_fn(st, sde)(ops.to_nhwc(x), torch.from_numpy(d["ode_rhs/t"]), ops.to_nhwc(eps))
    gu.assert_close(ops.to_nchw(drift), d["ode_rhs/drift"], 2e-4 * tol, 2e-5 * tol, "pf-ode drift")
    gu.assert_close(div, d["ode_rhs/div"], 2e-4 * tol, 2e-3 * tol, "pf-ode div")
    fn = likelihood.get_likelihood_fn(cfg, sde, inv, rtol=1e-3, atol=1e-3)
    rng = ReplayRng(gu.events(d, "nll"), device)
    bpd, z, nfe = fn(score, fl, x, rng=rng)
    assert rng.done()
    assert abs(nfe - int(d["nll/nfe"])) <= 12, (nfe, int(d["nll/nfe"]))
    gu.assert_close(bpd, d["nll/bpd"], 1e-3 * tol, 3e-3 * tol, "nll bpd")
    cfg.sampling.method = "ode"
    cfg.eval.rtol, cfg.eval.atol = 1e-3, 1e-3
    fn = sampling.get_sampling_fn(cfg, sde, tuple(x.shape), inv, cfg.sampling.truncation_time, device=device)
    rng = ReplayRng(gu.events(d, "ode"), device)
    before, after, nfe = fn(score, fl, rng=rng)
    assert rng.done()
    gu.assert_close(before, d["ode/before"], 5e-3 * tol, 5e-3 * tol, "ode before")
    gu.assert_close(after, d["ode/after"], 5e-3 * tol, 5e-3 * tol, "ode after")


def case_two_training_steps(fx, device, tol=1.0):
    d, cfg, score, fl = build(fx, device)
    sde = sde_lib.get_sde(cfg)
    state, flow_state, flat_g = utils.build_states(cfg, score, fl, device)
    step_fn = losses.get_step_fn(cfg, sde, train=True)
    score.train(); fl.train()
    for it in range(2):
        rng = ReplayRng(gu.events(d, f"step{it}"), device)
        out = step_fn(state, flow_state, torch.from_numpy(d[f"step{it}/x"]).to(device), rng=rng)
        assert rng.done()
        for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out):
            assert not v.is_cuda
            gu.assert_close(v, d[f"step{it}/{nm}"], 5e-4 * tol, 2e-5 * tol, f"step{it} {nm}")
    # AdamW's update is ~lr*sign(g) in the first steps, so an element whose gradient is at rounding-noise level
    # may legitimately move the other way on a different reduction order: elementwise checks therefore allow a
    # vanishing fraction of elements to differ by up to 2 steps * lr, everything else must agree tightly.
    for pre, model, st, lr in (("score", score, state, cfg.optim.lr), ("flow", fl, flow_state, cfg.flow.lr)):
        ema = dict(zip([n for n, p in model.named_parameters() if p.requires_grad], st["ema"].shadow_params()))
        for n, p in model.named_parameters():
            ref = torch.from_numpy(d[f"after2/{pre}/stat/{n}"]).double()
            got = gu.stat(p)
            scale = (ref[1].item() * p.numel()) ** 0.5 + 1e-12           # ~ |p|_2 * sqrt(numel) >= |sum p|
            assert abs(got[0] - ref[0]).item() <= 1e-4 * tol * scale + 2e-6, ("param sum " + n, got, ref)
            assert abs(got[1] - ref[1]).item() <= 1e-4 * tol * ref[1].item() + 1e-9, ("param sumsq " + n, got, ref)
            if p.requires_grad:
                ref = torch.from_numpy(d[f"after2/{pre}/emastat/{n}"]).double()
                got = gu.stat(ema[n])
                assert abs(got[0] - ref[0]).item() <= 1e-4 * tol * scale + 2e-6, ("ema sum " + n, got, ref)
                assert abs(got[1] - ref[1]).item() <= 1e-4 * tol * ref[1].item() + 1e-9, ("ema sumsq " + n, got, ref)
        sd = model.reference_state_dict(prefix="")
        for k, v in gu.sub(d, f"after2/{pre}/full").items():
            diff = (sd[k].detach().cpu().double() - v.double()).abs()
            assert diff.max().item() <= 2.2 * lr + 1e-6, ("full max " + k, diff.max().item())
            frac = (diff > 2e-6 * tol + 1e-4 * tol * v.double().abs()).double().mean().item()
            assert frac <= 2e-3, ("full frac " + k, frac)
    sd = fl.reference_state_dict(prefix="")
    for k, v in gu.sub(d, "after2/flow/buffers").items():
        gu.assert_close(sd[k].float(), v.float(), 1e-4 * tol, 1e-5 * tol, k)
    assert [state["step"], flow_state["step"], state["ema"].num_updates] == list(d["after2/steps"])


ALL_CASES = [case_state_dict_roundtrip, case_score_network, case_flow_forward_modes, case_nelbo, case_pc_sampler,
             case_two_training_steps]


# ----------------------------------------------------------------------------- SURVEY section 8(f) rows (mini_extra.npz)
def _extra_ve(device):
    d = gu.load("mini_extra")
    _, cfg, score, fl = build("mini_ve", device)
    cfg.model.num_scales = 4
    cfg.sampling.method = "pc"
    score.eval(); fl.eval()
    return d, cfg, score, fl


def case_pc_sampler_search(device, tol=1.0):
    d, cfg, score, fl = _extra_ve(device)
    cfg.sampling.pc_denoise, cfg.sampling.need_sample = True, False
    sde = sde_lib.get_sde(cfg)
    inv = utils.get_data_inverse_scaler(cfg)
    shape = (2, 3, 32, 32)
    fn = sampling.get_sampling_fn(cfg, sde, shape, inv, cfg.sampling.truncation_time, device=device)
    assert fn.__name__ == "pc_sampler_search"
    rng = ReplayRng(gu.events(d, "search"), device)
    before, after, nfe = fn(score, fl, final_time=0.0, rng=rng)
    assert rng.done() and nfe == int(d["search/nfe"])
    gu.assert_close(before, d["search/before"], 2e-4 * tol, 2e-4 * tol, "search before")
    gu.assert_close(after, d["search/after"], 2e-3 * tol, 2e-3 * tol, "search after")
    bd = torch.from_numpy(d["search_bd/before_data"])
    rng = ReplayRng(gu.events(d, "search_bd"), device)
    before, after, _ = fn(score, fl, temperature=0.9, final_time=5e-6, before_data=bd, rng=rng)
    assert rng.done()
    gu.assert_close(before, d["search_bd/before"], 2e-4 * tol, 2e-4 * tol, "search(before_data) before")
    gu.assert_close(after, d["search_bd/after"], 2e-3 * tol, 2e-3 * tol, "search(before_data) after")
    cfg.sampling.need_sample = True
    fn = sampling.get_sampling_fn(cfg, sde, shape, inv, cfg.sampling.truncation_time, device=device)
    rng = ReplayRng(gu.events(d, "search_ns"), device)
    before, after, _ = fn(score, fl, before_data=bd, rng=rng)
    assert rng.done()
    gu.assert_close(before, d["search_ns/before"], 1e-5, 1e-5, "search(need_sample) before")
    gu.assert_close(after, d["search_ns/after"], 2e-3 * tol, 2e-3 * tol, "search(need_sample) after")


def case_pc_sampler_more_step(device, tol=1.0):
    import pytest
    d, cfg, score, fl = _extra_ve(device)
    cfg.sampling.more_step, cfg.sampling.need_sample = True, False
    sde = sde_lib.get_sde(cfg)
    fn = sampling.get_sampling_fn(cfg, sde, (1, 3, 32, 32), utils.get_data_inverse_scaler(cfg),
                                  cfg.sampling.truncation_time, device=device)
    assert fn.__name__ == "pc_sampler_more_step"
    rng = ReplayRng(gu.events(d, "more_step"), device)
    before, after, _ = fn(score, fl, before_data=torch.from_numpy(d["search_bd/before_data"])[:1], rng=rng)
    assert rng.done()
    gu.assert_close(before, d["more_step/before"], 5e-4 * tol, 5e-4 * tol, "more_step before")   # 100 chained steps
    gu.assert_close(after, d["more_step/after"], 3e-3 * tol, 3e-3 * tol, "more_step after")
    with pytest.raises(TypeError):          # what the reference itself raises for before_data=None
        fn(score, fl)


def case_get_bpd(device, tol=1.0):
    import logging
    import test_oracle_golden as tog
    from indm_amd import evaluation
    d = gu.load("mini_extra")
    _, cfg, score, fl = build("mini_vp_nll", device)
    ref = tog.bpd_config()
    for k in ("num_test_data", "num_nelbo", "batch_size", "truncation_time"):
        cfg.eval[k] = ref.eval[k]
    sde = sde_lib.get_sde(cfg)
    inv, scaler = utils.get_data_inverse_scaler(cfg), utils.get_data_scaler(cfg)
    nelbo_fn = likelihood.get_elbo_fn(cfg, sde, inverse_scaler=inv)
    nll_fn = likelihood.get_likelihood_fn(cfg, sde, inv, rtol=1e-3, atol=1e-3)
    imgs = d["bpd/images_nhwc"]
    eval_ds = [{"image": imgs[i:i + 2]} for i in (0, 2)]       # TFDS-style NHWC batches
    score.eval(); fl.train()
    rng = ReplayRng(gu.events(d, "bpd"), device)
    res = evaluation.get_bpd(cfg, eval_ds, scaler, nelbo_fn, nll_fn, score, fl, step=7, eval=True, rng=rng)
    assert rng.done() and fl.training == bool(int(d["bpd/flow_training_after"]))
    logged = tog.parse_bpd_log(d)
    assert np.allclose(res["nelbo_bpd_per_eval"], logged["nelbo"], rtol=3e-4 * tol), (res, logged)
    assert np.allclose(res["nelbo_residual_bpd_per_eval"], logged["nelbo_residual"], rtol=3e-4 * tol)
    assert np.allclose([res["nelbo_bpd"], res["nelbo_residual_bpd"]], logged["avg"], rtol=3e-4 * tol)
    assert np.isclose(res["nll_wrong_bpd"], logged["nll_wrong"][-1], rtol=1e-3 * tol), (res, logged)
    assert np.isclose(res["nll_correct_bpd"], logged["nll_correct"][-1], rtol=1e-3 * tol), (res, logged)
    assert "nll_correct_train_eps_bpd" not in res


def case_deeper_score_net(device, tol=1.0):
    d = gu.load("mini_extra")
    cfg = gu.mini_config("mini_vp_nll")
    cfg.model.num_res_blocks = 2
    cfg.device = torch.device(device)
    score = ncsnpp.NCSNpp(cfg)
    score.load_reference_state_dict(gu.sub(d, "nrb2"))
    score = score.to(device).eval()
    out = score(torch.from_numpy(d["nrb2_io/x"]).to(device), (torch.from_numpy(d["nrb2_io/t"]) * 999).to(device))
    gu.assert_close(out, d["nrb2_io/raw_out"], 2e-4 * tol, 2e-5 * tol, "nrb=2 score net")


def case_celeba_squeeze(device, tol=1.0):
    """BASELINE config #4 flavour: vp/CELEBA/indm_fid (64x64, flow.squeeze=True, two-phase fid step)."""
    fx = "mini_celeba"
    d, cfg, score, fl = build(fx, device)
    sde = sde_lib.get_sde(cfg)
    x = torch.from_numpy(d["x"]).to(device)
    score.eval()
    out = score(x, (torch.from_numpy(d["score/t"]) * 999).to(device))
    gu.assert_close(out, d["score/raw_out"], 2e-4 * tol, 2e-5 * tol, "celeba score net")
    for mode in ("train", "eval"):
        fl.load_reference_state_dict(gu.state_dict(d, "flow"))
        fl.train() if mode == "train" else fl.eval()
        rng = ReplayRng(gu.events(d, f"flow_{mode}"), device)
        with torch.no_grad():
            z, ld = iflow.flow_forward(cfg, fl, x.clone(), reverse=False, rng=rng)
        assert rng.done()
        gu.assert_close(z, d[f"flow_{mode}/z"], 2e-5 * tol, 2e-5 * tol, f"celeba flow {mode} z")
        gu.assert_close(ld, d[f"flow_{mode}/logdet_kl"], 5e-5 * tol, 2e-4 * tol, f"celeba flow {mode} logdet")
    fl.load_reference_state_dict(gu.state_dict(d, "flow"))
    fl.eval()
    rng = ReplayRng(gu.events(d, "flow_reverse"), device)
    with torch.no_grad():
        xr, _ = iflow.flow_forward(cfg, fl, torch.from_numpy(d["flow_eval/z"]).to(device), log_det=None,
                                   reverse=True, rng=rng)
    assert rng.done()
    gu.assert_close(xr, d["flow_reverse/x"], 2e-4 * tol, 2e-4 * tol, "celeba flow reverse")
    fn = likelihood.get_elbo_fn(cfg, sde, inverse_scaler=utils.get_data_inverse_scaler(cfg))
    rng = ReplayRng(gu.events(d, "nelbo"), device)
    b1, b2 = fn(score, fl, x, rng=rng)
    assert rng.done()
    gu.assert_close(b1, d["nelbo/bpd"], 2e-4 * tol, 2e-4 * tol, "celeba nelbo")
    gu.assert_close(b2, d["nelbo/bpd_residual"], 2e-4 * tol, 2e-4 * tol, "celeba nelbo residual")
    state, flow_state, _ = utils.build_states(cfg, score, fl, device)
    step_fn = losses.get_step_fn(cfg, sde, train=True)
    score.train(); fl.train()
    rng = ReplayRng(gu.events(d, "step0"), device)
    outs = step_fn(state, flow_state, torch.from_numpy(d["step0/x"]).to(device), rng=rng)
    assert rng.done()
    for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), outs):
        gu.assert_close(v, d[f"step0/{nm}"], 5e-4 * tol, 2e-5 * tol, f"celeba step0 {nm}")
    for pre, model in (("score", score), ("flow", fl)):
        for n, p in model.named_parameters():
            ref = torch.from_numpy(d[f"after1/{pre}/stat/{n}"]).double()
            got = gu.stat(p)
            scale = (ref[1].item() * p.numel()) ** 0.5 + 1e-12
            assert abs(got[0] - ref[0]).item() <= 1e-4 * tol * scale + 2e-6, ("param sum " + n, got, ref)
            assert abs(got[1] - ref[1]).item() <= 1e-4 * tol * ref[1].item() + 1e-9, ("param sumsq " + n, got, ref)
    assert [state["step"], flow_state["step"], state["ema"].num_updates] == list(d["after1/steps"])


EXTRA_CASES = [case_pc_sampler_search, case_pc_sampler_more_step, case_get_bpd, case_deeper_score_net,
               case_celeba_squeeze]


# ----------------------------------------------------------------------------- round-2 fixtures (mini_r2.npz)
def _check_param_stats(d, prefix, pairs, tol, slack=0.0):
    """slack: absolute allowance on a tensor's sum -- AdamW's first update is lr*g/(|g|+eps), so an element whose
    gradient sits at rounding-noise level may move by a few per cent of lr differently (cf. case_two_training_steps)."""
    for pre, model in pairs:
        for n, p in model.named_parameters():
            ref = torch.from_numpy(d[f"{prefix}/{pre}/stat/{n}"]).double()
            got = gu.stat(p)
            scale = (ref[1].item() * p.numel()) ** 0.5 + 1e-12
            assert abs(got[0] - ref[0]).item() <= 1e-4 * tol * scale + 2e-6 + slack, ("param sum " + n, got, ref)
            assert abs(got[1] - ref[1]).item() <= 1e-4 * tol * ref[1].item() + 1e-9 + 2 * slack * p.detach().abs().max().item(), \
                ("param sumsq " + n, got, ref)


def case_micro_batch_step(tag, device, tol=1.0):
    """optim.num_micro_batch = 2 against the reference's own micro-batched step (losses.py:280-309 / :355-399):
    the 4 loss vectors in batch order and every parameter after the fused optimiser (gradients are the SUM of the
    per-micro-batch means)."""
    d = gu.load("mini_r2")
    fx = {"mb2_nll": "mini_vp_nll", "mb2_fid": "mini_vp_fid"}[tag]
    _, cfg, score, fl = build(fx, device)
    cfg.optim.num_micro_batch = 2
    sde = sde_lib.get_sde(cfg)
    state, flow_state, _ = utils.build_states(cfg, score, fl, device)
    step_fn = losses.get_step_fn(cfg, sde, train=True)
    score.train(); fl.train()
    rng = ReplayRng(gu.events(d, tag), device)
    out = step_fn(state, flow_state, torch.from_numpy(d[f"{tag}/x"]).to(device), rng=rng)
    assert rng.done()
    for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out):
        gu.assert_close(v, d[f"{tag}/{nm}"], 5e-4 * tol, 2e-5 * tol, f"{tag} {nm}")
    _check_param_stats(d, f"{tag}/after", (("score", score), ("flow", fl)), tol, slack=0.02 * cfg.flow.lr)
    assert [state["step"], flow_state["step"], state["ema"].num_updates] == list(d[f"{tag}/steps"])


def case_ve_ode(device, tol=1.0):
    """VE probability-flow ODE: RK45 likelihood (likelihood.py:41-140) and ODE sampler (sampling.py:547-621)."""
    d = gu.load("mini_r2")
    dv, cfg, score, fl = build("mini_ve", device)
    sde = sde_lib.get_sde(cfg)
    score.eval(); fl.eval()
    inv = utils.get_data_inverse_scaler(cfg)
    x = torch.from_numpy(dv["x"]).to(device)
    fn = likelihood.get_likelihood_fn(cfg, sde, inv, rtol=1e-3, atol=1e-3)
    rng = ReplayRng(gu.events(d, "ve_nll"), device)
    bpd, z, nfe = fn(score, fl, x, rng=rng)
    assert rng.done()
    assert abs(nfe - int(d["ve_nll/nfe"])) <= 12, (nfe, int(d["ve_nll/nfe"]))
    gu.assert_close(bpd, d["ve_nll/bpd"], 1e-3 * tol, 3e-3 * tol, "ve nll bpd")
    cfg.sampling.method = "ode"
    cfg.eval.rtol, cfg.eval.atol = 1e-3, 1e-3
    fn = sampling.get_sampling_fn(cfg, sde, tuple(x.shape), inv, cfg.sampling.truncation_time, device=device)
    rng = ReplayRng(gu.events(d, "ve_ode"), device)
    before, after, nfe = fn(score, fl, rng=rng)
    assert rng.done()
    gu.assert_close(before, d["ve_ode/before"], 5e-3 * tol, 5e-3 * tol, "ve ode before")
    gu.assert_close(after, d["ve_ode/after"], 5e-3 * tol, 5e-3 * tol, "ve ode after")


def case_reference_checkpoint(device, tol=1.0):
    """Restore checkpoints WRITTEN BY THE REFERENCE's utils.save_checkpoint (utils.py:36-48; tests/golden/
    mini_r2_ckpt_*.pth, taken after one optimisation step) into freshly initialised product networks and reproduce the
    reference's next step: losses, parameters (=> AdamW moments + step count restored) and EMA (=> shadow + num_updates)."""
    import os
    d = gu.load("mini_r2")
    cfg = gu.mini_config("mini_vp_nll")
    cfg.device = torch.device(device)
    torch.manual_seed(123)                       # weights the checkpoint must overwrite
    score = ncsnpp.NCSNpp(cfg).to(device)
    fl = iflow.create_flow_model(cfg).to(device)
    state, flow_state, _ = utils.build_states(cfg, score, fl, device)
    utils.restore_checkpoint(cfg, os.path.join(gu.GOLDEN, "mini_r2_ckpt_score.pth"), state, device)
    utils.restore_checkpoint(cfg, os.path.join(gu.GOLDEN, "mini_r2_ckpt_flow.pth"), flow_state, device)
    assert state["step"] == 1 and flow_state["step"] == 1 and state["ema"].num_updates == 1
    assert state["optimizer"].t == 1 and flow_state["optimizer"].t == 1
    sde = sde_lib.get_sde(cfg)
    step_fn = losses.get_step_fn(cfg, sde, train=True)
    score.train(); fl.train()
    rng = ReplayRng(gu.events(d, "ckpt"), device)
    out = step_fn(state, flow_state, torch.from_numpy(d["ckpt/x"]).to(device), rng=rng)
    assert rng.done()
    for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out):
        gu.assert_close(v, d[f"ckpt/{nm}"], 5e-4 * tol, 2e-5 * tol, f"ckpt step {nm}")
    _check_param_stats(d, "ckpt/after", (("score", score), ("flow", fl)), tol, slack=0.02 * cfg.flow.lr)
    for pre, model, st in (("score", score, state), ("flow", fl, flow_state)):
        ema = dict(zip([n for n, p in model.named_parameters() if p.requires_grad], st["ema"].shadow_params()))
        for n, s_ in ema.items():
            ref = torch.from_numpy(d[f"ckpt/after/{pre}/emastat/{n}"]).double()
            got = gu.stat(s_)
            scale = (ref[1].item() * s_.numel()) ** 0.5 + 1e-12
            assert abs(got[0] - ref[0]).item() <= 1e-4 * tol * scale + 2e-6, ("ema sum " + n, got, ref)
            assert abs(got[1] - ref[1]).item() <= 1e-4 * tol * ref[1].item() + 1e-9, ("ema sumsq " + n, got, ref)
    assert [state["step"], flow_state["step"], state["ema"].num_updates] == list(d["ckpt/steps"])


R2_CASES = [case_ve_ode, case_reference_checkpoint]


# ----------------------------------------------------------------------------- round-3 fixtures (mini_r3.npz)
R3_TAGS = {"vp_recon": ("mini_vp_nll", "scoreflow", False), "vp_recon_ddpm": ("mini_vp_nll", "ddpm", False),
           "vp_recon_st": ("mini_vp_nll", "scoreflow", True), "vp_fid_recon": ("mini_vp_fid", "scoreflow", False),
           "ve_recon": ("mini_ve", "scoreflow", False), "ve_recon_st": ("mini_ve", "ddpm", True)}


def case_reconstruction_loss(tag, device, tol=1.0):
    """get_sde_loss_fn.loss_fn with recon_loss=True (losses.py:120-140) and st (sde_lib.py:208-215): per-image losses
    and the gradient of the mean loss against the reference's goldens"""
    d = gu.load("mini_r3")
    fx, variance, st = R3_TAGS[tag]
    _, cfg, score, _ = build(fx, device)
    sde = sde_lib.get_sde(cfg)
    loss_fn = losses.get_sde_loss_fn(cfg, sde, train=True, variance=variance)
    score.train()
    np.random.seed(int(d[f"{tag}/numpy_seed"]))      # get_t_min(st=True) draws from numpy's global generator
    rng = ReplayRng(gu.events(d, tag), device)
    loss = loss_fn(score, torch.from_numpy(d[f"{tag}/x"]).to(device), st=st, recon_loss=True, rng=rng)
    assert rng.done()
    gu.assert_close(loss, d[f"{tag}/loss"], 2e-4 * tol, 1e-5 * tol, f"{tag} loss")
    for p in score.parameters():
        p.grad = None
    loss.mean().backward()
    checked = 0
    for n, p in score.named_parameters():
        key = f"{tag}/grad/stat/{n}"
        if key not in d or p.grad is None:
            continue
        got, ref = gu.stat(score.to_reference_layout(n, p.grad)), torch.from_numpy(d[key])
        scale = (ref[1].item() * p.numel()) ** 0.5 + 1e-12
        assert abs(got[0] - ref[0]).item() <= 5e-4 * tol * scale + 1e-6, ("grad sum " + n, got, ref)
        assert abs(got[1] - ref[1]).item() <= 1e-3 * tol * ref[1].item() + 1e-9, ("grad sumsq " + n, got, ref)
        checked += 1
    assert checked > 20


def case_training_st_step(tag, device, tol=1.0):
    """training.st = True through flow_step_fn_nll (losses.py:294) and the two-phase flow_step_fn_fid, whose second phase
    keeps and re-scales the first phase's score gradients (losses.py:378-395); one and two micro-batches"""
    d = gu.load("mini_r3")
    fx = "mini_vp_nll" if tag == "st_nll" else "mini_vp_fid"
    _, cfg, score, fl = build(fx, device)
    cfg.training.st = True
    cfg.optim.num_micro_batch = 2 if tag.endswith("mb2") else 1
    sde = sde_lib.get_sde(cfg)
    state, flow_state, _ = utils.build_states(cfg, score, fl, device)
    step_fn = losses.get_step_fn(cfg, sde, train=True)
    score.train(); fl.train()
    rng = ReplayRng(gu.events(d, tag), device)
    out = step_fn(state, flow_state, torch.from_numpy(d[f"{tag}/x"]).to(device), rng=rng)
    assert rng.done()
    for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out):
        gu.assert_close(v, d[f"{tag}/{nm}"], 5e-4 * tol, 2e-5 * tol, f"{tag} {nm}")
    _check_param_stats(d, f"{tag}/after", (("score", score), ("flow", fl)), tol, slack=0.02 * cfg.flow.lr)
    assert [state["step"], flow_state["step"], state["ema"].num_updates] == list(d[f"{tag}/steps"])


def case_probability_flow_sampler(device, tol=1.0):
    """mini_r4.npz: sampling.probability_flow = True -- reverse diffusion + Langevin on the VE net against the reference's
    samples (eager launches and the hipGraph-able predictor-only form are covered by the other sampler cases), and the
    Euler-Maruyama predictor, for which the reference raises TypeError (so must the mirror, eager and graphed)."""
    import pytest
    d = gu.load("mini_r4")
    _, cfg, score, fl = build("mini_ve", device)
    cfg.sampling.method, cfg.sampling.probability_flow, cfg.sampling.num_scales = "pc", True, 4
    sde = sde_lib.get_sde(cfg)
    score.eval(); fl.eval()
    shape = tuple(d["pf_rd/before"].shape)
    fn = sampling.get_sampling_fn(cfg, sde, shape, utils.get_data_inverse_scaler(cfg), cfg.sampling.truncation_time,
                                  device=device)
    rng = ReplayRng(gu.events(d, "pf_rd"), device)
    before, after, nfe = fn(score, fl, rng=rng)
    assert rng.done() and nfe == int(d["pf_rd/nfe"])
    gu.assert_close(before, d["pf_rd/before"], 2e-4 * tol, 2e-4 * tol, "pf rd before")
    gu.assert_close(after, d["pf_rd/after"], 2e-3 * tol, 2e-3 * tol, "pf rd after")
    _, cfg, score, fl = build("mini_vp_nll", device)
    cfg.sampling.method, cfg.sampling.probability_flow, cfg.sampling.num_scales = "pc", True, 4
    sde = sde_lib.get_sde(cfg)
    score.eval(); fl.eval()
    fn = sampling.get_sampling_fn(cfg, sde, (2, 3, 32, 32), utils.get_data_inverse_scaler(cfg),
                                  cfg.sampling.truncation_time, device=device)
    from indm_amd.rng import DeviceRng
    with pytest.raises(TypeError):
        fn(score, fl, rng=DeviceRng(device, seed=3))
