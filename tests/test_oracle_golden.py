"""Pin the CPU oracle (oracle/indm_oracle.py) to golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only; this is what makes every later HIP-vs-oracle comparison
a comparison with the reference."""
import numpy as np
import pytest
import torch

import golden_utils as gu
import indm_oracle as O

FIXTURES = ["mini_vp_nll", "mini_vp_fid", "mini_ve"]


def bit_equal(a, b):
    """bitwise equality with NaN == NaN (discretize() legitimately yields NaN for G^2 > 1)."""
    return a.shape == b.shape and torch.allclose(a, b, rtol=0, atol=0, equal_nan=True)


def test_sde_known_answers():
    d = gu.load("sde_known_answers")
    t, u, x = (torch.from_numpy(d[k]) for k in ("t", "u", "x"))
    for name, sde in (("vp", O.VPSDE()), ("ve", O.VESDE())):
        mean, std = sde.marginal_prob(x, t)
        drift, diff = sde.sde(x, t)
        for k, v in (("mean", mean), ("std", std), ("drift", drift), ("diffusion", diff),
                     ("prior_logp", sde.prior_logp(x)), ("antiderivative", sde.antiderivative(t)),
                     ("antiderivative_T", sde.antiderivative(1.0)), ("Z", sde.normalizing_constant(1e-5))):
            assert bit_equal(v, torch.from_numpy(d[f"{name}/{k}"])), (name, k)
        f, G = sde.discretize(x, t, None)
        assert bit_equal(f, torch.from_numpy(d[f"{name}/disc_f"])) and bit_equal(G, torch.from_numpy(d[f"{name}/disc_G"]))
        f, G = sde.discretize(x, t, t * 0.5)
        assert bit_equal(f, torch.from_numpy(d[f"{name}/disc2_f"])) and bit_equal(G, torch.from_numpy(d[f"{name}/disc2_G"]))
        tt, _ = sde.get_diffusion_time(7, 1e-5, True, O.ReplayRng([("rand", d["u"])]))
        assert bit_equal(tt, torch.from_numpy(d[f"{name}/is_t"])), name
    assert torch.equal(O.get_timestep_embedding(t * 999, 128), torch.from_numpy(d["timestep_embedding"]))
    # SURVEY Appendix E anchors
    vp = O.VPSDE()
    assert abs(float(vp.normalizing_constant(1e-5)) - 23.852275848388672) < 1e-6
    assert abs(float(vp.marginal_prob(x[:1], torch.tensor([1e-5]))[1]) - 0.00100661756) < 1e-9


def test_native_op_oracles():
    d = gu.load("native_ops")
    for i, (c, h, w, up, down, p0, p1) in enumerate(d["cases"]):
        for kn in ("fir", "rand"):
            g = gu.sub(d, f"ufd{i}_{kn}")
            x = g["x"].clone().requires_grad_(True)
            y = O.upfirdn2d_native(x, g["k"], int(up), int(up), int(down), int(down), int(p0), int(p1), int(p0), int(p1))
            gu.assert_close(y, g["y"], 1e-6, 1e-6, f"ufd{i}")
            (gx,) = torch.autograd.grad(y, x, g["gy"])
            gu.assert_close(gx, g["gx"], 1e-6, 1e-6, f"ufd{i} grad")
    g = gu.sub(d, "flr")
    gu.assert_close(O.fused_leaky_relu(g["x"], g["b"]), g["y"], 1e-6, 1e-6, "fused_leaky_relu")
    gu.assert_close(O.fused_bias_act(g["x"], g["b"], None, 3, 0, 0.2, 2 ** 0.5), g["y"], 1e-6, 1e-6, "fused_bias_act")


@pytest.mark.parametrize("fx", FIXTURES)
def test_score_network(fx):
    d, cfg = gu.load(fx), gu.mini_config(fx)
    P = gu.state_dict(d, "score")
    sde = O.get_sde(cfg)
    x = torch.from_numpy(d["x"]).clone().requires_grad_(True)
    t = torch.from_numpy(d["score/t"])
    names = [k[len("score/grad/"):] for k in d if k.startswith("score/grad/")]
    for n in names:
        P[n].requires_grad_(True)
    s = O.get_score_fn(P, cfg, sde, train=False)(x, t)
    gu.assert_close(s, d["score/out"], 1e-4, 1e-5, "score out")
    grads = torch.autograd.grad(s, [x] + [P[n] for n in names], torch.from_numpy(d["score/gy"]))
    gu.assert_close(grads[0], d["score/gx"], 1e-4, 1e-5, "score gx")
    for n, gr in zip(names, grads[1:]):
        gu.assert_close(gr, d[f"score/grad/{n}"], 1e-4, 1e-5, n)


@pytest.mark.parametrize("fx", FIXTURES)
def test_flow_forward_modes(fx):
    d, cfg = gu.load(fx), gu.mini_config(fx)
    x = torch.from_numpy(d["x"])
    for mode in ("train", "eval"):
        P = gu.state_dict(d, "flow")
        names = [k[len("flow_train/grad/"):] for k in d if k.startswith("flow_train/grad/")] if mode == "train" else []
        for n in names:
            P[n].requires_grad_(True)
        rng = O.ReplayRng(gu.events(d, f"flow_{mode}"))
        bn = {}
        z, ld = O.flow_forward(P, cfg, x.clone(), training=(mode == "train"), rng=rng, bn_updates=bn)
        assert rng.done()
        gu.assert_close(z, d[f"flow_{mode}/z"], 1e-5, 1e-5, f"flow {mode} z")
        gu.assert_close(ld, d[f"flow_{mode}/logdet_kl"], 2e-5, 1e-4, f"flow {mode} logdet")
        if mode == "train":
            gr = torch.autograd.grad([z, ld], [P[n] for n in names],
                                     [torch.from_numpy(d["flow_train/gz"]), torch.tensor([1.0, -0.5])])
            for n, g_ in zip(names, gr):
                gu.assert_close(g_, d[f"flow_train/grad/{n}"], 2e-4, 2e-5, "grad " + n)
            for k, v in bn.items():
                gu.assert_close(v, d[f"flow_train/buffers/{k}"], 1e-5, 1e-6, k)
    P = gu.state_dict(d, "flow")
    rng = O.ReplayRng(gu.events(d, "flow_nologdet"))
    with torch.no_grad():
        z, _ = O.flow_forward(P, cfg, x.clone(), log_det=None, training=False, rng=rng)
    gu.assert_close(z, d["flow_nologdet/z"], 1e-5, 1e-5, "flow nologdet")
    rng = O.ReplayRng(gu.events(d, "flow_reverse"))
    with torch.no_grad():
        xr, _ = O.flow_forward(P, cfg, torch.from_numpy(d["flow_eval/z"]), log_det=None, reverse=True, rng=rng)
    assert rng.done()
    gu.assert_close(xr, d["flow_reverse/x"], 1e-4, 1e-4, "flow reverse")


@pytest.mark.parametrize("fx", FIXTURES)
def test_nelbo(fx):
    d, cfg = gu.load(fx), gu.mini_config(fx)
    Ps, Pf = gu.state_dict(d, "score"), gu.state_dict(d, "flow")
    rng = O.ReplayRng(gu.events(d, "nelbo"))
    b1, b2 = O.nelbo_fn(Ps, Pf, cfg, O.get_sde(cfg), torch.from_numpy(d["x"]), rng)
    assert rng.done()
    gu.assert_close(b1, d["nelbo/bpd"], 1e-4, 1e-4, "nelbo")
    gu.assert_close(b2, d["nelbo/bpd_residual"], 1e-4, 1e-4, "nelbo residual")


@pytest.mark.parametrize("fx", FIXTURES)
def test_pc_sampler(fx):
    d, cfg = gu.load(fx), gu.mini_config(fx)
    cfg.sampling.method = "pc"
    Ps, Pf = gu.state_dict(d, "score"), gu.state_dict(d, "flow")
    rng = O.ReplayRng(gu.events(d, "pc"))
    shape = tuple(d["x"].shape)
    before, after, nfe = O.pc_sampler(Ps, Pf, cfg, O.get_sde(cfg), shape, rng, cfg.sampling.truncation_time)
    assert rng.done() and nfe == int(d["pc/nfe"])
    gu.assert_close(before, d["pc/before"], 1e-4, 1e-4, "pc before")
    gu.assert_close(after, d["pc/after"], 1e-3, 1e-3, "pc after")


@pytest.mark.parametrize("fx", ["mini_vp_nll"])
def test_pf_ode_likelihood_and_sampler(fx):
    d, cfg = gu.load(fx), gu.mini_config(fx)
    Ps, Pf = gu.state_dict(d, "score"), gu.state_dict(d, "flow")
    sde = O.get_sde(cfg)
    x = torch.from_numpy(d["x"])
    drift, div = O.pf_ode_drift_and_div(Ps, cfg, sde, x, torch.from_numpy(d["ode_rhs/t"]),
                                        torch.from_numpy(d["ode_rhs/eps"]))
    gu.assert_close(drift, d["ode_rhs/drift"], 1e-4, 1e-5, "pf-ode drift")
    gu.assert_close(div, d["ode_rhs/div"], 1e-4, 1e-3, "pf-ode div")
    rng = O.ReplayRng(gu.events(d, "nll"))
    bpd, z, nfe = O.likelihood_fn(Ps, Pf, cfg, sde, x, rng, rtol=1e-3, atol=1e-3)
    assert rng.done() and nfe == int(d["nll/nfe"])
    gu.assert_close(bpd, d["nll/bpd"], 1e-4, 1e-3, "nll bpd")
    gu.assert_close(z, d["nll/z"], 1e-3, 1e-3, "nll z")
    rng = O.ReplayRng(gu.events(d, "ode"))
    before, after, nfe = O.ode_sampler(Ps, Pf, cfg, sde, tuple(x.shape), rng, cfg.sampling.truncation_time, 1e-3, 1e-3)
    assert rng.done() and nfe == int(d["ode/nfe"])
    gu.assert_close(before, d["ode/before"], 1e-3, 1e-3, "ode before")
    gu.assert_close(after, d["ode/after"], 1e-3, 1e-3, "ode after")


@pytest.mark.parametrize("fx", FIXTURES)
def test_two_training_steps(fx):
    d, cfg = gu.load(fx), gu.mini_config(fx)
    sde = O.get_sde(cfg)
    state = O.OracleState(gu.state_dict(d, "score"), cfg.optim.lr, cfg.model.ema_rate, cfg)
    fstate = O.OracleState(gu.state_dict(d, "flow"), cfg.flow.lr, cfg.flow.ema_rate, cfg)
    assert state.trainable == list(d["score_param_order"]) and fstate.trainable == list(d["flow_param_order"])
    step = O.flow_step_fn_nll if cfg.training.likelihood_weighting else O.flow_step_fn_fid
    for it in range(2):
        rng = O.ReplayRng(gu.events(d, f"step{it}"))
        out = step(cfg, sde, state, fstate, torch.from_numpy(d[f"step{it}/x"]), rng)
        assert rng.done()
        for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out):
            gu.assert_close(v, d[f"step{it}/{nm}"], 2e-4, 1e-5, f"step{it} {nm}")
        if it == 0:
            for pre, st in (("score", state), ("flow", fstate)):
                for k in st.trainable:
                    key = f"step0/gradstat/{pre}/{k}"
                    if key in d:
                        gu.assert_close(gu.stat(st.P[k].grad), d[key], 1e-3, 1e-7, "gradstat " + k)
                    else:
                        assert st.P[k].grad is None, k
    for pre, st in (("score", state), ("flow", fstate)):
        for k in st.trainable:
            gu.assert_close(gu.stat(st.P[k]), d[f"after2/{pre}/stat/{k}"], 1e-5, 1e-6, "param " + k)
        for k, s_ in zip(st.trainable, st.ema):
            gu.assert_close(gu.stat(s_), d[f"after2/{pre}/emastat/{k}"], 1e-5, 1e-6, "ema " + k)
        for k, v in gu.sub(d, f"after2/{pre}/full").items():
            gu.assert_close(st.P[k], v, 1e-4, 2e-6, "full " + k)
    for k, v in gu.sub(d, "after2/flow/buffers").items():
        gu.assert_close(fstate.P[k].float(), v.float(), 1e-5, 1e-6, k)
    assert [state.step, fstate.step, state.num_updates] == list(d["after2/steps"])


# ----------------------------------------------------------------------------- SURVEY section 8(f) rows (mini_extra.npz)
def _extra_ve():
    d, w = gu.load("mini_extra"), gu.load("mini_ve")
    cfg = gu.mini_config("mini_ve")
    cfg.model.num_scales = 4
    cfg.sampling.method = "pc"
    return d, cfg, gu.state_dict(w, "score"), gu.state_dict(w, "flow")


def test_pc_sampler_search_and_denoise():
    d, cfg, Ps, Pf = _extra_ve()
    cfg.sampling.pc_denoise, cfg.sampling.need_sample = True, False
    sde = O.get_sde(cfg)
    shape = (2, 3, 32, 32)
    rng = O.ReplayRng(gu.events(d, "search"))
    before, after, nfe = O.pc_sampler_search(Ps, Pf, cfg, sde, shape, rng, cfg.sampling.truncation_time)
    assert rng.done() and nfe == int(d["search/nfe"])
    gu.assert_close(before, d["search/before"], 1e-4, 1e-4, "search before")
    gu.assert_close(after, d["search/after"], 1e-3, 1e-3, "search after")
    rng = O.ReplayRng(gu.events(d, "search_bd"))
    bd = torch.from_numpy(d["search_bd/before_data"])
    before, after, _ = O.pc_sampler_search(Ps, Pf, cfg, sde, shape, rng, cfg.sampling.truncation_time,
                                           temperature=0.9, final_time=5e-6, before_data=bd)
    assert rng.done()
    gu.assert_close(before, d["search_bd/before"], 1e-4, 1e-4, "search(before_data) before")
    gu.assert_close(after, d["search_bd/after"], 1e-3, 1e-3, "search(before_data) after")
    cfg.sampling.need_sample = True
    rng = O.ReplayRng(gu.events(d, "search_ns"))
    before, after, _ = O.pc_sampler_search(Ps, Pf, cfg, sde, shape, rng, cfg.sampling.truncation_time, before_data=bd)
    assert rng.done()
    gu.assert_close(before, d["search_ns/before"], 1e-5, 1e-5, "search(need_sample) before")
    gu.assert_close(after, d["search_ns/after"], 1e-3, 1e-3, "search(need_sample) after")


def test_pc_sampler_more_step():
    d, cfg, Ps, Pf = _extra_ve()
    cfg.sampling.more_step, cfg.sampling.need_sample = True, False
    sde = O.get_sde(cfg)
    rng = O.ReplayRng(gu.events(d, "more_step"))
    bd = torch.from_numpy(d["search_bd/before_data"])[:1]
    before, after, _ = O.pc_sampler_more_step(Ps, Pf, cfg, sde, (1, 3, 32, 32), rng, cfg.sampling.truncation_time,
                                              before_data=bd)
    assert rng.done()
    gu.assert_close(before, d["more_step/before"], 1e-4, 1e-4, "more_step before")
    gu.assert_close(after, d["more_step/after"], 1e-3, 1e-3, "more_step after")
    assert str(d["more_step/none_raises"]) == "TypeError"
    with pytest.raises(TypeError):
        O.pc_sampler_more_step(Ps, Pf, cfg, sde, (1, 3, 32, 32), O.Rng(), cfg.sampling.truncation_time)


def bpd_config():
    cfg = gu.mini_config("mini_vp_nll")
    cfg.eval.num_test_data, cfg.eval.num_nelbo, cfg.eval.batch_size, cfg.eval.truncation_time = 4, 2, 2, 1e-4
    return cfg


def parse_bpd_log(d):
    """the means evaluation.get_bpd logged (evaluation.py:423-471), in order"""
    import re
    vals = {}
    for line in [str(s) for s in d["bpd/log"]]:
        m = re.search(r"mean (nelbo|nelbo_residual|nll) bpd: ([0-9.e+-]+)", line)
        a = re.search(r"average nelbo bpd out of \d+ evaluations: ([0-9.e+-]+)", line)
        if a:
            vals.setdefault("avg", []).append(float(a.group(1)))
        elif m:
            tag = m.group(1) if m.group(1) != "nll" else ("nll_wrong" if "WRONG" in line else "nll_correct")
            if "num samples: 4" in line:
                vals.setdefault(tag, []).append(float(m.group(2)))
    return vals


def test_get_bpd_protocol():
    d, w = gu.load("mini_extra"), gu.load("mini_vp_nll")
    cfg = bpd_config()
    Ps, Pf = gu.state_dict(w, "score"), gu.state_dict(w, "flow")
    imgs = torch.from_numpy(d["bpd/images_nhwc"]).permute(0, 3, 1, 2).contiguous()
    rng = O.ReplayRng(gu.events(d, "bpd"))
    res, calls = O.get_bpd(Ps, Pf, cfg, O.get_sde(cfg), [imgs[:2], imgs[2:]], rng, eval=True, rtol=1e-3, atol=1e-3)
    assert rng.done() and len(calls) == int(d["bpd/n_calls"])
    for i, c in enumerate(calls):
        gu.assert_close(c[1], d[f"bpd/call{i:02d}_{c[0]}_a"], 2e-4, 2e-4, f"get_bpd call {i} ({c[0]})")
        if c[0] == "nelbo":
            gu.assert_close(c[2], d[f"bpd/call{i:02d}_nelbo_b"], 2e-4, 2e-4, f"get_bpd call {i} residual")
        else:
            ref = d[f"bpd/call{i:02d}_nll_b"]
            assert (float(c[2][0]), c[2][1], c[2][2]) == (ref[0], ref[1], int(ref[2]))
    logged = parse_bpd_log(d)
    assert np.allclose(res["nelbo_bpd_per_eval"], logged["nelbo"], rtol=2e-4)
    assert np.allclose([res["nelbo_bpd"], res["nelbo_residual_bpd"]], logged["avg"], rtol=2e-4)
    assert np.isclose(res["nll_wrong_bpd"], logged["nll_wrong"][-1], rtol=2e-4)
    assert np.isclose(res["nll_correct_bpd"], logged["nll_correct"][-1], rtol=2e-4)
    assert "nll_correct_train_eps_bpd" not in res      # training.truncation_time == 1e-5


def test_deeper_score_net():
    d = gu.load("mini_extra")
    cfg = gu.mini_config("mini_vp_nll")
    cfg.model.num_res_blocks = 2
    P = gu.sub(d, "nrb2")
    out = O.ncsnpp_forward(P, cfg, torch.from_numpy(d["nrb2_io/x"]), torch.from_numpy(d["nrb2_io/t"]) * 999)
    gu.assert_close(out, d["nrb2_io/raw_out"], 1e-4, 1e-5, "nrb=2 score net")


def test_celeba_squeeze_config():
    """BASELINE config #4 flavour (vp/CELEBA/indm_fid: 64x64, flow.squeeze, two-phase fid step) -- mini_celeba.npz"""
    d, cfg = gu.load("mini_celeba"), gu.mini_config("mini_celeba")
    assert cfg.flow.squeeze and cfg.data.image_size == 64
    sde = O.get_sde(cfg)
    x = torch.from_numpy(d["x"])
    Ps = gu.state_dict(d, "score")
    out = O.ncsnpp_forward(Ps, cfg, x, torch.from_numpy(d["score/t"]) * 999)
    gu.assert_close(out, d["score/raw_out"], 1e-4, 1e-5, "celeba score net")
    for mode in ("train", "eval"):
        rng = O.ReplayRng(gu.events(d, f"flow_{mode}"))
        with torch.no_grad():
            z, ld = O.flow_forward(gu.state_dict(d, "flow"), cfg, x.clone(), training=(mode == "train"), rng=rng,
                                   bn_updates={})
        assert rng.done()
        gu.assert_close(z, d[f"flow_{mode}/z"], 1e-5, 1e-5, f"celeba flow {mode} z")
        gu.assert_close(ld, d[f"flow_{mode}/logdet_kl"], 2e-5, 1e-4, f"celeba flow {mode} logdet")
    rng = O.ReplayRng(gu.events(d, "flow_reverse"))
    with torch.no_grad():
        xr, _ = O.flow_forward(gu.state_dict(d, "flow"), cfg, torch.from_numpy(d["flow_eval/z"]), log_det=None,
                               reverse=True, rng=rng)
    assert rng.done()
    gu.assert_close(xr, d["flow_reverse/x"], 1e-4, 1e-4, "celeba flow reverse")
    rng = O.ReplayRng(gu.events(d, "nelbo"))
    b1, b2 = O.nelbo_fn(Ps, gu.state_dict(d, "flow"), cfg, sde, x, rng)
    assert rng.done()
    gu.assert_close(b1, d["nelbo/bpd"], 1e-4, 1e-4, "celeba nelbo")
    gu.assert_close(b2, d["nelbo/bpd_residual"], 1e-4, 1e-4, "celeba nelbo residual")
    state = O.OracleState(gu.state_dict(d, "score"), cfg.optim.lr, cfg.model.ema_rate, cfg)
    fstate = O.OracleState(gu.state_dict(d, "flow"), cfg.flow.lr, cfg.flow.ema_rate, cfg)
    rng = O.ReplayRng(gu.events(d, "step0"))
    outs = O.flow_step_fn_fid(cfg, sde, state, fstate, torch.from_numpy(d["step0/x"]), rng)
    assert rng.done()
    for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), outs):
        gu.assert_close(v, d[f"step0/{nm}"], 2e-4, 1e-5, f"celeba step0 {nm}")
    for pre, st in (("score", state), ("flow", fstate)):
        for k in st.trainable:
            gu.assert_close(gu.stat(st.P[k]), d[f"after1/{pre}/stat/{k}"], 1e-5, 1e-6, "param " + k)
    assert [state.step, fstate.step, state.num_updates] == list(d["after1/steps"])


# ----------------------------------------------------------------------------- round-2 fixtures (mini_r2.npz)
@pytest.mark.parametrize("tag", ["mb2_nll", "mb2_fid"])
def test_micro_batched_step(tag):
    """optim.num_micro_batch = 2 (losses.py:280-309, :355-399)"""
    d = gu.load("mini_r2")
    fx = {"mb2_nll": "mini_vp_nll", "mb2_fid": "mini_vp_fid"}[tag]
    dw, cfg = gu.load(fx), gu.mini_config(fx)
    cfg.optim.num_micro_batch = 2
    sde = O.get_sde(cfg)
    state = O.OracleState(gu.state_dict(dw, "score"), cfg.optim.lr, cfg.model.ema_rate, cfg)
    fstate = O.OracleState(gu.state_dict(dw, "flow"), cfg.flow.lr, cfg.flow.ema_rate, cfg)
    step = O.flow_step_fn_nll if cfg.training.likelihood_weighting else O.flow_step_fn_fid
    rng = O.ReplayRng(gu.events(d, tag))
    out = step(cfg, sde, state, fstate, torch.from_numpy(d[f"{tag}/x"]), rng)
    assert rng.done()
    for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out):
        gu.assert_close(v, d[f"{tag}/{nm}"], 2e-4, 1e-5, f"{tag} {nm}")
    for pre, st in (("score", state), ("flow", fstate)):
        for k in st.trainable:
            gu.assert_close(gu.stat(st.P[k]), d[f"{tag}/after/{pre}/stat/{k}"], 1e-5, 1e-6, "param " + k)
    assert [state.step, fstate.step, state.num_updates] == list(d[f"{tag}/steps"])


def test_ve_pf_ode_likelihood_and_sampler():
    d, dv, cfg = gu.load("mini_r2"), gu.load("mini_ve"), gu.mini_config("mini_ve")
    Ps, Pf = gu.state_dict(dv, "score"), gu.state_dict(dv, "flow")
    sde = O.get_sde(cfg)
    x = torch.from_numpy(dv["x"])
    rng = O.ReplayRng(gu.events(d, "ve_nll"))
    bpd, z, nfe = O.likelihood_fn(Ps, Pf, cfg, sde, x, rng, rtol=1e-3, atol=1e-3)
    assert rng.done() and nfe == int(d["ve_nll/nfe"])
    gu.assert_close(bpd, d["ve_nll/bpd"], 1e-4, 1e-3, "ve nll bpd")
    gu.assert_close(z, d["ve_nll/z"], 1e-3, 1e-3, "ve nll z")
    rng = O.ReplayRng(gu.events(d, "ve_ode"))
    before, after, nfe = O.ode_sampler(Ps, Pf, cfg, sde, tuple(x.shape), rng, cfg.sampling.truncation_time, 1e-3, 1e-3)
    assert rng.done() and nfe == int(d["ve_ode/nfe"])
    gu.assert_close(before, d["ve_ode/before"], 1e-3, 1e-3, "ve ode before")
    gu.assert_close(after, d["ve_ode/after"], 1e-3, 1e-3, "ve ode after")


def test_step_after_reference_checkpoint():
    """the oracle restarts from the checkpoint files the reference's save_checkpoint wrote (utils.py:36-48)"""
    import os
    d, cfg = gu.load("mini_r2"), gu.mini_config("mini_vp_nll")
    sde = O.get_sde(cfg)
    states = []
    for name, lr, rate in (("score", cfg.optim.lr, cfg.model.ema_rate), ("flow", cfg.flow.lr, cfg.flow.ema_rate)):
        ck = torch.load(os.path.join(gu.GOLDEN, f"mini_r2_ckpt_{name}.pth"), map_location="cpu", weights_only=False)
        st = O.OracleState({k[len("module."):]: v for k, v in ck["model"].items()}, lr, rate, cfg)
        st.opt.load_state_dict(ck["optimizer"])
        st.ema = [s.clone() for s in ck["ema"]["shadow_params"]]
        st.num_updates, st.step = ck["ema"]["num_updates"], ck["step"]
        assert len(st.ema) == len(st.trainable)
        states.append(st)
    state, fstate = states
    rng = O.ReplayRng(gu.events(d, "ckpt"))
    out = O.flow_step_fn_nll(cfg, sde, state, fstate, torch.from_numpy(d["ckpt/x"]), rng)
    assert rng.done()
    for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out):
        gu.assert_close(v, d[f"ckpt/{nm}"], 2e-4, 1e-5, f"ckpt {nm}")
    for pre, st in (("score", state), ("flow", fstate)):
        for k in st.trainable:
            gu.assert_close(gu.stat(st.P[k]), d[f"ckpt/after/{pre}/stat/{k}"], 1e-5, 1e-6, "param " + k)
        for k, s_ in zip(st.trainable, st.ema):
            gu.assert_close(gu.stat(s_), d[f"ckpt/after/{pre}/emastat/{k}"], 1e-5, 1e-6, "ema " + k)
    assert [state.step, fstate.step, state.num_updates] == list(d["ckpt/steps"])


def test_chunked_step_matches_reference_goldens():
    """flow_step_fn_nll_chunked (the bounded-memory evaluation of the same step that the batch-128 GPU parity test
    uses) against the reference's own two-step golden: whole-batch encoder statistics, one image per chunk."""
    fx = "mini_vp_nll"
    d, cfg = gu.load(fx), gu.mini_config(fx)
    sde = O.get_sde(cfg)
    state = O.OracleState(gu.state_dict(d, "score"), cfg.optim.lr, cfg.model.ema_rate, cfg)
    fstate = O.OracleState(gu.state_dict(d, "flow"), cfg.flow.lr, cfg.flow.ema_rate, cfg)
    for it in range(2):
        out = O.flow_step_fn_nll_chunked(cfg, sde, state, fstate, torch.from_numpy(d[f"step{it}/x"]),
                                         gu.events(d, f"step{it}"), chunk=1)
        for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out):
            gu.assert_close(v, d[f"step{it}/{nm}"], 2e-4, 1e-5, f"chunked step{it} {nm}")
        if it == 0:
            for pre, st in (("score", state), ("flow", fstate)):
                for k in st.trainable:
                    key = f"step0/gradstat/{pre}/{k}"
                    if key in d:
                        gu.assert_close(gu.stat(st.P[k].grad), d[key], 1e-3, 1e-7, "chunked gradstat " + k)
    for pre, st in (("score", state), ("flow", fstate)):
        for k in st.trainable:
            gu.assert_close(gu.stat(st.P[k]), d[f"after2/{pre}/stat/{k}"], 1e-5, 1e-6, "chunked param " + k)
        for k, v in gu.sub(d, f"after2/{pre}/full").items():
            gu.assert_close(st.P[k], v, 1e-4, 2e-6, "chunked full " + k)
    for k, v in gu.sub(d, "after2/flow/buffers").items():
        gu.assert_close(fstate.P[k].float(), v.float(), 1e-5, 1e-6, "chunked " + k)


@pytest.mark.parametrize("fx,steps", [("mini_vp_fid", 2), ("mini_celeba", 1)])
def test_chunked_fid_step_matches_reference_goldens(fx, steps):
    """flow_step_fn_fid_chunked (the bounded-memory two-phase step the full-size CelebA shard test uses) against the
    reference's own step goldens (two steps on the VP-fid net, one on the CelebA-squeeze net), one image per chunk"""
    d, cfg = gu.load(fx), gu.mini_config(fx)
    sde = O.get_sde(cfg)
    state = O.OracleState(gu.state_dict(d, "score"), cfg.optim.lr, cfg.model.ema_rate, cfg)
    fstate = O.OracleState(gu.state_dict(d, "flow"), cfg.flow.lr, cfg.flow.ema_rate, cfg)
    for it in range(steps):
        out = O.flow_step_fn_fid_chunked(cfg, sde, state, fstate, torch.from_numpy(d[f"step{it}/x"]),
                                         gu.events(d, f"step{it}"), chunk=1)
        for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out):
            gu.assert_close(v, d[f"step{it}/{nm}"], 2e-4, 1e-5, f"chunked fid step{it} {nm}")
    for pre, st in (("score", state), ("flow", fstate)):
        for k in st.trainable:
            gu.assert_close(gu.stat(st.P[k]), d[f"after{steps}/{pre}/stat/{k}"], 1e-5, 1e-6, "chunked fid param " + k)
    for k, v in gu.sub(d, f"after{steps}/flow/buffers").items():
        gu.assert_close(fstate.P[k].float(), v.float(), 1e-5, 1e-6, "chunked fid " + k)


# ----------------------------------------------------------------------------- round-3 fixtures (mini_r3.npz)
R3_CASES = {"vp_recon": ("mini_vp_nll", "scoreflow", False), "vp_recon_ddpm": ("mini_vp_nll", "ddpm", False),
            "vp_recon_st": ("mini_vp_nll", "scoreflow", True), "vp_fid_recon": ("mini_vp_fid", "scoreflow", False),
            "ve_recon": ("mini_ve", "scoreflow", False), "ve_recon_st": ("mini_ve", "ddpm", True)}


@pytest.mark.parametrize("tag", sorted(R3_CASES))
def test_reconstruction_loss_and_sampled_truncation_time(tag):
    """get_sde_loss_fn.loss_fn with recon_loss=True (losses.py:120-140, both `variance` settings) and st=True
    (sde_lib.py:208-215 / 343-350): per-image losses, the sampled t_min and the gradient of the mean loss"""
    d = gu.load("mini_r3")
    fx, variance, st = R3_CASES[tag]
    dw, cfg = gu.load(fx), gu.mini_config(fx)
    sde = O.get_sde(cfg)
    P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in gu.state_dict(dw, "score").items()}
    np.random.seed(int(d[f"{tag}/numpy_seed"]))
    assert abs(O.get_t_min(sde, cfg, st) - float(d[f"{tag}/t_min"])) <= 1e-12      # numpy's global stream, as seeded
    rng = O.ReplayRng(gu.events(d, tag))
    loss = O.sde_loss(P, cfg, sde, torch.from_numpy(d[f"{tag}/x"]), True, rng, recon_loss=True, st=st, variance=variance)
    assert rng.done()
    gu.assert_close(loss, d[f"{tag}/loss"], 2e-5, 1e-6, f"{tag} loss")
    loss.mean().backward()
    keys = [k[len(f"{tag}/grad/stat/"):] for k in d if k.startswith(f"{tag}/grad/stat/")]
    assert len(keys) > 20
    for k in keys:
        gu.assert_close(gu.stat(P[k].grad), d[f"{tag}/grad/stat/{k}"], 2e-4, 1e-5 * max(1.0, float(abs(d[f"{tag}/grad/stat/{k}"]).max())), f"{tag} grad {k}")


@pytest.mark.parametrize("tag", ["st_nll", "st_fid", "st_fid_mb2"])
def test_training_st_steps(tag):
    """training.st = True through flow_step_fn_nll (losses.py:294) and the two-phase flow_step_fn_fid, whose second
    phase keeps and re-scales the first phase's score gradients (losses.py:378-395), with one and two micro-batches"""
    d = gu.load("mini_r3")
    fx = "mini_vp_nll" if tag == "st_nll" else "mini_vp_fid"
    dw, cfg = gu.load(fx), gu.mini_config(fx)
    cfg.training.st = True
    cfg.optim.num_micro_batch = 2 if tag.endswith("mb2") else 1
    sde = O.get_sde(cfg)
    state = O.OracleState(gu.state_dict(dw, "score"), cfg.optim.lr, cfg.model.ema_rate, cfg)
    fstate = O.OracleState(gu.state_dict(dw, "flow"), cfg.flow.lr, cfg.flow.ema_rate, cfg)
    step = O.flow_step_fn_nll if cfg.training.likelihood_weighting else O.flow_step_fn_fid
    rng = O.ReplayRng(gu.events(d, tag))
    out = step(cfg, sde, state, fstate, torch.from_numpy(d[f"{tag}/x"]), rng)
    assert rng.done()
    for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out):
        gu.assert_close(v, d[f"{tag}/{nm}"], 2e-4, 1e-5, f"{tag} {nm}")
    for pre, st in (("score", state), ("flow", fstate)):
        for k in st.trainable:
            gu.assert_close(gu.stat(st.P[k]), d[f"{tag}/after/{pre}/stat/{k}"], 1e-5, 1e-6, "param " + k)
    assert [state.step, fstate.step, state.num_updates] == list(d[f"{tag}/steps"])


def test_probability_flow_predictors():
    """mini_r4.npz: sampling.probability_flow = True through pc_sampler -- the reverse-diffusion predictor (VE net, with
    the Langevin corrector) against the reference's samples, and the Euler-Maruyama predictor, for which the reference
    itself raises TypeError (it indexes the float diffusion of the probability-flow RSDE)"""
    d = gu.load("mini_r4")
    dv, cfg = gu.load("mini_ve"), gu.mini_config("mini_ve")
    cfg.sampling.method, cfg.sampling.probability_flow, cfg.sampling.num_scales = "pc", True, 4
    assert str(d["pf_rd/predictor"]) == cfg.sampling.predictor == "reverse_diffusion"
    rng = O.ReplayRng(gu.events(d, "pf_rd"))
    before, after, nfe = O.pc_sampler(gu.state_dict(dv, "score"), gu.state_dict(dv, "flow"), cfg, O.get_sde(cfg),
                                      tuple(d["pf_rd/before"].shape), rng, cfg.sampling.truncation_time)
    assert rng.done() and nfe == int(d["pf_rd/nfe"])
    gu.assert_close(before, d["pf_rd/before"], 1e-4, 1e-4, "pf rd before")
    gu.assert_close(after, d["pf_rd/after"], 1e-3, 1e-3, "pf rd after")
    dp, cfg = gu.load("mini_vp_nll"), gu.mini_config("mini_vp_nll")
    cfg.sampling.method, cfg.sampling.probability_flow, cfg.sampling.num_scales = "pc", True, 4
    assert str(d["pf_em/raises"]) == "TypeError" and cfg.sampling.predictor == "euler_maruyama"
    with pytest.raises(TypeError):
        O.pc_sampler(gu.state_dict(dp, "score"), gu.state_dict(dp, "flow"), cfg, O.get_sde(cfg), (2, 3, 32, 32), O.Rng(),
                     cfg.sampling.truncation_time)
