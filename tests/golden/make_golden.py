"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE (/root/reference)
in this container (CPU, fp32) with inert stubs (ref_import.py).  Run manually:

    cd /root/repo && python tests/golden/make_golden.py

Fixtures hold data only: inputs, weights of a mini configuration, every random draw the reference
made (in call order) and the outputs it produced.  No reference source is copied.
The reference never travels to the GPU box; tests read only the .npz files.
"""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

torch = ref_import.install()
import numpy as np  # noqa: E402

import losses  # noqa: E402
import likelihood  # noqa: E402
import sampling  # noqa: E402
import sde_lib  # noqa: E402
import utils as ref_utils  # noqa: E402
from models import ncsnpp  # noqa: F401,E402
from models import utils as mutils  # noqa: E402
from models import layers as ref_layers  # noqa: E402
from models import layerspp as ref_layerspp  # noqa: E402
from models.ema import ExponentialMovingAverage  # noqa: E402
from flow_models.flow_model import create_flow_model, flow_forward  # noqa: E402
from op.upfirdn2d import upfirdn2d_native  # noqa: E402
from op.fused_act import fused_leaky_relu  # noqa: E402
import datasets as ref_datasets  # noqa: E402


class _NullFile(io.BytesIO):
    def __init__(self, *a, **k):
        super().__init__()


sys.modules["tensorflow"].io.gfile.GFile = _NullFile
sampling.tf = sys.modules["tensorflow"]


# ----------------------------------------------------------------------------- noise recording
class Recorder:
    def __init__(self):
        self.events = []

    @contextlib.contextmanager
    def on(self):
        o = dict(randn=torch.randn, randn_like=torch.randn_like, rand=torch.rand, rand_like=torch.rand_like,
                 randint_like=torch.randint_like, poisson=np.random.poisson, np_rand=np.random.rand)
        ev = self.events

        def randn(*a, **k):
            r = o["randn"](*a, **k); ev.append(("randn", r.detach().cpu().numpy().copy())); return r

        def randn_like(x, **k):
            r = o["randn_like"](x, **k); ev.append(("randn", r.detach().cpu().numpy().copy())); return r

        def rand(*a, **k):
            r = o["rand"](*a, **k); ev.append(("rand", r.detach().cpu().numpy().copy())); return r

        def rand_like(x, **k):
            r = o["rand_like"](x, **k); ev.append(("rand", r.detach().cpu().numpy().copy())); return r

        def randint_like(x, **k):
            r = o["randint_like"](x, **k)
            ev.append(("rademacher", (r.float() * 2 - 1).cpu().numpy().copy())); return r

        def poisson(lam, n):
            r = o["poisson"](lam, n); ev.append(("poisson", np.asarray(r[0], dtype=np.int64))); return r

        def np_rand(*a):   # sde.get_t_min(st=True) draws its truncation time here (sde_lib.py:208-215)
            r = o["np_rand"](*a); ev.append(("np_rand", np.asarray(r, dtype=np.float64))); return r

        torch.randn, torch.randn_like, torch.rand, torch.rand_like = randn, randn_like, rand, rand_like
        torch.randint_like, np.random.poisson, np.random.rand = randint_like, poisson, np_rand
        try:
            yield
        finally:
            torch.randn, torch.randn_like, torch.rand = o["randn"], o["randn_like"], o["rand"]
            torch.rand_like = o["rand_like"]
            torch.randint_like, np.random.poisson, np.random.rand = o["randint_like"], o["poisson"], o["np_rand"]


def pack_events(prefix, events, out):
    out[f"{prefix}/n_events"] = np.asarray(len(events))
    for i, (k, v) in enumerate(events):
        out[f"{prefix}/ev{i:03d}_{k}"] = np.asarray(v)


def pack_sd(prefix, module, out):
    for k, v in module.state_dict().items():
        out[f"{prefix}/{k[len('module.'):] if k.startswith('module.') else k}"] = v.detach().cpu().numpy().copy()


def tensor_stats(named):
    return {k: np.asarray([float(v.double().sum()), float(v.double().pow(2).sum())]) for k, v in named}


# ----------------------------------------------------------------------------- configs
def mini(cfg, batch=2):
    cfg.model.nf = 16
    cfg.model.ch_mult = (1, 2)
    cfg.model.num_res_blocks = 1
    cfg.model.dropout = 0.0
    cfg.flow.nblocks = "2-2"
    cfg.flow.intermediate_dim = 32
    # shrunken wolf JSON (same structure as wolf_configs/cifar10/glow/resflow-gaussian-uni.json:1-34)
    import json
    wolf = json.load(open(os.path.join(ref_import.REF, cfg.flow.model_config)))
    wolf["discriminator"]["encoder"]["hidden_planes"] = [8, 16, 16]
    wolf["discriminator"]["prior"]["hidden_features"] = 32
    path = "/tmp/mini_wolf.json"
    json.dump(wolf, open(path, "w"))
    cfg.flow.model_config = path
    cfg.training.batch_size = batch
    cfg.sampling.batch_size = batch
    cfg.sampling.num_scales = 3
    cfg.eval.batch_size = batch
    return cfg


def build(cfg, seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    score = mutils.create_model(cfg)
    flow = create_flow_model(cfg)
    # make the untrained nets non-degenerate: init_scale=0 zeroes the last conv / attention out-proj
    with torch.no_grad():
        for n, p in score.named_parameters():
            if p.abs().max() < 1e-6 and p.dim() > 1:
                p.copy_(torch.randn_like(p) * 0.05)
        for n, b in flow.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(torch.randn_like(b) * 0.1)
            if n.endswith("running_var"):
                b.copy_(torch.rand_like(b) + 0.5)
    return score, flow


def make_states(cfg, score, flow):
    opt = losses.get_optimizer(cfg, score.parameters())
    ema = ExponentialMovingAverage(score.parameters(), decay=cfg.model.ema_rate)
    state = dict(optimizer=opt, model=score, ema=ema, step=0)
    fopt = losses.get_optimizer(cfg, flow.parameters(), lr=cfg.flow.lr)
    fema = ExponentialMovingAverage(flow.parameters(), decay=cfg.flow.ema_rate)
    fstate = dict(optimizer=fopt, model=flow, ema=fema, step=0)
    return state, fstate


def data_batch(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, cfg.data.image_size, cfg.data.image_size, generator=g)
    x = (255.0 * x + torch.rand(x.shape, generator=g)) / 256.0  # run_lib.py:86
    return ref_datasets.get_data_scaler(cfg)(x)


# ----------------------------------------------------------------------------- fixtures
def golden_sde(out_dir):
    out = {}
    t = torch.tensor([1e-5, 8.2764716353e-05, 1e-3, 0.1, 0.5, 0.999, 1.0])
    u = torch.tensor([0.0, 1e-4, 0.01, 0.3, 0.5, 0.9, 0.9999])
    x = torch.linspace(-1, 1, 7 * 3 * 4 * 4).reshape(7, 3, 4, 4)
    out["t"], out["u"], out["x"] = t.numpy(), u.numpy(), x.numpy()
    cfg = ref_import.get_config()
    for name, sde in (("vp", sde_lib.VPSDE(1e-5, 0.1, 20.0, 1000)), ("ve", sde_lib.VESDE(1e-5, 0.01, 50.0, 1000))):
        mean, std = sde.marginal_prob(x, t)
        drift, diff = sde.sde(x, t)
        out[f"{name}/mean"], out[f"{name}/std"] = mean.numpy(), std.numpy()
        out[f"{name}/drift"], out[f"{name}/diffusion"] = drift.numpy(), diff.numpy()
        out[f"{name}/prior_logp"] = sde.prior_logp(x).numpy()
        f, G = sde.discretize(x, t, None)
        out[f"{name}/disc_f"], out[f"{name}/disc_G"] = f.numpy(), G.numpy()
        nt = t * 0.5
        f, G = sde.discretize(x, t, nt)
        out[f"{name}/disc2_f"], out[f"{name}/disc2_G"] = f.numpy(), G.numpy()
        out[f"{name}/antiderivative"] = sde.antiderivative(t).numpy()
        out[f"{name}/antiderivative_T"] = sde.antiderivative(1.0).numpy()
        out[f"{name}/Z"] = sde.normalizing_constant(1e-5).numpy()
        orig = torch.rand
        torch.rand = lambda *a, **k: u.clone()
        try:
            tt, Z = sde.get_diffusion_time(cfg, 7, torch.device("cpu"), 1e-5, importance_sampling=True)
        finally:
            torch.rand = orig
        out[f"{name}/is_t"] = tt.numpy()
        # reverse-time SDE / probability-flow ODE (sde_lib.py:74-120) around a closed-form stand-in score
        score_fn = lambda xx, tt_: torch.sin(3.0 * xx) * (1.0 + tt_)[:, None, None, None]  # noqa: E731
        for pf in (False, True):
            rsde = sde.reverse(score_fn, probability_flow=pf)
            tag = f"{name}/rsde_pf{int(pf)}"
            drift, diff = rsde.sde(x, t)
            out[f"{tag}/drift"] = drift.numpy()
            out[f"{tag}/diffusion"] = np.asarray(diff.numpy() if torch.is_tensor(diff) else diff, dtype=np.float32)
            for nm, nxt in (("disc", None), ("disc2", nt), ("disc0", torch.zeros_like(t))):
                f, G = rsde.discretize(x, t, nxt)
                out[f"{tag}/{nm}_f"], out[f"{tag}/{nm}_G"] = f.numpy(), G.numpy()
    out["timestep_embedding"] = ref_layers.get_timestep_embedding(t * 999, 128).numpy()
    torch.manual_seed(3)
    gfp = ref_layerspp.GaussianFourierProjection(embedding_size=8, scale=16.0)
    out["gfp/W"] = gfp.W.detach().numpy()
    out["gfp/out"] = gfp(torch.log(torch.tensor([0.01, 0.5, 3.0, 50.0]))).detach().numpy()
    np.savez_compressed(os.path.join(out_dir, "sde_known_answers.npz"), **out)


def golden_ops(out_dir):
    out = {}
    g = torch.Generator().manual_seed(11)
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k = torch.outer(k1, k1)
    k = k / k.sum()
    krand = torch.rand(4, 4, generator=g)
    # the 9 VE call signatures with shrunken channel counts + crops + random kernels
    cases = [(8, 32, 32, 1, 2, 1, 1), (3, 32, 32, 1, 1, 2, 2), (8, 16, 16, 1, 2, 1, 1), (8, 16, 16, 1, 1, 2, 2),
             (8, 8, 8, 1, 2, 1, 1), (8, 8, 8, 1, 1, 2, 2), (8, 4, 4, 2, 1, 2, 1), (8, 8, 8, 2, 1, 2, 1),
             (8, 16, 16, 2, 1, 2, 1), (5, 9, 7, 2, 3, -1, 2), (4, 8, 8, 1, 1, -1, -1)]
    out["cases"] = np.asarray(cases)
    out["k_fir"], out["k_rand"] = k.numpy(), krand.numpy()
    for i, (c, h, w, up, down, p0, p1) in enumerate(cases):
        for kn, kk in (("fir", k * (up ** 2)), ("rand", krand)):
            x = torch.rand(2, c, h, w, generator=g, requires_grad=True)
            y = upfirdn2d_native(x, kk, up, up, down, down, p0, p1, p0, p1)
            gy = torch.rand(y.shape, generator=g)
            (gx,) = torch.autograd.grad(y, x, gy)
            out[f"ufd{i}_{kn}/x"], out[f"ufd{i}_{kn}/y"] = x.detach().numpy(), y.detach().numpy()
            out[f"ufd{i}_{kn}/gy"], out[f"ufd{i}_{kn}/gx"] = gy.numpy(), gx.numpy()
            out[f"ufd{i}_{kn}/k"] = kk.numpy()
    x = torch.randn(3, 8, 5, 5, generator=g)
    b = torch.randn(8, generator=g)
    out["flr/x"], out["flr/b"] = x.numpy(), b.numpy()
    out["flr/y"] = fused_leaky_relu(x, b, 0.2, 2 ** 0.5).numpy()
    np.savez_compressed(os.path.join(out_dir, "native_ops.npz"), **out)


def golden_model(out_dir, name, cfg_name, patch):
    cfg = mini(ref_import.get_config(cfg_name))
    patch(cfg)
    out = {}
    score, flow = build(cfg, seed=1234)
    pack_sd("score", score, out)
    pack_sd("flow", flow, out)
    out["flow_param_order"] = np.asarray([n[len("module."):] for n, p in flow.named_parameters() if p.requires_grad])
    out["score_param_order"] = np.asarray([n[len("module."):] for n, p in score.named_parameters() if p.requires_grad])
    sde = sde_lib.get_sde(cfg)
    B = 2
    x = data_batch(cfg, B, seed=5)
    out["x"] = x.numpy()
    inverse_scaler = ref_datasets.get_data_inverse_scaler(cfg)
    scaler = ref_datasets.get_data_scaler(cfg)

    # ---- score network forward + grads (eval mode, fixed t) ----
    score.eval()
    t = torch.tensor([0.3, 0.9])
    xs = x.clone().requires_grad_(True)
    score_fn = mutils.get_score_fn(cfg, sde, score, train=False, continuous=True)
    s = score_fn(xs, t)
    gy = torch.linspace(-1, 1, s.numel()).reshape(s.shape)
    pnames = [n for n, p in score.named_parameters() if p.requires_grad]
    pick = [pnames[2], pnames[len(pnames) // 2], pnames[-2]] + [n for n in pnames if "NIN_1.W" in n][:1] + \
           [n for n in pnames if "GroupNorm_1.weight" in n][:1] + [n for n in pnames if "Dense_0.weight" in n][:1]
    params = dict(score.named_parameters())
    grads = torch.autograd.grad(s, [xs] + [params[n] for n in pick], gy)
    out["score/t"], out["score/out"], out["score/gy"] = t.numpy(), s.detach().numpy(), gy.numpy()
    out["score/gx"] = grads[0].numpy()
    for n, gr in zip(pick, grads[1:]):
        out[f"score/grad/{n[len('module.'):]}"] = gr.numpy()
    raw = score(x, t * 999 if cfg.training.sde == "vpsde" else sde.marginal_prob(x, t)[1])
    out["score/raw_out"] = raw.detach().numpy()

    # ---- flow forward: train (Neumann surrogate), eval (power series), no-logdet, reverse ----
    for mode in ("train", "eval"):
        flow.train() if mode == "train" else flow.eval()
        saved = {k: v.clone() for k, v in flow.state_dict().items()}
        rec = Recorder()
        with rec.on():
            z, ld = flow_forward(cfg, flow, x.clone(), reverse=False)
        if mode == "train":
            gz = torch.linspace(-1, 1, z.numel()).reshape(z.shape)
            fparams = dict(flow.named_parameters())
            fpn = [n for n, p in fparams.items() if p.requires_grad and "lamb" not in n and "geom_p" not in n]
            fpick = [fpn[0], fpn[2], fpn[4], fpn[5], fpn[6], fpn[8]] + [n for n in fpn if "encoder" in n][:2] + \
                    [n for n in fpn if "discriminator.fc" in n] + [n for n in fpn if "steps.0.linear.weight" in n] + \
                    [n for n in fpn if "steps.1.unit.coupling2_dn.net.fc3" in n] + \
                    [n for n in fpn if "steps.0.actnorm" in n] + [n for n in fpn if "transforms.1.chain.1.nnet.3" in n]
            gr = torch.autograd.grad([z, ld], [fparams[n] for n in fpick], [gz, torch.tensor([1.0, -0.5])])
            out["flow_train/gz"] = gz.numpy()
            for n, g_ in zip(fpick, gr):
                out[f"flow_train/grad/{n[len('module.'):]}"] = g_.numpy()
            for k, v in flow.state_dict().items():
                if "running" in k or "last_" in k or k.endswith(".scale") or "num_batches" in k:
                    out[f"flow_train/buffers/{k[len('module.'):]}"] = v.detach().numpy().copy()
        out[f"flow_{mode}/z"], out[f"flow_{mode}/logdet_kl"] = z.detach().numpy(), ld.detach().numpy()
        pack_events(f"flow_{mode}", rec.events, out)
        flow.load_state_dict(saved)
    flow.eval()
    rec = Recorder()
    with rec.on(), torch.no_grad():
        z2, _ = flow_forward(cfg, flow, x.clone(), log_det=None, reverse=False)
    out["flow_nologdet/z"] = z2.numpy()
    pack_events("flow_nologdet", rec.events, out)
    rec = Recorder()
    with rec.on(), torch.no_grad():
        xr, _ = flow_forward(cfg, flow, torch.tensor(out["flow_eval/z"]), log_det=None, reverse=True)
    out["flow_reverse/x"] = xr.numpy()
    pack_events("flow_reverse", rec.events, out)

    # ---- NELBO ----
    nelbo_fn = likelihood.get_elbo_fn(cfg, sde, inverse_scaler=inverse_scaler)
    score.eval(); flow.eval()
    rec = Recorder()
    with rec.on():
        n1, n2 = nelbo_fn(score, flow, x.clone())
    out["nelbo/bpd"], out["nelbo/bpd_residual"] = n1.detach().numpy(), n2.detach().numpy()
    pack_events("nelbo", rec.events, out)

    # ---- PC sampler, 3 scales ----
    shape = (B, 3, cfg.data.image_size, cfg.data.image_size)
    cfg.sampling.method = "pc"
    sampler = sampling.get_sampling_fn(cfg, sde, shape, inverse_scaler, cfg.sampling.truncation_time)
    rec = Recorder()
    with rec.on():
        before, after, nfe = sampler(score, flow, sample_dir="/tmp", r=0)
    out["pc/before"], out["pc/after"], out["pc/nfe"] = before.numpy(), after.numpy(), np.asarray(nfe)
    pack_events("pc", rec.events, out)

    # ---- PF-ODE: one RHS evaluation + full RK45 NLL + ODE sampler ----
    if cfg.training.sde == "vpsde":
        lik = likelihood.get_likelihood_fn(cfg, sde, inverse_scaler, rtol=1e-3, atol=1e-3)
        rec = Recorder()
        with rec.on():
            bpd, zlat, nfe = lik(score, flow, x.clone())
        out["nll/bpd"], out["nll/z"], out["nll/nfe"] = bpd.numpy(), zlat.numpy(), np.asarray(nfe)
        pack_events("nll", rec.events, out)
        score_fn = mutils.get_score_fn(cfg, sde, score, train=False, continuous=True)
        rsde = sde.reverse(score_fn, probability_flow=True)
        eps_r = torch.tensor(out["nll/ev%03d_rademacher" % [i for i, e in enumerate(rec.events) if e[0] == "rademacher"][0]])
        vt = torch.ones(B) * 0.4
        xx = x.clone()
        drift = rsde.sde(xx, vt)[0]
        div = likelihood.get_div_fn(lambda a, b: rsde.sde(a, b)[0])(xx, vt, eps_r)
        out["ode_rhs/t"], out["ode_rhs/drift"], out["ode_rhs/div"] = vt.numpy(), drift.detach().numpy(), div.numpy()
        out["ode_rhs/eps"] = eps_r.numpy()
        cfg.sampling.method = "ode"
        cfg.eval.rtol, cfg.eval.atol = 1e-3, 1e-3
        osamp = sampling.get_sampling_fn(cfg, sde, shape, inverse_scaler, cfg.sampling.truncation_time)
        rec = Recorder()
        with rec.on():
            before, after, nfe = osamp(score, flow)
        out["ode/before"], out["ode/after"], out["ode/nfe"] = before.numpy(), after.numpy(), np.asarray(nfe)
        pack_events("ode", rec.events, out)

    # ---- two training steps (the north-star path) ----
    optimize_fn = losses.optimization_manager(cfg)
    step_fn = losses.get_step_fn(cfg, sde, train=True, optimize_fn=optimize_fn, scaler=scaler)
    state, fstate = make_states(cfg, score, flow)
    score.train(); flow.train()
    for it in range(2):
        xb = data_batch(cfg, B, seed=100 + it)
        out[f"step{it}/x"] = xb.numpy()
        rec = Recorder()
        with rec.on():
            l, ls, lf, lp = step_fn(state, fstate, xb)
        for nm, v in (("loss", l), ("loss_score", ls), ("loss_flow", lf), ("loss_logp", lp)):
            out[f"step{it}/{nm}"] = v.numpy()
        pack_events(f"step{it}", rec.events, out)
        if it == 0:
            gs = {n[len("module."):]: p.grad for n, p in score.named_parameters() if p.grad is not None}
            gf = {n[len("module."):]: p.grad for n, p in flow.named_parameters() if p.grad is not None}
            # grads after clipping (optimize_fn clips in place, losses.py:58-59)
            for k, v in tensor_stats(gs.items()).items():
                out[f"step0/gradstat/score/{k}"] = v
            for k, v in tensor_stats(gf.items()).items():
                out[f"step0/gradstat/flow/{k}"] = v
    for pre, mod, st in (("score", score, state), ("flow", flow, fstate)):
        named = [(n[len("module."):], p.detach()) for n, p in mod.named_parameters()]
        for k, v in tensor_stats(named).items():
            out[f"after2/{pre}/stat/{k}"] = v
        tr = [n for n, p in named if dict(mod.named_parameters())["module." + n].requires_grad]
        for n, s_ in zip(tr, st["ema"].shadow_params):
            out[f"after2/{pre}/emastat/{n}"] = np.asarray([float(s_.double().sum()), float(s_.double().pow(2).sum())])
        for n, p in named[:: max(1, len(named) // 6)]:
            out[f"after2/{pre}/full/{n}"] = p.numpy().copy()
    for k, v in flow.state_dict().items():
        if "running" in k or "num_batches" in k:
            out[f"after2/flow/buffers/{k[len('module.'):]}"] = v.detach().numpy().copy()
    out["after2/steps"] = np.asarray([state["step"], fstate["step"], state["ema"].num_updates])
    np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **out)
    print(name, "done:", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")


# ----------------------------------------------------------------------------- section-8(f) rows: sampler variants,
# the get_bpd evaluation protocol, a deeper score net, the CelebA (squeeze) flow
def _load_fixture_weights(fixture, score, flow):
    d = dict(np.load(os.path.join(HERE, fixture + ".npz")))
    skip = {"t", "out", "gy", "gx", "raw_out"}
    for pre, mod in (("score", score), ("flow", flow)):
        sd = {}
        for k, v in d.items():
            if k.startswith(pre + "/") and k[len(pre) + 1:] not in skip and not k.startswith(pre + "/grad/"):
                sd["module." + k[len(pre) + 1:]] = torch.from_numpy(v)
        mod.load_state_dict(sd)
    return d


class _TfBatch:
    """what `datasets.get_batch_` (datasets.py:115-128) expects from a TFDS iterator: batch['image']._numpy()"""
    def __init__(self, a):
        self.a = a

    def _numpy(self):
        return self.a


def golden_extra(out_dir):
    import logging
    out = {}
    # ---- (f3) pc_sampler_search / denoise_update_fn / pc_sampler_more_step, VE mini, sde.N = 4 ----
    cfg = mini(ref_import.get_config("configs.ve.CIFAR10.indm"))
    score, flow = build(cfg, seed=1234)
    _load_fixture_weights("mini_ve", score, flow)
    cfg.model.num_scales = 4            # sde.N (the score net's `sigmas` buffer is only read by discrete-time models)
    score.eval(); flow.eval()
    sde = sde_lib.get_sde(cfg)
    inverse_scaler = ref_datasets.get_data_inverse_scaler(cfg)
    B = 2
    shape = (B, 3, 32, 32)
    cfg.sampling.method = "pc"
    cfg.sampling.pc_denoise = True
    cfg.sampling.need_sample = False
    sampler = sampling.get_sampling_fn(cfg, sde, shape, inverse_scaler, cfg.sampling.truncation_time)
    rec = Recorder()
    with rec.on():
        before, after, nfe = sampler(score, flow, final_time=0.0)
    out["search/before"], out["search/after"], out["search/nfe"] = before.numpy(), after.numpy(), np.asarray(nfe)
    pack_events("search", rec.events, out)
    g = torch.Generator().manual_seed(11)
    bd = torch.randn(shape, generator=g) * 0.5
    out["search_bd/before_data"] = bd.numpy()
    rec = Recorder()
    with rec.on():
        before, after, nfe = sampler(score, flow, final_time=5e-6, before_data=bd.clone(), temperature=0.9)
    out["search_bd/before"], out["search_bd/after"] = before.numpy(), after.numpy()
    pack_events("search_bd", rec.events, out)
    cfg.sampling.need_sample = True     # no denoising step: before_data passes straight to the flow inverse
    rec = Recorder()
    with rec.on():
        before, after, nfe = sampling.get_sampling_fn(cfg, sde, shape, inverse_scaler, cfg.sampling.truncation_time)(
            score, flow, before_data=bd.clone())
    out["search_ns/before"], out["search_ns/after"] = before.numpy(), after.numpy()
    pack_events("search_ns", rec.events, out)
    cfg.sampling.pc_denoise = False
    cfg.sampling.more_step = True
    cfg.sampling.need_sample = False
    shape1 = (1, 3, 32, 32)
    sampler = sampling.get_sampling_fn(cfg, sde, shape1, inverse_scaler, cfg.sampling.truncation_time)
    rec = Recorder()
    with rec.on():
        before, after, nfe = sampler(score, flow, before_data=bd[:1].clone())
    out["more_step/before"], out["more_step/after"] = before.numpy(), after.numpy()
    pack_events("more_step", rec.events, out)
    try:  # before_data=None: the reference's own loop is broken (sampling.py:512 omits snr; :510 indexes past the end)
        sampler(score, flow)
        out["more_step/none_raises"] = np.asarray("")
    except Exception as e:  # noqa: BLE001
        out["more_step/none_raises"] = np.asarray(type(e).__name__)

    # ---- (f2) evaluation.get_bpd protocol, VP-NLL mini ----
    ref_import._stub("cleanfid"); ref_import._stub("cleanfid.fid")
    sys.modules["cleanfid"].fid = sys.modules["cleanfid.fid"]
    tf = sys.modules["tensorflow"]
    tf.float32, tf.function = "float32", (lambda f: f)
    import evaluation
    cfg = mini(ref_import.get_config("configs.vp.CIFAR10.indm_nll"))
    score, flow = build(cfg, seed=1234)
    _load_fixture_weights("mini_vp_nll", score, flow)
    sde = sde_lib.get_sde(cfg)
    inverse_scaler = ref_datasets.get_data_inverse_scaler(cfg)
    scaler = ref_datasets.get_data_scaler(cfg)
    cfg.eval.num_test_data = 4
    cfg.eval.num_nelbo = 2
    cfg.eval.batch_size = 2
    cfg.eval.truncation_time = 1e-4
    g = torch.Generator().manual_seed(21)
    imgs = torch.randint(0, 256, (4, 32, 32, 3), generator=g).float() / 255.0   # NHWC in [0,1] as TFDS yields
    out["bpd/images_nhwc"] = imgs.numpy()
    eval_ds = [{"image": _TfBatch(imgs[i:i + 2].numpy())} for i in (0, 2)]
    nelbo_fn = likelihood.get_elbo_fn(cfg, sde, inverse_scaler=inverse_scaler)
    nll_fn = likelihood.get_likelihood_fn(cfg, sde, inverse_scaler, rtol=1e-3, atol=1e-3)
    calls = []

    def nelbo_rec(*a, **k):
        r = nelbo_fn(*a, **k); calls.append(("nelbo", r[0].detach().numpy().copy(), r[1].detach().numpy().copy())); return r

    def nll_rec(*a, **k):
        r = nll_fn(*a, **k)
        calls.append(("nll", r[0].detach().numpy().copy(), np.asarray([float(k["residual"]), k["eps_bpd"], r[2]])))
        return r

    msgs = []

    class H(logging.Handler):
        def emit(self, record):
            msgs.append(record.getMessage())

    h = H(); logging.getLogger().addHandler(h); logging.getLogger().setLevel(logging.INFO)
    rec = Recorder()
    with rec.on():
        evaluation.get_bpd(cfg, eval_ds, scaler, nelbo_rec, nll_rec, score, flow, step=7, eval=True)
    logging.getLogger().removeHandler(h)
    pack_events("bpd", rec.events, out)
    for i, c in enumerate(calls):
        out[f"bpd/call{i:02d}_{c[0]}_a"], out[f"bpd/call{i:02d}_{c[0]}_b"] = c[1], c[2]
    out["bpd/n_calls"] = np.asarray(len(calls))
    out["bpd/log"] = np.asarray([m for m in msgs if m.startswith("step:")])
    out["bpd/flow_training_after"] = np.asarray(int(flow.training))

    # ---- config #5 flavour: more than one residual block per resolution (num_res_for_upsampling = nrb+1) ----
    cfg = mini(ref_import.get_config("configs.vp.CIFAR10.indm_nll"))
    cfg.model.num_res_blocks = 2
    score, _ = build(cfg, seed=77)
    pack_sd("nrb2", score, out)
    score.eval()
    x = data_batch(cfg, 2, seed=9)
    t = torch.tensor([0.2, 0.8])
    out["nrb2_io/x"], out["nrb2_io/t"] = x.numpy(), t.numpy()
    out["nrb2_io/raw_out"] = score(x, t * 999).detach().numpy()

    np.savez_compressed(os.path.join(out_dir, "mini_extra.npz"), **out)
    print("mini_extra done:", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")


def golden_celeba(out_dir):
    """BASELINE config #4 flavour: vp/CELEBA/indm_fid -- 64x64 images, flow.squeeze=True (the flow and its encoder
    see the space-to-depth image [12,32,32]), imagenet/64x64 wolf JSON, two-phase flow_step_fn_fid."""
    cfg = mini(ref_import.get_config("configs.vp.CELEBA.indm_fid"))
    cfg.model.ch_mult = (1, 2, 2)       # 64 -> 32 -> 16: keeps one attention block (attn_resolutions = (16,))
    out = {}
    score, flow = build(cfg, seed=4321)
    pack_sd("score", score, out)
    pack_sd("flow", flow, out)
    sde = sde_lib.get_sde(cfg)
    B = 2
    x = data_batch(cfg, B, seed=6)
    out["x"] = x.numpy()
    inverse_scaler = ref_datasets.get_data_inverse_scaler(cfg)
    scaler = ref_datasets.get_data_scaler(cfg)
    score.eval()
    t = torch.tensor([0.25, 0.75])
    out["score/t"] = t.numpy()
    out["score/raw_out"] = score(x, t * 999).detach().numpy()
    for mode in ("train", "eval"):
        flow.train() if mode == "train" else flow.eval()
        saved = {k: v.clone() for k, v in flow.state_dict().items()}
        rec = Recorder()
        with rec.on():
            z, ld = flow_forward(cfg, flow, x.clone(), reverse=False)
        out[f"flow_{mode}/z"], out[f"flow_{mode}/logdet_kl"] = z.detach().numpy(), ld.detach().numpy()
        pack_events(f"flow_{mode}", rec.events, out)
        flow.load_state_dict(saved)
    flow.eval()
    rec = Recorder()
    with rec.on(), torch.no_grad():
        xr, _ = flow_forward(cfg, flow, torch.tensor(out["flow_eval/z"]), log_det=None, reverse=True)
    out["flow_reverse/x"] = xr.numpy()
    pack_events("flow_reverse", rec.events, out)
    nelbo_fn = likelihood.get_elbo_fn(cfg, sde, inverse_scaler=inverse_scaler)
    rec = Recorder()
    with rec.on():
        n1, n2 = nelbo_fn(score, flow, x.clone())
    out["nelbo/bpd"], out["nelbo/bpd_residual"] = n1.detach().numpy(), n2.detach().numpy()
    pack_events("nelbo", rec.events, out)
    optimize_fn = losses.optimization_manager(cfg)
    step_fn = losses.get_step_fn(cfg, sde, train=True, optimize_fn=optimize_fn, scaler=scaler)
    state, fstate = make_states(cfg, score, flow)
    score.train(); flow.train()
    xb = data_batch(cfg, B, seed=100)
    out["step0/x"] = xb.numpy()
    rec = Recorder()
    with rec.on():
        l, ls, lf, lp = step_fn(state, fstate, xb)
    for nm, v in (("loss", l), ("loss_score", ls), ("loss_flow", lf), ("loss_logp", lp)):
        out[f"step0/{nm}"] = v.numpy()
    pack_events("step0", rec.events, out)
    for pre, mod, st in (("score", score, state), ("flow", flow, fstate)):
        named = [(n[len("module."):], p.detach()) for n, p in mod.named_parameters()]
        for k, v in tensor_stats(named).items():
            out[f"after1/{pre}/stat/{k}"] = v
    out["after1/steps"] = np.asarray([state["step"], fstate["step"], state["ema"].num_updates])
    np.savez_compressed(os.path.join(out_dir, "mini_celeba.npz"), **out)
    print("mini_celeba done:", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")


# ----------------------------------------------------------------------------- round-2 fixtures
def golden_round2(out_dir):
    """mini_r2.npz + mini_r2_ckpt_{score,flow}.pth:
      * optim.num_micro_batch = 2 training steps (losses.py:280-309), nll and fid step functions, batch 4;
      * VE likelihood_fn / ODE sampler (likelihood.py:41-140, sampling.py:547-621);
      * checkpoints written by the REFERENCE's utils.save_checkpoint (utils.py:36-48) after one optimisation step and
        the step that follows a restore (losses, parameter statistics);
      * moments of the reference's initialisers (layers.py:54-91 default_init / variance_scaling)."""
    out = {}
    # ---- micro-batched steps ----
    for tag, fixture, cfg_name, patch in (("mb2_nll", "mini_vp_nll", "configs.vp.CIFAR10.indm_nll", patch_vp_nll),
                                          ("mb2_fid", "mini_vp_fid", "configs.vp.CIFAR10.indm_fid", patch_vp_fid)):
        cfg = mini(ref_import.get_config(cfg_name), batch=4)
        patch(cfg)
        cfg.optim.num_micro_batch = 2
        score, flow = build(cfg, seed=1234)
        _load_fixture_weights(fixture, score, flow)
        sde = sde_lib.get_sde(cfg)
        scaler = ref_datasets.get_data_scaler(cfg)
        step_fn = losses.get_step_fn(cfg, sde, train=True, optimize_fn=losses.optimization_manager(cfg), scaler=scaler)
        state, fstate = make_states(cfg, score, flow)
        score.train(); flow.train()
        xb = data_batch(cfg, 4, seed=300)
        out[f"{tag}/x"] = xb.numpy()
        rec = Recorder()
        with rec.on():
            l, ls, lf, lp = step_fn(state, fstate, xb)
        for nm, v in (("loss", l), ("loss_score", ls), ("loss_flow", lf), ("loss_logp", lp)):
            out[f"{tag}/{nm}"] = v.numpy()
        pack_events(tag, rec.events, out)
        for pre, mod in (("score", score), ("flow", flow)):
            named = [(n[len("module."):], p.detach()) for n, p in mod.named_parameters()]
            for k, v in tensor_stats(named).items():
                out[f"{tag}/after/{pre}/stat/{k}"] = v
        out[f"{tag}/steps"] = np.asarray([state["step"], fstate["step"], state["ema"].num_updates])

    # ---- VE: RK45 likelihood + ODE sampler ----
    cfg = mini(ref_import.get_config("configs.ve.CIFAR10.indm"))
    score, flow = build(cfg, seed=1234)
    d = _load_fixture_weights("mini_ve", score, flow)
    score.eval(); flow.eval()
    sde = sde_lib.get_sde(cfg)
    inverse_scaler = ref_datasets.get_data_inverse_scaler(cfg)
    x = torch.from_numpy(d["x"])
    lik = likelihood.get_likelihood_fn(cfg, sde, inverse_scaler, rtol=1e-3, atol=1e-3)
    rec = Recorder()
    with rec.on():
        bpd, zlat, nfe = lik(score, flow, x.clone())
    out["ve_nll/bpd"], out["ve_nll/z"], out["ve_nll/nfe"] = bpd.numpy(), zlat.numpy(), np.asarray(nfe)
    pack_events("ve_nll", rec.events, out)
    cfg.sampling.method = "ode"
    cfg.eval.rtol, cfg.eval.atol = 1e-3, 1e-3
    osamp = sampling.get_sampling_fn(cfg, sde, tuple(x.shape), inverse_scaler, cfg.sampling.truncation_time)
    rec = Recorder()
    with rec.on():
        before, after, nfe = osamp(score, flow)
    out["ve_ode/before"], out["ve_ode/after"], out["ve_ode/nfe"] = before.numpy(), after.numpy(), np.asarray(nfe)
    pack_events("ve_ode", rec.events, out)

    # ---- checkpoints written by the reference ----
    cfg = mini(ref_import.get_config("configs.vp.CIFAR10.indm_nll"))
    score, flow = build(cfg, seed=1234)
    _load_fixture_weights("mini_vp_nll", score, flow)
    sde = sde_lib.get_sde(cfg)
    scaler = ref_datasets.get_data_scaler(cfg)
    step_fn = losses.get_step_fn(cfg, sde, train=True, optimize_fn=losses.optimization_manager(cfg), scaler=scaler)
    state, fstate = make_states(cfg, score, flow)
    score.train(); flow.train()
    step_fn(state, fstate, data_batch(cfg, 2, seed=400))
    ref_utils.save_checkpoint(cfg, os.path.join(out_dir, "mini_r2_ckpt_score.pth"), state)
    ref_utils.save_checkpoint(cfg, os.path.join(out_dir, "mini_r2_ckpt_flow.pth"), fstate)
    xb = data_batch(cfg, 2, seed=401)
    out["ckpt/x"] = xb.numpy()
    rec = Recorder()
    with rec.on():
        l, ls, lf, lp = step_fn(state, fstate, xb)
    for nm, v in (("loss", l), ("loss_score", ls), ("loss_flow", lf), ("loss_logp", lp)):
        out[f"ckpt/{nm}"] = v.numpy()
    pack_events("ckpt", rec.events, out)
    for pre, mod, st in (("score", score, state), ("flow", flow, fstate)):
        named = [(n[len("module."):], p.detach()) for n, p in mod.named_parameters()]
        for k, v in tensor_stats(named).items():
            out[f"ckpt/after/{pre}/stat/{k}"] = v
        tr = [n for n, p in named if dict(mod.named_parameters())["module." + n].requires_grad]
        for n, s_ in zip(tr, st["ema"].shadow_params):
            out[f"ckpt/after/{pre}/emastat/{n}"] = np.asarray([float(s_.double().sum()), float(s_.double().pow(2).sum())])
    out["ckpt/steps"] = np.asarray([state["step"], fstate["step"], state["ema"].num_updates])

    # ---- initialiser moments (D5): default_init(scale) = variance_scaling(scale, 'fan_avg', 'uniform') ----
    torch.manual_seed(99)
    shapes = [(128, 3, 3, 3), (256, 128, 3, 3), (256, 256, 1, 1), (512, 128), (128, 256)]
    out["init/shapes"] = np.asarray([list(s_) + [0] * (4 - len(s_)) for s_ in shapes])
    for i, shp in enumerate(shapes):
        for scale in (1.0, 0.1, 0.0):
            w = ref_layers.default_init(scale)(shp)
            out[f"init/{i}/scale{scale}"] = np.asarray([float(w.double().mean()), float(w.double().var(unbiased=False)),
                                                        float(w.min()), float(w.max())])
    # whole-network initialisation from a seed (mutils.create_model / create_flow_model constructors)
    cfg = mini(ref_import.get_config("configs.vp.CIFAR10.indm_nll"))
    torch.manual_seed(7)
    np.random.seed(7)
    score = mutils.create_model(cfg)
    torch.manual_seed(8)
    flow = create_flow_model(cfg)
    for pre, mod in (("score", score), ("flow", flow)):
        for k, v in mod.state_dict().items():
            v = v.detach().double()
            out[f"init_net/{pre}/{k[len('module.'):]}"] = np.asarray([float(v.sum()), float(v.pow(2).sum()),
                                                                      float(v.min()), float(v.max())])
    np.savez_compressed(os.path.join(out_dir, "mini_r2.npz"), **out)
    print("mini_r2 done:", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")


def golden_round3(out_dir):
    """mini_r3.npz: the optional branches of get_sde_loss_fn.loss_fn -- the reconstruction term (losses.py:120-140, both
    `variance` settings) and the sampled truncation time st=True (sde_lib.py:208-215 / 343-350) -- on the mini nets of
    mini_vp_nll / mini_vp_fid / mini_ve: per-image losses, the sampled t_min, gradient statistics of the mean loss."""
    out = {}
    cases = (("vp_recon", "mini_vp_nll", "configs.vp.CIFAR10.indm_nll", patch_vp_nll, "scoreflow", False),
             ("vp_recon_ddpm", "mini_vp_nll", "configs.vp.CIFAR10.indm_nll", patch_vp_nll, "ddpm", False),
             ("vp_recon_st", "mini_vp_nll", "configs.vp.CIFAR10.indm_nll", patch_vp_nll, "scoreflow", True),
             ("vp_fid_recon", "mini_vp_fid", "configs.vp.CIFAR10.indm_fid", patch_vp_fid, "scoreflow", False),
             ("ve_recon", "mini_ve", "configs.ve.CIFAR10.indm", patch_ve, "scoreflow", False),
             ("ve_recon_st", "mini_ve", "configs.ve.CIFAR10.indm", patch_ve, "ddpm", True))
    for tag, fixture, cfg_name, patch, variance, st in cases:
        cfg = mini(ref_import.get_config(cfg_name))
        patch(cfg)
        score, flow = build(cfg, seed=1234)
        _load_fixture_weights(fixture, score, flow)
        sde = sde_lib.get_sde(cfg)
        loss_fn = losses.get_sde_loss_fn(cfg, sde, train=True, variance=variance)
        x = data_batch(cfg, 2, seed=400)
        score.train()
        np.random.seed(77)                       # get_t_min(st=True) draws from numpy's global generator
        tmin_probe = sde.get_t_min(cfg, st)
        np.random.seed(77)
        rec = Recorder()
        with rec.on():
            l = loss_fn(score, x, st=st, recon_loss=True)
        score.zero_grad()
        l.mean().backward()
        out[f"{tag}/x"], out[f"{tag}/loss"] = x.numpy(), l.detach().numpy()
        out[f"{tag}/t_min"] = np.asarray(float(tmin_probe))
        out[f"{tag}/numpy_seed"] = np.asarray(77)
        pack_events(tag, rec.events, out)
        named = [(n[len("module."):], p.grad.detach()) for n, p in score.named_parameters() if p.grad is not None]
        for k, v in tensor_stats(named).items():
            out[f"{tag}/grad/stat/{k}"] = v
        print(tag, l.detach().numpy(), float(tmin_probe), len(rec.events), flush=True)
    # ---- training.st = True through both step functions (losses.py:294 and :378-395, incl. the in-place re-scaling of the
    # first phase's score gradients), one and two micro-batches ----
    for tag, fixture, cfg_name, patch, nmb in (("st_nll", "mini_vp_nll", "configs.vp.CIFAR10.indm_nll", patch_vp_nll, 1),
                                               ("st_fid", "mini_vp_fid", "configs.vp.CIFAR10.indm_fid", patch_vp_fid, 1),
                                               ("st_fid_mb2", "mini_vp_fid", "configs.vp.CIFAR10.indm_fid", patch_vp_fid, 2)):
        cfg = mini(ref_import.get_config(cfg_name), batch=4)
        patch(cfg)
        cfg.training.st = True
        cfg.optim.num_micro_batch = nmb
        score, flow = build(cfg, seed=1234)
        _load_fixture_weights(fixture, score, flow)
        sde = sde_lib.get_sde(cfg)
        scaler = ref_datasets.get_data_scaler(cfg)
        step_fn = losses.get_step_fn(cfg, sde, train=True, optimize_fn=losses.optimization_manager(cfg), scaler=scaler)
        state, fstate = make_states(cfg, score, flow)
        score.train(); flow.train()
        xb = data_batch(cfg, 4, seed=500)
        out[f"{tag}/x"] = xb.numpy()
        np.random.seed(78)
        out[f"{tag}/numpy_seed"] = np.asarray(78)
        rec = Recorder()
        with rec.on():
            l, ls, lf, lp = step_fn(state, fstate, xb)
        for nm, v in (("loss", l), ("loss_score", ls), ("loss_flow", lf), ("loss_logp", lp)):
            out[f"{tag}/{nm}"] = v.numpy()
        pack_events(tag, rec.events, out)
        for pre, mod in (("score", score), ("flow", flow)):
            named = [(n[len("module."):], p.detach()) for n, p in mod.named_parameters()]
            for k, v in tensor_stats(named).items():
                out[f"{tag}/after/{pre}/stat/{k}"] = v
        out[f"{tag}/steps"] = np.asarray([state["step"], fstate["step"], state["ema"].num_updates])
        print(tag, l.numpy(), ls.numpy(), len(rec.events), flush=True)
    np.savez_compressed(os.path.join(out_dir, "mini_r3.npz"), **out)


def golden_round4(out_dir):
    """mini_r4.npz: the PC sampler with sampling.probability_flow = True (Predictor.__init__ sampling.py:139-144 builds
    rsde = sde.reverse(score_fn, probability_flow): half the score term and no diffusion) -- reverse diffusion + Langevin
    on the mini VE net, 4 scales; and the Euler-Maruyama predictor on the mini VP net, where the reference RAISES
    (RSDE.sde returns the float 0. as diffusion, which update_fn indexes, sampling.py:196): the exception type is the golden."""
    out = {}
    for tag, fixture, cfg_name, patch in (("pf_em", "mini_vp_nll", "configs.vp.CIFAR10.indm_nll", patch_vp_nll),
                                          ("pf_rd", "mini_ve", "configs.ve.CIFAR10.indm", patch_ve)):
        cfg = mini(ref_import.get_config(cfg_name))
        patch(cfg)
        cfg.sampling.method = "pc"
        cfg.sampling.probability_flow = True
        cfg.sampling.num_scales = 4
        score, flow = build(cfg, seed=1234)
        _load_fixture_weights(fixture, score, flow)
        sde = sde_lib.get_sde(cfg)
        score.eval(); flow.eval()
        shape = (2, 3, cfg.data.image_size, cfg.data.image_size)
        sampler = sampling.get_sampling_fn(cfg, sde, shape, ref_datasets.get_data_inverse_scaler(cfg),
                                           cfg.sampling.truncation_time)
        torch.manual_seed(91)
        rec = Recorder()
        try:
            with rec.on():
                before, after, nfe = sampler(score, flow, sample_dir="/tmp", r=0)
        except Exception as e:  # noqa: BLE001  -- the reference's own behaviour for this setting is the golden
            out[f"{tag}/raises"] = np.asarray(type(e).__name__)
            print(tag, "the reference raises", type(e).__name__, e, flush=True)
            continue
        out[f"{tag}/before"], out[f"{tag}/after"], out[f"{tag}/nfe"] = before.numpy(), after.numpy(), np.asarray(nfe)
        out[f"{tag}/predictor"] = np.asarray(cfg.sampling.predictor)
        pack_events(tag, rec.events, out)
        print(tag, cfg.sampling.predictor, cfg.sampling.corrector, float(before.abs().mean()), len(rec.events), flush=True)
    np.savez_compressed(os.path.join(out_dir, "mini_r4.npz"), **out)


def patch_vp_nll(cfg):
    pass


def patch_vp_fid(cfg):
    cfg.training.likelihood_weighting = False
    cfg.training.importance_sampling = False


def patch_ve(cfg):
    pass


if __name__ == "__main__":
    which = sys.argv[1:] or ["sde", "ops", "vp_nll", "vp_fid", "ve"]
    if "sde" in which:
        golden_sde(HERE)
    if "ops" in which:
        golden_ops(HERE)
    if "vp_nll" in which:
        golden_model(HERE, "mini_vp_nll", "configs.vp.CIFAR10.indm_nll", patch_vp_nll)
    if "vp_fid" in which:
        golden_model(HERE, "mini_vp_fid", "configs.vp.CIFAR10.indm_fid", patch_vp_fid)
    if "ve" in which:
        golden_model(HERE, "mini_ve", "configs.ve.CIFAR10.indm", patch_ve)
    if "extra" in which:
        golden_extra(HERE)
    if "celeba" in which:
        golden_celeba(HERE)
    if "r2" in which:
        golden_round2(HERE)
    if "r3" in which:
        golden_round3(HERE)
    if "r4" in which:
        golden_round4(HERE)
