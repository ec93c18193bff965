"""Samplers with the reference's API (sampling.py:86-133, 186-210, 263-292, 365-456, 547-621).

    get_sampling_fn(config, sde, shape, inverse_scaler, eps) -> sampler(model, flow_model, temperature=1.,
        data_mean=None, final_time=0., before_data=None, sample_dir=None, r=None) -> (before, after, nfe)
(pc_sampler, pc_sampler_search [sampling.pc_denoise], pc_sampler_more_step [sampling.more_step], ode_sampler)

Every predictor / corrector update is ONE fused kernel  x' = a[b]*x + b[b]*net_out + c[b]*z  whose per-image
coefficients come from the host-side SDE schedule (sde_lib.py); the Langevin step size (two batch-mean norms)
is computed on the device without a host round trip.  The state stays NHWC in HBM for the whole loop.
"""
import os

import numpy as np
import torch

from . import hip, ops
from .flow import flow_forward
from .rng import default_rng
from .sde_lib import VESDE, VPSDE


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


class _Stepper:
    """Shared machinery: raw network output + per-image coefficient vectors."""

    def __init__(self, config, sde, model, rng):
        self.cfg, self.sde, self.net, self.rng = config, sde, _unwrap(model), rng
        self.net.eval()
        self.vp = isinstance(sde, VPSDE)
        self.ddpm = self.vp and config.training.ddpm_score

    def net_out(self, x, t):
        """raw model output and the factor k[b] with score = k * out"""
        labels = self.sde.model_labels(t)
        out = self.net(x, labels.to(x.device), nhwc=True)
        return out, self.score_factor(t)

    def em_terms(self, t):
        """The reference's quantities of EulerMaruyamaPredictor.update_fn (sampling.py:191-197, RSDE.sde
        sde_lib.py:96-103) as [B] host vectors: forward drift f = f_lin x, diffusion g, score = k net_out."""
        sde = self.sde
        f_lin = -0.5 * sde.beta_t(t) if self.vp else torch.zeros_like(t)
        return f_lin, sde.diffusion(t), self.score_factor(t)

    def rd_terms(self, t, next_t=None):
        """ReverseDiffusionPredictor.update_fn (sampling.py:205-210, RSDE.discretize sde_lib.py:105-118): discretised
        forward step f = f_lin x, G; and the denoising step to next_t <= 0 of this repo's reference (sampling.py:402-408)"""
        sde = self.sde
        if next_t is None or next_t[0].item() > 0:
            a_f, G = sde.discretize_coeffs(t, next_t)      # f = (a_f - 1) x
            f_lin = a_f - 1.0
        else:
            f_lin = torch.zeros_like(t)
            G = sde.diffusion(t) * torch.sqrt(t - next_t)
        return f_lin, G, self.score_factor(t)

    def score_factor(self, t):
        """k[b] with score = k * raw network output (models/utils.py:167-177)"""
        return -1.0 / self.sde.marginal_coeffs(t)[1] if self.ddpm else torch.ones_like(t)

    def _noise(self, x):
        """-> (z, philox): the device-resident Philox state when the source has one (the update kernel draws the noise
        itself), else an explicit tensor from the source (replayed / CPU-generated noise)"""
        ph = getattr(self.rng, "philox", None)
        if ph is not None and x.is_cuda:
            return None, ph
        return self.rng.randn_image(*_nchw(x)), None

    def em(self, x, t, probability_flow=False):
        """probability_flow: Predictor.__init__ builds rsde = sde.reverse(score_fn, probability_flow) (sampling.py:139-144).
        With it the reference's RSDE.sde returns the float 0. as diffusion (sde_lib.py:103), which
        EulerMaruyamaPredictor.update_fn indexes (sampling.py:196): the reference RAISES TypeError for this setting
        (golden mini_r4.npz: pf_em/raises), and so does this mirror -- indm_pc_step_em_f32 itself implements the flag."""
        if probability_flow:
            raise TypeError("'float' object is not subscriptable (EulerMaruyamaPredictor with probability_flow=True: "
                            "the reference indexes RSDE.sde's float diffusion, sampling.py:196)")
        out, _ = self.net_out(x, t)
        f_lin, g, k = self.em_terms(t)
        z, ph = self._noise(x)   # drawn by the reference's probability-flow step too (multiplied by a zero diffusion)
        return hip.pc_step_em(x, out, _d(f_lin, x), _d(g, x), _d(k, x), -1.0 / self.sde.N, probability_flow, z, ph)

    def rd(self, x, t, next_t=None, probability_flow=False):
        out, _ = self.net_out(x, t)
        f_lin, G, k = self.rd_terms(t, next_t)
        z, ph = self._noise(x)   # drawn (and discarded, G = 0) by the reference's probability-flow step too
        return hip.pc_step_rd(x, out, _d(f_lin, x), _d(G, x), _d(k, x), probability_flow, z, ph)

    def langevin(self, x, t, snr, n_steps):
        """LangevinCorrector.update_fn sampling.py:272-292"""
        sde = self.sde
        if self.vp:
            alpha = sde.alphas[(t * (sde.N - 1) / sde.T).long()]
        else:
            alpha = torch.ones_like(t)
        alpha = _d(alpha, x)
        x_mean = x
        for _ in range(n_steps):
            out, k = self.net_out(x, t)
            z, ph = self._noise(x)
            x, x_mean = hip.pc_step_langevin(x, out, _d(k, x), alpha, float(snr), z, ph)
        return x, x_mean


def predictor_table(st, predictor, timesteps, B):
    """Everything the predictor steps need from the host, for ALL steps, as one [steps, 4, B] tensor: network labels and
    the three per-image vectors of the update (f_lin, g | G, k)."""
    rows = []
    for tt in timesteps:
        vec_t = torch.ones(B) * tt
        terms = st.em_terms(vec_t) if predictor == "euler_maruyama" else st.rd_terms(vec_t, None)
        rows.append(torch.stack([st.sde.model_labels(vec_t).float()] + [v.float() for v in terms]))
    return torch.stack(rows)


class _GraphedPredictor:
    """The predictor-only sampler step (score network forward + one indm_pc_step_* launch that draws its own Philox noise) captured ONCE as a
    hipGraph and replayed for every step: its launch sequence is static (unlike the training step, whose Poisson
    series lengths change per block per step).  Everything a step needs from the host -- network labels and the three
    coefficient vectors -- is precomputed for ALL steps into one device table [steps, 4, B]; a replay is preceded only
    by a device-to-device copy of that step's row into the graph's static input.  The state x lives in a static
    buffer the graph updates in place; the noise comes from the device-resident Philox state, which the captured
    launch advances itself.  The first step runs eagerly (it readies workspaces / kernel attributes), the second is
    captured, all later ones are replays: the kernels and their order are exactly the eager path's."""

    def __init__(self, st, x, table, use_graph=True, predictor="euler_maruyama", probability_flow=False):
        self.st, self.table, self.use_graph = st, table, use_graph
        self.predictor, self.pf = predictor, probability_flow
        if predictor == "euler_maruyama" and probability_flow:
            raise TypeError("'float' object is not subscriptable (EulerMaruyamaPredictor with probability_flow=True, "
                            "sampling.py:196)")
        self.x = x.clone()
        self.row = torch.empty_like(table[0])
        self.graph, self.x_mean, self.steps_done = None, None, 0
        self._keep = None   # workspace buffers whose addresses the captured launches hold

    def _body(self):
        out = self.st.net(self.x, self.row[0], nhwc=True)
        ph = self.st.rng.philox
        if self.predictor == "euler_maruyama":
            x_new, self.x_mean = hip.pc_step_em(self.x, out, self.row[1], self.row[2], self.row[3], -1.0 / self.st.sde.N,
                                                self.pf, None, ph)
        else:
            x_new, self.x_mean = hip.pc_step_rd(self.x, out, self.row[1], self.row[2], self.row[3], self.pf, None, ph)
        self.x.copy_(x_new)

    def step(self, i):
        self.row.copy_(self.table[i])
        if self.steps_done == 0 or not self.use_graph:
            self._body()
        elif self.graph is None:
            torch.cuda.synchronize()
            self._keep = list(hip._ws_cache.values())
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body()
            self.graph.replay()
        else:
            self.graph.replay()
        self.steps_done += 1
        return self.x, self.x_mean


def _nchw(x):
    n, h, w, c = x.shape
    return n, c, h, w


def _d(v, x):
    return v.float().contiguous().to(x.device)


def get_pc_sampler(config, sde, shape, predictor, corrector, inverse_scaler, snr, n_steps=1,
                   probability_flow=False, continuous=False, denoise=True, eps=1e-3, device="cuda"):
    """sampling.py:365-545.  predictor in {'euler_maruyama','reverse_diffusion','none'}, corrector in
    {'langevin','none'} (the registries of sampling.py:36-83 reduced to what the configs select).  Returns
    pc_sampler_search when sampling.pc_denoise, pc_sampler_more_step when sampling.more_step, else pc_sampler
    (sampling.py:538-545)."""
    B, C, H, W = shape

    def prior(rng, data_mean):
        x = rng.randn_image(B, C, H, W)                       # sde.prior_sampling (sde_lib.py:159-163/289-293)
        if isinstance(sde, VESDE):
            x = hip.axpby(x, None, float(sde.sigma_max), 0.0, float(data_mean or 0.0))
        elif data_mean:
            x = hip.axpby(x, None, 1.0, 0.0, float(data_mean))
        return x

    def corrector_update(st, x, vec_t, snr_):
        if corrector == "langevin":
            return st.langevin(x, vec_t, snr_, n_steps)
        return x, x

    def predictor_update(st, x, vec_t, vec_next_t):
        if predictor == "euler_maruyama":
            return st.em(x, vec_t, probability_flow)
        if predictor == "reverse_diffusion":
            return st.rd(x, vec_t, vec_next_t, probability_flow)
        return x, x

    def finish(before_nhwc, flow_model, temperature, rng, iters):
        before = ops.to_nchw(before_nhwc)
        if config.flow.model != "identity":
            after, _ = flow_forward(config, flow_model, hip.axpby(before, None, float(temperature), 0.0),
                                    log_det=None, reverse=True, rng=rng, iters=iters)
        else:
            after = before
        return inverse_scaler(before), inverse_scaler(after), sde.N * (n_steps + 1)

    def given(before_data):
        return ops.to_nhwc(before_data.to(device).float())

    def pc_sampler(model, flow_model, temperature=1.0, data_mean=None, final_time=0.0, before_data=None,
                   sample_dir=None, r=None, rng=None, iters=None, graph=None):
        """sampling.py:410-456.  graph: capture the predictor step as a hipGraph (None = automatically, when the
        sampler is predictor-only, runs on the GPU and draws its noise from the device-side Philox state)."""
        rng = rng or default_rng(device)
        with torch.no_grad():
            num_scales = config.sampling.num_scales if config.sampling.num_scales != sde.N else sde.N
            x = prior(rng, data_mean)
            timesteps = torch.linspace(sde.T, eps, num_scales)
            st = _Stepper(config, sde, model, rng)
            x_mean = x
            graphable = (corrector == "none" and predictor in ("euler_maruyama", "reverse_diffusion") and x.is_cuda
                         and getattr(rng, "philox", None) is not None)
            if graph is None:
                graph = graphable and num_scales >= 8
            elif graph and not graphable:
                raise RuntimeError("pc_sampler(graph=True) needs a predictor-only sampler on the GPU with a DeviceRng")
            gp = None
            if graph:
                gp = _GraphedPredictor(st, x, predictor_table(st, predictor, timesteps, B).to(x.device),
                                       predictor=predictor, probability_flow=probability_flow)
            for i in range(num_scales):
                vec_t = torch.ones(B) * timesteps[i]
                if config.sampling.snr_scheduling == "none":
                    snr_temp = config.sampling.snr
                else:
                    snr_temp = config.sampling.begin_snr + (config.sampling.end_snr - config.sampling.begin_snr) * i / num_scales
                if gp is not None:
                    x, x_mean = gp.step(i)
                else:
                    x, x_mean = corrector_update(st, x, vec_t, snr_temp)
                    x, x_mean = predictor_update(st, x, vec_t, None)
                if i == num_scales - 2 and sample_dir is not None:
                    # the start point pc_sampler_search resumes from (sampling.py:436-445; read back by
                    # sampling_lib.get_samples for VE pc_denoise)
                    snap = inverse_scaler(ops.to_nchw(x_mean)).permute(0, 2, 3, 1).cpu().numpy() * 255.0
                    os.makedirs(sample_dir, exist_ok=True)
                    with open(os.path.join(sample_dir, f"samples_{r}_before_flow_for_search.npz"), "wb") as f:
                        np.savez_compressed(f, samples=snap)
            return finish(x_mean if denoise else x, flow_model, temperature, rng, iters)

    def denoise_update_fn(st, x, final_time):
        """sampling.py:402-408: one reverse-diffusion probability-flow step from eps to final_time"""
        vec_eps = torch.ones(B) * eps
        return st.rd(x, vec_eps, torch.ones_like(vec_eps) * final_time, probability_flow=True)[1]

    def pc_sampler_search(model, flow_model, temperature=1.0, data_mean=None, final_time=0.0, before_data=None,
                          sample_dir=None, r=None, rng=None, iters=None):
        """sampling.py:458-495"""
        rng = rng or default_rng(device)
        with torch.no_grad():
            st = _Stepper(config, sde, model, rng)
            if before_data is None:
                x = prior(rng, data_mean)
                timesteps = torch.linspace(sde.T, eps, sde.N)
                x_mean = x
                for i in range(sde.N - 1):
                    vec_t = torch.ones(B) * timesteps[i]
                    vec_next_t = torch.ones(B) * timesteps[i + 1]
                    x, x_mean = corrector_update(st, x, vec_t, config.sampling.snr)
                    x, x_mean = predictor_update(st, x, vec_t, vec_next_t)
            else:
                x_mean = x = given(before_data)
            if not config.sampling.need_sample:
                x_mean = x = denoise_update_fn(st, x_mean if denoise else x, final_time)
            return finish(x_mean if denoise else x, flow_model, temperature, rng, iters)

    def pc_sampler_more_step(model, flow_model, temperature=1.0, data_mean=None, final_time=0.0, before_data=None,
                             sample_dir=None, r=None, rng=None, iters=None):
        """sampling.py:497-535"""
        rng = rng or default_rng(device)
        with torch.no_grad():
            st = _Stepper(config, sde, model, rng)
            if before_data is None:
                # the reference calls corrector_update_fn without its required `snr` here (sampling.py:512) and
                # would index timesteps[sde.N] two lines above: same error, raised before any work is queued
                raise TypeError("shared_corrector_update_fn() missing 1 required positional argument: 'snr'")
            x_mean = x = given(before_data)
            if not config.sampling.need_sample:
                timesteps = torch.exp(torch.linspace(np.log(1e-3), np.log(1e-5), 100))
                for i in range(100):
                    vec_t = torch.ones(B) * timesteps[i]
                    x, x_mean = corrector_update(st, x, vec_t, config.sampling.snr)
                    x, x_mean = predictor_update(st, x, vec_t, None)
            return finish(x_mean if denoise else x, flow_model, temperature, rng, iters)

    if config.sampling.pc_denoise:
        return pc_sampler_search
    if config.sampling.more_step:
        return pc_sampler_more_step
    return pc_sampler


def get_ode_sampler(config, sde, shape, inverse_scaler, denoise=False, rtol=1e-5, atol=1e-5, method="RK45",
                    eps=1e-3, device="cuda"):
    """sampling.py:547-621: SciPy RK45 over the probability-flow ODE (float64 host state, device RHS)."""
    from scipy import integrate

    def ode_sampler(model, flow_model, temperature=1.0, data_mean=None, final_time=0.0, before_data=None,
                    sample_dir=None, r=None, rng=None, iters=None):
        rng = rng or default_rng(device)
        with torch.no_grad():
            B, C, H, W = shape
            st = _Stepper(config, sde, model, rng)
            x = rng.randn_image(B, C, H, W)
            if isinstance(sde, VESDE):
                x = hip.axpby(x, None, float(sde.sigma_max), 0.0, float(data_mean or 0.0))

            def drift(xd, t):
                """RSDE(probability_flow).sde: f - 0.5 g^2 score  (sde_lib.py:96-103)"""
                out, k = st.net_out(xd, t)
                g = sde.diffusion(t)
                a = -0.5 * sde.beta_t(t) if st.vp else torch.zeros_like(t)
                b = -(g ** 2) * k * 0.5
                return hip.affine_update(xd, out, None, _d(a, xd), _d(b, xd), _d(b, xd), want_mean=False)[0]

            def ode_func(t, xf):
                xd = torch.from_numpy(xf.reshape(B, C, H, W)).float().to(device)
                d = drift(ops.to_nhwc(xd), torch.ones(B) * t)
                return ops.to_nchw(d).cpu().numpy().reshape(-1).astype(np.float64)

            if method == "RK45":   # on-device Dormand-Prince (ode.py): float64 state in HBM, one scalar per step to the host
                from . import ode
                yT, nfev = ode.rk45_solve(lambda t, y: drift(y.view(x.shape).float(), torch.ones(B) * t).reshape(-1).double(),
                                          sde.T, eps, x.reshape(-1).double(), rtol=rtol, atol=atol)
                x = yT.view(x.shape).float()
            else:
                sol = integrate.solve_ivp(ode_func, (sde.T, eps), ops.to_nchw(x).cpu().numpy().reshape(-1),
                                          rtol=rtol, atol=atol, method=method)
                nfev = sol.nfev
                x = ops.to_nhwc(torch.tensor(sol.y[:, -1]).reshape(shape).float().to(device))
            if denoise:  # denoise_update_fn :568-574
                vec_eps = torch.ones(B) * eps
                _, x = st.rd(x, vec_eps, torch.zeros_like(vec_eps), probability_flow=False)
            before = ops.to_nchw(x)
            if config.flow.model != "identity":
                after, _ = flow_forward(config, flow_model, hip.axpby(before, None, float(temperature), 0.0),
                                        log_det=None, reverse=True, rng=rng, iters=iters)
            else:
                after = before
            return inverse_scaler(before), inverse_scaler(after), nfev

    return ode_sampler


def get_sampling_fn(config, sde, shape, inverse_scaler, eps, device=None):
    """sampling.py:86-133"""
    device = device or getattr(config, "device", "cuda")
    name = config.sampling.method.lower()
    if name == "ode":
        return get_ode_sampler(config=config, sde=sde, shape=shape, inverse_scaler=inverse_scaler,
                               denoise=config.sampling.noise_removal, eps=eps, rtol=config.eval.rtol,
                               atol=config.eval.atol, device=device)
    if name == "pc":
        pred, corr = config.sampling.predictor.lower(), config.sampling.corrector.lower()
        if pred not in ("euler_maruyama", "reverse_diffusion", "none") or corr not in ("langevin", "none"):
            raise ValueError(f"predictor {pred} / corrector {corr} not supported")
        return get_pc_sampler(config=config, sde=sde, shape=shape, predictor=pred, corrector=corr,
                              inverse_scaler=inverse_scaler, snr=config.sampling.snr,
                              n_steps=config.sampling.n_steps_each,
                              probability_flow=config.sampling.probability_flow,
                              continuous=config.training.continuous, denoise=config.sampling.noise_removal,
                              eps=eps, device=device)
    raise ValueError(f"Sampler name {name} unknown.")
