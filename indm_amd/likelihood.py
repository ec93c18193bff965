"""NELBO and PF-ODE likelihood with the reference's API (likelihood.py:27-283).

    get_elbo_fn(config, sde, inverse_scaler)        -> loss_fn(model, flow_model, batch, logdet=None)
    get_likelihood_fn(config, sde, inverse_scaler)  -> likelihood_fn(model, flow_model, data, logdet, residual, eps_bpd)

The Hutchinson terms are vector-Jacobian products through the NCSN++ data-grad kernels (first-order
autograd over the HIP ops); the flow's eval-mode log-det is the explicit-VJP power series of flow.py.
"""
import numpy as np
import torch

from . import hip, ops, sde_lib
from .flow import flow_forward
from .rng import default_rng
from .sampling import _Stepper, _d, _unwrap
from .sde_lib import VPSDE


def _perturb(sde, batch_nhwc, t, z):
    mc, std = sde.marginal_coeffs(t)
    return ops.sde_perturb(batch_nhwc, z, _d(mc, batch_nhwc), _d(std, batch_nhwc)), mc, std


def get_likelihood_residual_fn(config, sde, stepper, variance="scoreflow", eps_bpd=1e-5):
    """likelihood.py:241-283.  residual_fn(batch_nhwc, rng) -> [B]"""

    def likelihood_residual_fn(batch, rng):
        B = batch.shape[0]
        n_dim = float(np.prod(batch.shape[1:]))
        eps_vec = torch.ones(B) * config.training.truncation_time
        z = rng.randn_image(*_nchw(batch))
        xt, _, std = _perturb(sde, batch, eps_vec, z)
        out, k = stepper.net_out(xt, eps_vec)
        # noise = -std * score = (-std*k) * out
        nk = -std * k
        eps_vec = torch.ones(B) * eps_bpd
        z = rng.randn_image(*_nchw(batch))
        xt, alpha, beta = _perturb(sde, batch, eps_vec, z)     # marginal_prob(ones) -> (alpha, beta) = (mc, std)
        # q_mean = xt/alpha - beta*noise/alpha ; d = batch - q_mean
        q_mean, _ = hip.affine_update(xt, out, None, _d(1.0 / alpha, xt), _d(-beta * nk / alpha, xt),
                                      _d(alpha, xt), want_mean=False)
        diff = hip.axpby(batch.contiguous(), q_mean, 1.0, -1.0)
        sq = hip.rowreduce(diff, None, 1).cpu()
        q_std = beta / alpha                                     # 'scoreflow': beta / mean(alpha)
        std2 = sde.marginal_coeffs(eps_vec)[1]
        p_entropy = n_dim / 2.0 * (np.log(2 * np.pi) + 2 * torch.log(std2) + 1.0)
        q_recon = n_dim / 2.0 * (np.log(2 * np.pi) + 2 * torch.log(q_std)) + 0.5 / (q_std ** 2) * sq
        return q_recon - p_entropy

    return likelihood_residual_fn


def _nchw(x):
    n, h, w, c = x.shape
    return n, c, h, w


def get_elbo_fn(config, sde, inverse_scaler=None, hutchinson_type="Rademacher"):
    """likelihood.py:142-238 -> (nelbo_bpd [B], nelbo_residual_bpd [B]) on the CPU."""

    def loss_fn(model, flow_model, batch, logdet=None, rng=None, stats=None):
        rng = rng or default_rng(batch.device)
        dev = batch.device
        B = batch.shape[0]
        if config.flow.model != "identity":
            with torch.no_grad():
                z_nchw, log_jacob = flow_forward(config, flow_model, batch, reverse=False, rng=rng, stats=stats)
            log_jacob = log_jacob.reshape(B).cpu()
        else:
            z_nchw, log_jacob = batch, torch.zeros(B)
        if logdet is None:
            logdet = torch.zeros(B)
        zb = ops.to_nhwc(z_nchw)
        st = _Stepper(config, sde, model, rng)
        t, Z = sde.get_diffusion_time(config, B, "cpu", sde.eps, importance_sampling=True, rng=rng)
        qt = 1 / sde.T
        noise = rng.randn_image(*_nchw(zb))
        xt, mc, std = _perturb(sde, zb, t, noise)
        xt = xt.detach().requires_grad_(True)
        with torch.enable_grad():
            out, k = st.net_out(xt, t)                     # score = k * out
            g = sde.diffusion(t)
            # a = std*score ; mu = std^2*score - std^2/g^2 * f,  f = -0.5 beta x (VP) | 0 (VE)
            if hutchinson_type == "Rademacher":
                epsilon = rng.rademacher_image(*_nchw(zb))
            else:
                epsilon = rng.randn_image(*_nchw(zb))
            fcoef = (-0.5 * sde.beta_t(t)) if isinstance(sde, VPSDE) else torch.zeros_like(t)
            c_out = (std ** 2) * k
            c_x = -(std ** 2) / (g ** 2) * fcoef
            # <mu, eps> = sum c_out*out*eps + c_x*x*eps  ->  d/dx gives c_out * J^T eps + c_x * eps
            (vjp,) = torch.autograd.grad(out, xt, ops.scale_rows(epsilon, _d(c_out, xt)).detach())
        tr = hip.hutch_div(vjp.contiguous(), epsilon).cpu() + c_x * hip.rowreduce(epsilon, None, 1).cpu()
        Mu = -tr * Z / qt
        a_sq = hip.rowreduce(out.detach(), None, 1).cpu() * (std * k) ** 2
        Nu = -a_sq * Z / 2 / qt
        lp_t = torch.ones_like(t) * sde.T
        lp_z = rng.randn_image(*_nchw(zb))
        yT, _, _ = _perturb(sde, zb, lp_t, lp_z)
        lp = sde_lib.prior_logp(sde, yT).cpu()
        elbos = lp + Mu + Nu + log_jacob
        residual_fn = get_likelihood_residual_fn(config, sde, st, eps_bpd=config.training.truncation_time)
        with torch.no_grad():
            residual_nll = residual_fn(zb, rng)
        elbos_residual = elbos - residual_nll
        D = float(np.prod(list(batch.shape[1:])))
        off = 7.0 - inverse_scaler(-1.0)
        return -(elbos + logdet) / D / np.log(2) + off, -(elbos_residual + logdet) / D / np.log(2) + off

    return loss_fn


def get_div_fn(stepper, sde, on_device=False):
    """drift of the probability-flow ODE and its Hutchinson divergence at (x, t) in one pass
    (likelihood.py:27-38, 62-71).  on_device: the [B] divergence stays in HBM (the on-device RK45 of ode.py)."""

    def drift_div(x, t, epsilon):
        xg = x.detach().requires_grad_(True)
        with torch.enable_grad():
            out, k = stepper.net_out(xg, t)
            g = sde.diffusion(t)
            a = (-0.5 * sde.beta_t(t)) if stepper.vp else torch.zeros_like(t)
            b = -(g ** 2) * k * 0.5
            (vjp,) = torch.autograd.grad(out, xg, ops.scale_rows(epsilon, _d(b, xg)).detach())
        drift = hip.affine_update(x, out.detach(), None, _d(a, x), _d(b, x), _d(b, x), want_mean=False)[0]
        if on_device:
            return drift, hip.hutch_div(vjp.contiguous(), epsilon) + _d(a, x) * hip.rowreduce(epsilon, None, 1)
        div = hip.hutch_div(vjp.contiguous(), epsilon).cpu() + a * hip.rowreduce(epsilon, None, 1).cpu()
        return drift, div

    return drift_div


def get_likelihood_fn(config, sde, inverse_scaler, hutchinson_type="Rademacher", rtol=1e-5, atol=1e-5,
                      method="RK45"):
    """likelihood.py:41-140.  method='RK45' (the reference's only choice) runs the adaptive Dormand-Prince integrator of
    ode.py with its float64 state resident in HBM: no state copy per function evaluation, one scalar to the host per
    step attempt.  Any other method name goes to scipy.integrate.solve_ivp over a host state, as in the reference."""
    from . import ode

    def likelihood_fn(model, flow_model, data, logdet=None, residual=True, eps_bpd=1e-5, rng=None, stats=None):
        rng = rng or default_rng(data.device)
        dev = data.device
        shape = data.shape
        B = shape[0]
        with torch.no_grad():
            st = _Stepper(config, sde, model, rng)
            if hutchinson_type == "Rademacher":
                epsilon = rng.rademacher_image(*shape)
            else:
                epsilon = rng.randn_image(*shape)
            drift_div = get_div_fn(st, sde)
            if config.flow.model != "identity":
                z_nchw, log_jacob = flow_forward(config, flow_model, data, reverse=False, rng=rng, stats=stats)
                log_jacob = log_jacob.reshape(B).cpu()
            else:
                z_nchw, log_jacob = data, torch.zeros(B)
            zb = ops.to_nhwc(z_nchw)

            if residual:
                noise = rng.randn_image(*shape)
                start, _, _ = _perturb(sde, zb, torch.ones(B) * eps_bpd, noise)
            else:
                start = zb
            if method == "RK45":
                # state = [x (NHWC, the layout the kernels use; the solver's norms are permutation-invariant), logp]
                dd = get_div_fn(st, sde, on_device=True)
                nx = start.numel()

                def ode_dev(t, y):
                    drift, div = dd(y[:nx].view(start.shape).float(), torch.ones(B) * t, epsilon)
                    return torch.cat([drift.reshape(-1).double(), div.double()])

                init = torch.cat([start.reshape(-1).double(), torch.zeros(B, dtype=torch.float64, device=dev)])
                yT, nfe = ode.rk45_solve(ode_dev, eps_bpd, sde.T, init, rtol=rtol, atol=atol)
                z = ops.to_nchw(yT[:nx].view(start.shape).float()).cpu()
                delta_logp = yT[nx:].float().cpu()
            else:
                from scipy import integrate

                def ode_func(t, x):
                    sample = ops.to_nhwc(torch.from_numpy(x[:-B].reshape(shape)).float().to(dev))
                    drift, div = drift_div(sample, torch.ones(B) * t, epsilon)
                    return np.concatenate([ops.to_nchw(drift).cpu().numpy().reshape(-1).astype(np.float64),
                                           div.numpy().reshape(-1).astype(np.float64)], axis=0)

                init = np.concatenate([ops.to_nchw(start).cpu().numpy().reshape(-1), np.zeros((B,))], axis=0)
                sol = integrate.solve_ivp(ode_func, (eps_bpd, sde.T), init, rtol=rtol, atol=atol, method=method)
                nfe = sol.nfev
                zp = sol.y[:, -1]
                z = torch.from_numpy(zp[:-B].reshape(shape)).float()
                delta_logp = torch.from_numpy(zp[-B:]).float()
            prior = sde.prior_logp(z)
            if residual:
                residual_fn = get_likelihood_residual_fn(config, sde, st, eps_bpd=eps_bpd)
                delta_logp = delta_logp - residual_fn(zb, rng)
            if logdet is None:
                logdet = torch.zeros(B)
            bpd = -(prior + delta_logp + logdet + log_jacob) / np.log(2)
            bpd = bpd / float(np.prod(shape[1:])) + (7.0 - inverse_scaler(-1.0))
            return bpd, z, nfe

    return likelihood_fn
