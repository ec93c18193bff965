"""VP / VE SDEs and the reverse-time SDE (mirror of sde_lib.py:7-120, 123-215, 257-350, 469-481).

Split of labour on MI355X: everything that is a function of the per-image time only (a [B] vector:
mean coefficient, std, beta_t, g, discretisation coefficients, importance-sampled t and Z) is evaluated
here with exactly the reference's fp32 op sequence -- on the HOST by default, because values such as
std(1e-5) = sqrt(1 - exp(2*lmc)) sit on an fp32 cancellation cliff and must
round like the reference's; the [B,C,H,W]-sized arithmetic (perturbation, drift, updates, losses) is
done by fused HIP kernels that take these [B] vectors (indm_amd/csrc/sde_optim.hip).
The tensor-returning methods of the reference (`sde`, `marginal_prob`, `discretize`, `prior_logp`,
`prior_sampling`) are kept with the same signatures for drop-in use; they run on whatever device
their inputs live on.
"""
import abc

import numpy as np
import torch

from . import ops


def _b(v, x):
    return v.reshape(-1, *([1] * (x.dim() - 1))).to(x.device)


class SDE(abc.ABC):
    def __init__(self, N):
        super().__init__()
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        pass

    def discretize(self, x, t, next_t=None):  # sde_lib.py:55-72
        dt = 1 / self.N
        drift, diffusion = self.sde(x, t)
        return drift * dt, diffusion * torch.sqrt(torch.tensor(dt, device=t.device))

    def reverse(self, score_fn, probability_flow=False):
        """sde_lib.py:74-120"""
        N, T, sde_fn, discretize_fn = self.N, self.T, self.sde, self.discretize
        pf = 0.5 if probability_flow else 1.0

        class RSDE(self.__class__):
            def __init__(self):
                self.N = N
                self.probability_flow = probability_flow

            @property
            def T(self):
                return T

            def sde(self, x, t):
                drift, diffusion = sde_fn(x, t)
                score = score_fn(x, t)
                drift = drift - _b(diffusion, x) ** 2 * score * pf
                return drift, (0.0 if probability_flow else diffusion)

            def discretize(self, x, t, next_t=None):
                if next_t is None or next_t[0].item() > 0:
                    f, G = discretize_fn(x, t, next_t)
                else:
                    f = torch.zeros(x.shape, device=x.device)
                    _, G = sde_fn(x, t)
                    G = G * torch.sqrt(t - next_t)
                rev_f = f - _b(G, x) ** 2 * score_fn(x, t) * pf
                return rev_f, (torch.zeros_like(G) if probability_flow else G)

        return RSDE()


class VPSDE(SDE):
    """sde_lib.py:123-215"""

    def __init__(self, truncation_time=1e-5, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0, self.beta_1, self.eps, self.N = beta_min, beta_max, truncation_time, N
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1.0 - self.discrete_betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)

    @property
    def T(self):
        return 1

    # ---- [B]-vector schedule (host fp32, reference op order) ----
    def beta_t(self, t):
        return self.beta_0 + t * (self.beta_1 - self.beta_0)

    def marginal_coeffs(self, t):  # :153-157
        lmc = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.exp(lmc), torch.sqrt(1.0 - torch.exp(2.0 * lmc))

    def diffusion(self, t):  # :147-151
        return torch.sqrt(self.beta_t(t))

    def model_labels(self, t):  # models/utils.py:167
        return t * 999

    # ---- reference tensor API ----
    def sde(self, x, t):
        beta_t = self.beta_t(t)
        return -0.5 * _b(beta_t, x) * x, torch.sqrt(beta_t)

    def marginal_prob(self, x, t):
        mc, std = self.marginal_coeffs(t)
        return _b(mc, x) * x, std

    def prior_sampling(self, shape, data_mean=None):
        return torch.randn(*shape) + (0.0 if data_mean is None else data_mean)

    def prior_logp(self, z):  # :165-169
        n = np.prod(z.shape[1:])
        return -n / 2.0 * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.0

    def prior_logp_const(self, n):
        """(additive constant, coefficient of sum z^2) of prior_logp"""
        return -n / 2.0 * np.log(2 * np.pi), -0.5

    def discretize_coeffs(self, t, next_t=None):
        """f = (a - 1) * x, G   (:171-184)"""
        if next_t is None:
            ts = (t * (self.N - 1) / self.T).long()
            beta, alpha = self.discrete_betas[ts], self.alphas[ts]
            return torch.sqrt(alpha), torch.sqrt(beta)
        G = torch.sqrt((t - next_t) * (self.beta_0 + (self.beta_1 - self.beta_0) * t))
        return torch.sqrt(1.0 - G ** 2), G

    def discretize(self, x, t, next_t=None):
        a, G = self.discretize_coeffs(t.cpu(), None if next_t is None else next_t.cpu())
        return _b(a, x) * x - x, G.to(x.device)

    def integral_beta(self, t):
        return 0.5 * t ** 2 * (self.beta_1 - self.beta_0) + t * self.beta_0

    def antiderivative(self, t, stabilizing_constant=0.0):
        if isinstance(t, (float, int)):
            t = torch.tensor(t).float()
        return torch.log(1.0 - torch.exp(-self.integral_beta(t)) + stabilizing_constant) + self.integral_beta(t)

    def normalizing_constant(self, t_min):
        return self.antiderivative(self.T) - self.antiderivative(t_min)

    def get_diffusion_time(self, config, batch_size, batch_device, t_min, importance_sampling=None, rng=None):
        """:197-206.  Returns HOST tensors (t [B], Z)."""
        if importance_sampling is None:
            importance_sampling = config.training.importance_sampling
        u = rng.rand_host(batch_size) if rng is not None else torch.rand(batch_size)
        if importance_sampling:
            Z = self.normalizing_constant(t_min)
            t = (-self.beta_0 + torch.sqrt(self.beta_0 ** 2 + 2 * (self.beta_1 - self.beta_0) * torch.log(
                1.0 + torch.exp(Z * u + self.antiderivative(t_min))))) / (self.beta_1 - self.beta_0)
            return t, Z.detach()
        return u * (self.T - t_min) + t_min, 1

    def get_t_min(self, config, st=False, rng=None):  # :208-215
        """rng: a noise source with np_rand() (indm_amd.rng); None = numpy's global generator, as the reference"""
        if st:
            u = rng.np_rand() if rng is not None and hasattr(rng, "np_rand") else np.random.rand()
            if config.training.k == 1.0:
                return self.eps ** (1.0 - u)
            return self.eps / (1.0 - u * (1 - self.eps ** (config.training.k - 1))) ** (1.0 / (config.training.k - 1))
        return self.eps


class VESDE(SDE):
    """sde_lib.py:257-350"""

    def __init__(self, truncation_time=1e-5, sigma_min=0.01, sigma_max=50, N=1000):
        super().__init__(N)
        self.sigma_min, self.sigma_max, self.eps, self.N = sigma_min, sigma_max, truncation_time, N
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), N))

    @property
    def T(self):
        return 1

    def marginal_coeffs(self, t):
        return torch.ones_like(t), self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def diffusion(self, t):
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        return sigma * torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min)), device=t.device))

    def model_labels(self, t):  # models/utils.py:183
        return self.marginal_coeffs(t)[1]

    def sde(self, x, t):
        return torch.zeros_like(x), self.diffusion(t)

    def marginal_prob(self, x, t):
        return x, self.marginal_coeffs(t)[1]

    def prior_sampling(self, shape, data_mean=None):
        return torch.randn(*shape) * self.sigma_max + (0.0 if data_mean is None else data_mean)

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2.0 * np.log(2 * np.pi * self.sigma_max ** 2) - torch.sum(z ** 2, dim=(1, 2, 3)) / (
            2 * self.sigma_max ** 2)

    def prior_logp_const(self, n):
        return -n / 2.0 * np.log(2 * np.pi * self.sigma_max ** 2), -1.0 / (2 * self.sigma_max ** 2)

    def discretize_coeffs(self, t, next_t=None):
        if next_t is None:
            ts = (t * (self.N - 1) / self.T).long()
            sigma = self.discrete_sigmas[ts]
            adj = torch.where(ts == 0, torch.zeros_like(t), self.discrete_sigmas[ts - 1])
            return torch.ones_like(t), torch.sqrt(sigma ** 2 - adj ** 2)
        s0, s1 = self.marginal_coeffs(t)[1], self.marginal_coeffs(next_t)[1]
        return torch.ones_like(t), torch.sqrt(s0 ** 2 - s1 ** 2)

    def discretize(self, x, t, next_t=None):
        _, G = self.discretize_coeffs(t.cpu(), None if next_t is None else next_t.cpu())
        return torch.zeros_like(x), G.to(x.device)

    def antiderivative(self, t):
        if isinstance(t, (float, int)):
            t = torch.tensor(t).float()
        return 2.0 * torch.log(self.sigma_min * (self.sigma_max / self.sigma_min) ** t)

    def normalizing_constant(self, t_min):
        return self.antiderivative(self.T) - self.antiderivative(t_min)

    def get_diffusion_time(self, config, batch_size, batch_device, t_min, importance_sampling=None, rng=None):
        if importance_sampling is None:
            importance_sampling = config.training.importance_sampling
        u = rng.rand_host(batch_size) if rng is not None else torch.rand(batch_size)
        if importance_sampling:
            Z = self.normalizing_constant(t_min)
            return t_min + ((Z * u) / (2.0 * (np.log(self.sigma_max) - np.log(self.sigma_min)))), Z.detach()
        return u * (self.T - t_min) + t_min, 1

    def get_t_min(self, config, st=False, rng=None):
        return VPSDE.get_t_min(self, config, st, rng)


def get_sde(config):
    """sde_lib.py:469-481"""
    name = config.training.sde.lower()
    if name == "vpsde":
        return VPSDE(truncation_time=config.training.truncation_time, beta_min=config.model.beta_min,
                     beta_max=config.model.beta_max, N=config.model.num_scales)
    if name == "vesde":
        return VESDE(truncation_time=config.training.truncation_time, sigma_min=config.model.sigma_min,
                     sigma_max=config.model.sigma_max, N=config.model.num_scales)
    raise NotImplementedError(f"SDE {config.training.sde} unknown.")


def prior_logp(sde, z):
    """Differentiable sde.prior_logp on the fused row-reduction kernel."""
    n = int(np.prod(z.shape[1:]))
    const, coef = sde.prior_logp_const(n)
    return ops.rowsumsq(z, coef) + const
