"""Noise sources for the INDM step.

The reference draws every random number with torch / numpy global generators in a fixed call order
(SURVEY.md Appendix C).  The product asks an `Rng` object for the same draws in the same order:
  * DeviceRng  -- production: device-side torch generators (images directly in NHWC), host-side
                  numpy for the Poisson series lengths and the [B] diffusion-time uniforms
  * ReplayRng  -- parity tests: replays draws recorded from the reference (NCHW) in order
Image-shaped noise is always requested with the reference's NCHW shape and returned NHWC.
"""
import numpy as np
import torch


class DeviceRng:
    def __init__(self, device, seed=0, poisson_seed=None):
        """seed: noise streams (per rank in data-parallel runs).  poisson_seed: stream of the residual-flow series
        lengths n ~ Poisson(2) (iresblock.py:110-113); data-parallel ranks pass the SAME value so that every rank
        unrolls the same number of vector-Jacobian products per block -- the single-device reference draws one n per
        block for its whole batch, and equal work per rank removes the wait-for-the-slowest at the all-reduce."""
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.cpu_gen = torch.Generator().manual_seed(seed + 1)
        self.np = np.random.RandomState((seed + 2 if poisson_seed is None else poisson_seed) % (1 << 32))
        self._drop = seed * 1000003 + 12345

    def randn_image(self, n, c, h, w):
        return torch.randn((n, h, w, c), device=self.device, generator=self.gen)

    def randn(self, *shape):
        return torch.randn(*shape, device=self.device, generator=self.gen)

    def rand(self, *shape):
        """device uniforms (the dequantisation noise of evaluation.py:410)"""
        return torch.rand(*shape, device=self.device, generator=self.gen)

    def rand_host(self, *shape):
        """[B] uniforms for the diffusion time (sde_lib.py:202); drawn on the host so the fp32 scalar
        schedule can be evaluated there without a device round trip."""
        return torch.rand(*shape, generator=self.cpu_gen)

    def rademacher_image(self, n, c, h, w):
        r = torch.randint(0, 2, (n, h, w, c), device=self.device, generator=self.gen)
        return r.float() * 2 - 1.0

    def poisson(self, lam):
        return int(self.np.poisson(lam, 1)[0])

    def dropout_seed(self):
        self._drop = (self._drop * 6364136223846793005 + 1442695040888963407) % (1 << 63)
        return self._drop


class ReplayRng:
    """events: list of (kind, numpy array) as recorded by tests/golden/make_golden.py."""

    def __init__(self, events, device):
        self.events = list(events)
        self.i = 0
        self.device = torch.device(device)

    def _next(self, kind, shape=None):
        assert self.i < len(self.events), f"replay exhausted at {kind}"
        k, v = self.events[self.i]
        self.i += 1
        assert k == kind, f"replay event {self.i - 1}: recorded {k}, requested {kind}"
        if shape is not None:
            assert tuple(v.shape) == tuple(shape), (k, tuple(v.shape), tuple(shape))
        return torch.as_tensor(np.asarray(v)).clone()

    def randn_image(self, n, c, h, w):
        v = self._next("randn", (n, c, h, w))
        return v.permute(0, 2, 3, 1).contiguous().to(self.device)

    def randn(self, *shape):
        return self._next("randn", shape).to(self.device)

    def rand(self, *shape):
        return self._next("rand", shape).to(self.device)

    def rand_host(self, *shape):
        return self._next("rand", shape)

    def rademacher_image(self, n, c, h, w):
        v = self._next("rademacher", (n, c, h, w))
        return v.permute(0, 2, 3, 1).contiguous().to(self.device)

    def poisson(self, lam):
        return int(self._next("poisson"))

    def dropout_seed(self):
        return 0

    def done(self):
        return self.i == len(self.events)
