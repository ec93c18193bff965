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


def mix64(*vals):
    """splitmix64-style mix of integers -> one 63-bit seed.  Streams keyed by (step, rank) must not collide the way a
    linear combination does (step + 7919*rank reuses rank 0's stream of step s+7919 for rank 1 at step s)."""
    m = (1 << 64) - 1
    x = 0x9E3779B97F4A7C15
    for v in vals:
        x = (x ^ (int(v) & m)) & m
        x = (x + 0x9E3779B97F4A7C15) & m
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
        x = x ^ (x >> 31)
    return x >> 1


_default = {}      # device key -> the process-wide advancing DeviceRng
_default_seed = [None]


def manual_seed(seed):
    """Re-seed the process-wide default noise source (the analogue of torch.manual_seed for the global generator the
    reference draws from)."""
    _default_seed[0] = int(seed)
    _default.clear()


def default_rng(device):
    """The noise source used whenever a caller passes no rng=: ONE advancing DeviceRng per (process, device), like the
    reference's global torch / numpy generators -- two consecutive sampling_fn / nelbo_fn / loss_fn calls therefore
    draw DIFFERENT noise.  Seeded from manual_seed() (else torch.initial_seed()) mixed with the data-parallel rank."""
    device = torch.device(device)
    key = (device.type, device.index)
    r = _default.get(key)
    if r is None:
        base = _default_seed[0] if _default_seed[0] is not None else torch.initial_seed()
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        r = _default[key] = DeviceRng(device, seed=mix64(base, rank, 0x1D))
    return r


class DeviceRng:
    def __init__(self, device, seed=0, poisson_seed=None):
        """seed: noise streams (per rank in data-parallel runs).  poisson_seed: stream of the residual-flow series
        lengths n ~ Poisson(2) (iresblock.py:110-113); data-parallel ranks pass the SAME value so that every rank
        unrolls the same number of vector-Jacobian products per block -- the single-device reference draws one n per
        block for its whole batch, and equal work per rank removes the wait-for-the-slowest at the all-reduce."""
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.cpu_gen = torch.Generator()
        # on the GPU the normals come from the library's own Philox kernel (indm_philox_randn_f32): its {seed, offset}
        # state lives in device memory and advances in stream order, so a captured hipGraph draws fresh noise per replay
        self.philox = None
        self.reseed(seed, poisson_seed)

    def reseed(self, seed, poisson_seed=None):
        """Re-key every stream of this source in place (a training step re-keys ONE resident source by
        (seed, step, rank) instead of building a new one: no generator objects, and the Philox state goes to the
        device through a pinned non-blocking copy into the SAME tensor -- no synchronous upload on the hot path)."""
        seed = int(seed) % (1 << 63)
        self.gen.manual_seed(seed)
        self.cpu_gen.manual_seed((seed + 1) % (1 << 63))
        self.np = np.random.RandomState((seed + 2 if poisson_seed is None else poisson_seed) % (1 << 32))
        self._drop = (seed * 1000003 + 12345) % (1 << 63)
        if self.device.type == "cuda":
            from . import hip
            host = torch.tensor([mix64(seed, 0x5eed) & ((1 << 63) - 1), 0], dtype=torch.int64)
            if self.philox is None:
                self.philox = hip.to_device(host, self.device)
            else:
                self.philox.copy_(host.pin_memory(), non_blocking=True)
        return self

    def randn_image(self, n, c, h, w):
        if self.philox is not None:
            from . import hip
            return hip.philox_randn((n, h, w, c), self.philox)
        return torch.randn((n, h, w, c), device=self.device, generator=self.gen)

    def randn(self, *shape):
        if self.philox is not None:
            from . import hip
            return hip.philox_randn(tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape,
                                    self.philox)
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

    def np_rand(self):
        """one host uniform: the sampled truncation time of training.st (sde_lib.py:208-215 draws it from numpy's
        global generator, the stream the flow's Poisson draws use -- here the source's own numpy stream)"""
        return float(self.np.rand())

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

    def np_rand(self):
        return float(self._next("np_rand"))

    def dropout_seed(self):
        """The parity fixtures are recorded with model.dropout = 0 (masks cannot be replayed by seed): the value is
        never consumed.  NCSNpp raises if a replay is asked to train with dropout > 0."""
        return None

    def done(self):
        return self.i == len(self.events)
