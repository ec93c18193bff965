"""Out-of-bounds WRITE detector for the HIP kernels (GPU AddressSanitizer is not available on this pool).

`install()` replaces the name `torch` inside the package's modules with a proxy whose empty / zeros / empty_like /
zeros_like pad every DEVICE allocation with a guard band of sentinel values on both sides and hand the caller the
middle.  `check()` synchronises and verifies every band: a kernel that stores past (or before) the tensor it was given
-- harmless when the neighbouring block happens to be free, a run-to-run difference in the last bits when it is somebody
else's live tensor -- is reported with the allocation site.  Test infrastructure only (tests/test_guard_gpu.py,
tools/)."""
import math
import traceback

import torch as _torch

GUARD_BYTES = 1 << 16   # 64 KiB either side: a stray ROW (2 KiB at 512 channels) lands inside the band too
_registry = []
_installed = {}


def _sentinel(dtype):
    if dtype in (_torch.float32, _torch.float64, _torch.bfloat16, _torch.float16):
        return -1.25e-11
    return 85


def _guarded(shape, dtype, device, zero, site):
    n = int(math.prod(shape))
    item = _torch.empty((), dtype=dtype).element_size()
    g = GUARD_BYTES // item
    raw = _torch.empty(n + 2 * g, dtype=dtype, device=device)
    s = _sentinel(dtype)
    raw[:g].fill_(s)
    raw[g + n:].fill_(s)
    mid = raw[g:g + n]
    if zero:
        mid.zero_()
    _registry.append((raw, n, g, s, site))
    return mid.view(shape)


def _site():
    fr = [f for f in traceback.extract_stack(limit=8)[:-3]]
    return " <- ".join(f"{f.filename.rsplit('/', 1)[-1]}:{f.lineno}:{f.name}" for f in reversed(fr[-4:]))


def _shape_of(size):
    if len(size) == 1 and isinstance(size[0], (tuple, list, _torch.Size)):
        return tuple(int(v) for v in size[0])
    return tuple(int(v) for v in size)


class _Proxy:
    def __init__(self):
        self.__dict__["_t"] = _torch

    def __getattr__(self, name):
        return getattr(_torch, name)

    def _make(self, size, kw, zero):
        dev = kw.get("device", None)
        dev = _torch.device(dev) if dev is not None else _torch.device("cpu")
        plain = set(kw) - {"device", "dtype"}
        if dev.type != "cuda" or plain:
            return (_torch.zeros if zero else _torch.empty)(*size, **kw)
        return _guarded(_shape_of(size), kw.get("dtype", None) or _torch.get_default_dtype(), dev, zero, _site())

    def empty(self, *size, **kw):
        return self._make(size, kw, False)

    def zeros(self, *size, **kw):
        return self._make(size, kw, True)

    def _like(self, x, kw, zero):
        if not x.is_cuda or (set(kw) - {"dtype"}):
            return (_torch.zeros_like if zero else _torch.empty_like)(x, **kw)
        return _guarded(tuple(x.shape), kw.get("dtype", x.dtype), x.device, zero, _site())

    def empty_like(self, x, **kw):
        return self._like(x, kw, False)

    def zeros_like(self, x, **kw):
        return self._like(x, kw, True)


def install():
    """Route the package's own allocations through the guard proxy (idempotent)."""
    import indm_amd.hip, indm_amd.ops, indm_amd.flow, indm_amd.ncsnpp, indm_amd.optim, indm_amd.losses  # noqa: E401
    proxy = _Proxy()
    for m in (indm_amd.hip, indm_amd.ops, indm_amd.flow, indm_amd.ncsnpp, indm_amd.optim, indm_amd.losses):
        if m.__name__ not in _installed and getattr(m, "torch", None) is _torch:
            _installed[m.__name__] = m
            m.torch = proxy
    # the grow-only scratch buffer predates the proxy: drop it so that it is re-allocated with guard bands
    indm_amd.hip._ws_cache.clear()


def uninstall():
    import indm_amd.hip
    for m in _installed.values():
        m.torch = _torch
    _installed.clear()
    indm_amd.hip._ws_cache.clear()
    _registry.clear()


def count():
    return len(_registry)


def check(keep_live=True):
    """-> list of overrun reports (empty = clean).  Allocations are forgotten after the check unless they are the
    scratch buffer (kept: it lives across steps)."""
    _torch.cuda.synchronize()
    import indm_amd.hip
    scratch = {b.untyped_storage().data_ptr() for b in indm_amd.hip._ws_cache.values()}
    bad, keep = [], []
    for raw, n, g, s, site in _registry:
        ref = _torch.full((g,), s, dtype=raw.dtype, device=raw.device)
        lo, hi = raw[:g] != ref, raw[g + n:] != ref
        if lo.any() or hi.any():
            where = []
            if lo.any():
                idx = lo.nonzero().flatten()
                where.append(f"{int(idx.numel())} elements BEFORE the start (nearest {g - int(idx.max())} elements before)")
            if hi.any():
                idx = hi.nonzero().flatten()
                where.append(f"{int(idx.numel())} elements PAST the end (offsets {int(idx.min())}..{int(idx.max())} past, "
                             f"values {raw[g + n:][idx[:4]].tolist()})")
            bad.append(f"{raw.dtype} tensor of {n} elements allocated at {site}: " + "; ".join(where))
            raw[:g].fill_(s)
            raw[g + n:].fill_(s)
        if keep_live and raw.untyped_storage().data_ptr() in scratch:
            keep.append((raw, n, g, s, site))
    _registry[:] = keep
    return bad
