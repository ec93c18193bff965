"""Helpers shared by the golden-fixture tests (CPU oracle tests and GPU parity tests)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from indm_amd import configs  # noqa: E402

_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    return _cache[name]


def sub(d, prefix):
    """{key-without-prefix: torch tensor} for all keys under prefix/"""
    p = prefix + "/"
    return {k[len(p):]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith(p)}


def state_dict(d, prefix):
    """The network weights stored under prefix/ (minus the test vectors that share the prefix)."""
    skip = {"t", "out", "gy", "gx", "raw_out"}
    return {k: v for k, v in sub(d, prefix).items() if k not in skip and not k.startswith("grad/")}


def events(d, prefix):
    n = int(d[f"{prefix}/n_events"])
    keys = sorted(k for k in d if k.startswith(f"{prefix}/ev"))
    assert len(keys) == n
    return [(k[len(prefix) + 7:], d[k]) for k in keys]  # "prefix/evNNN_kind"


CFG = {"mini_vp_nll": "vp/CIFAR10/indm_nll", "mini_vp_fid": "vp/CIFAR10/indm_fid", "mini_ve": "ve/CIFAR10/indm",
       "mini_celeba": "vp/CELEBA/indm_fid"}


def mini_config(fixture):
    cfg = configs.mini(configs.get_config(CFG[fixture]))
    if fixture == "mini_celeba":
        cfg.model.ch_mult = (1, 2, 2)   # as in make_golden.golden_celeba: 64 -> 32 -> 16, one attention block
    return cfg


def assert_close(a, b, rtol=1e-4, atol=1e-5, name=""):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (name, tuple(a.shape), tuple(b.shape))
    err = (a - b).abs().max().item() if a.numel() else 0.0
    ref = b.abs().max().item() if b.numel() else 0.0
    assert err <= atol + rtol * ref, f"{name}: max|diff| {err:.3e} vs ref max {ref:.3e} (rtol {rtol}, atol {atol})"


def stat(t):
    t = t.detach().double()
    return torch.tensor([t.sum().item(), t.pow(2).sum().item()], dtype=torch.float64)
