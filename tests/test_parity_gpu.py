"""GPU parity tests proper: the product path on the real HIP kernels (through the C ABI) against the golden
outputs the reference produced for the same weights / inputs / noise.  fp32 tolerances are stated in
tests/parity_cases.py (per-image losses 5e-4 relative, parameters after two optimiser steps 2e-5 relative on
their first two moments)."""
import pytest
import torch

import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fx", pc.FIXTURES)
@pytest.mark.parametrize("case", pc.ALL_CASES, ids=lambda f: f.__name__)
def test_case(case, fx):
    assert torch.cuda.is_available()
    case(fx, "cuda")


def test_pf_ode():
    pc.case_pf_ode("mini_vp_nll", "cuda")


@pytest.mark.parametrize("case", pc.EXTRA_CASES, ids=lambda f: f.__name__)
def test_extra_case(case):
    """sampler variants, the get_bpd protocol, a deeper score net -- on the real kernels."""
    case("cuda")


def test_no_cpu_fallback():
    """The product path must refuse CPU tensors instead of silently falling back."""
    from indm_amd import hip
    with pytest.raises(RuntimeError):
        hip.axpby(torch.zeros(4), None, 1.0, 0.0)
