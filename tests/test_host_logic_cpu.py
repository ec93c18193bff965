"""Host-logic tests on CPU: the product's module graph / explicit-VJP log-det engine / step functions are run
with the kernel layer replaced by its plain-PyTorch statement (tests/_emu.py via the emulate_hip fixture) and
compared with the REFERENCE's golden outputs.  This validates everything above the C ABI without a GPU; the
`-m gpu` suite repeats the same cases on the real kernels."""
import pytest

import parity_cases as pc


@pytest.mark.parametrize("fx", pc.FIXTURES)
@pytest.mark.parametrize("case", pc.ALL_CASES, ids=lambda f: f.__name__)
def test_case(case, fx, emulate_hip):
    case(fx, "cpu")


def test_pf_ode(emulate_hip):
    pc.case_pf_ode("mini_vp_nll", "cpu")


@pytest.mark.parametrize("case", pc.EXTRA_CASES, ids=lambda f: f.__name__)
def test_extra_case(case, emulate_hip):
    case("cpu")
