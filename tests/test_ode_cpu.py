"""indm_amd/ode.py restates SciPy's RK45 (the integrator the reference calls: likelihood.py:116, sampling.py:603) on torch
tensors.  On CPU float64 tensors it must reproduce scipy.integrate.solve_ivp(method='RK45') step for step: same number
of function evaluations, same final state to rounding."""
import numpy as np
import pytest
import torch
from scipy import integrate

from indm_amd import ode


def _systems():
    rs = np.random.RandomState(0)
    Amat = rs.randn(12, 12) * 0.7
    w = rs.randn(12)

    def lin(t, y):            # stiff-ish linear + forcing
        return Amat @ y * np.cos(t) + np.sin(3 * t) * w

    def vdp(t, y):            # van der Pol (nonlinear), 6 oscillators
        x, v = y[:6], y[6:]
        return np.concatenate([v, 2.0 * (1 - x ** 2) * v - x])

    return [("linear", lin, rs.randn(12), (0.0, 3.0)), ("van der pol", vdp, rs.randn(12), (0.0, 6.0)),
            ("backward in time", lin, rs.randn(12), (1.0, 1e-3))]


@pytest.mark.parametrize("rtol,atol", [(1e-3, 1e-3), (1e-5, 1e-5), (1e-7, 1e-9)])
def test_rk45_matches_scipy(rtol, atol):
    for name, f, y0, (t0, t1) in _systems():
        sol = integrate.solve_ivp(f, (t0, t1), y0, rtol=rtol, atol=atol, method="RK45")
        calls = [0]

        def ft(t, y):
            calls[0] += 1
            return torch.from_numpy(f(t, y.numpy()))

        y, nfev = ode.rk45_solve(ft, t0, t1, torch.from_numpy(y0.copy()), rtol=rtol, atol=atol)
        assert nfev == sol.nfev == calls[0], (name, nfev, sol.nfev, calls[0])
        assert np.allclose(y.numpy(), sol.y[:, -1], rtol=1e-10, atol=1e-12), (name, y.numpy() - sol.y[:, -1])


def test_rk45_keeps_float64():
    with pytest.raises(ValueError):
        ode.rk45_solve(lambda t, y: y, 0.0, 1.0, torch.zeros(3))
