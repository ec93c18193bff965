"""Adaptive Dormand-Prince 5(4) integrator with the state resident in HBM.

The reference integrates the probability-flow ODE with scipy.integrate.solve_ivp(method='RK45')
(likelihood.py:116, sampling.py:603): a float64 NumPy state on the host, i.e. one device->host and one
host->device copy of the whole state per function evaluation (6 per step).  This module restates SciPy's RK45
(scipy/integrate/_ivp/rk.py: rk_step, RungeKutta._step_impl, select_initial_step; scipy 1.15, unpinned in the
reference's requirements.txt:12) on device tensors: the float64 state, the seven stage derivatives and every
linear combination stay on the GPU, and the only thing that crosses PCIe per step attempt is ONE scalar -- the
error norm the step-size controller branches on (three more in the initial-step heuristic).

Same tableau, same controller constants (SAFETY 0.9, MIN_FACTOR 0.2, MAX_FACTOR 10), same first-same-as-last
reuse, same nfev accounting, so the accepted/rejected step sequence follows SciPy's up to rounding of the norms.
The state is a handful of MB (B x 3072 doubles): these are plumbing-sized ATen ops; the cost of an evaluation is
the score network behind `fun`.
"""
import math

import torch

C = (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0)
A = ((),
     (1 / 5,),
     (3 / 40, 9 / 40),
     (44 / 45, -56 / 15, 32 / 9),
     (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
     (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656))
B = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84)
E = (-71 / 57600, 0.0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40)
SAFETY, MIN_FACTOR, MAX_FACTOR = 0.9, 0.2, 10.0
ORDER = 4           # error-estimator order of RK45
ERR_EXP = -1.0 / (ORDER + 1)


def _rms(v):
    """scipy's norm(): ||v||_2 / sqrt(n) -- one scalar to the host"""
    return float(torch.linalg.vector_norm(v)) / math.sqrt(v.numel())


def _initial_step(fun, t0, y0, t_bound, f0, direction, rtol, atol):
    """scipy.integrate._ivp.common.select_initial_step"""
    interval = abs(t_bound - t0)
    if y0.numel() == 0 or interval == 0.0:
        return float("inf") if y0.numel() == 0 else 0.0, 0
    scale = atol + y0.abs() * rtol
    d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = min(h0, interval)
    f1 = fun(t0 + h0 * direction, y0 + (h0 * direction) * f0)
    d2 = _rms((f1 - f0) / scale) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = max(1e-6, h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1.0 / (ORDER + 1))
    return min(100 * h0, h1, interval), 1


def rk45_solve(fun, t0, t_bound, y0, rtol=1e-3, atol=1e-6, max_steps=1_000_000):
    """Integrate dy/dt = fun(t, y) from t0 to t_bound.  y0: 1-D float64 device tensor; fun(t: float, y) -> tensor
    like y.  Returns (y(t_bound), nfev) with solve_ivp's nfev accounting."""
    if y0.dtype != torch.float64:
        raise ValueError("rk45_solve keeps SciPy's float64 state")
    t, y = float(t0), y0
    direction = 1.0 if t_bound >= t0 else -1.0
    f = fun(t, y)
    nfev = 1
    h_abs, n = _initial_step(fun, t, y, t_bound, f, direction, rtol, atol)
    nfev += n
    K = [None] * 7
    for _ in range(max_steps):
        if direction * (t - t_bound) >= 0:
            break
        min_step = 10 * abs(math.nextafter(t, direction * math.inf) - t)
        h_abs = max(h_abs, min_step)
        step_rejected = False
        while True:
            if h_abs < min_step:
                raise RuntimeError("rk45_solve: required step size is less than spacing between numbers")
            h = h_abs * direction
            t_new = t + h
            if direction * (t_new - t_bound) > 0:
                t_new = t_bound
            h = t_new - t
            h_abs = abs(h)
            # rk_step
            K[0] = f
            for s in range(1, 6):
                dy = K[0] * A[s][0]
                for j in range(1, s):
                    dy = dy + K[j] * A[s][j]
                K[s] = fun(t + C[s] * h, y + dy * h)
            acc = K[0] * B[0]
            for j in range(1, 6):
                if B[j] != 0.0:
                    acc = acc + K[j] * B[j]
            y_new = y + acc * h
            f_new = fun(t + h, y_new)
            K[6] = f_new
            nfev += 6
            err = K[0] * E[0]
            for j in range(1, 7):
                if E[j] != 0.0:
                    err = err + K[j] * E[j]
            scale = atol + torch.maximum(y.abs(), y_new.abs()) * rtol
            error_norm = _rms(err * h / scale)            # the ONE scalar that goes to the host per attempt
            if error_norm < 1:
                factor = MAX_FACTOR if error_norm == 0 else min(MAX_FACTOR, SAFETY * error_norm ** ERR_EXP)
                if step_rejected:
                    factor = min(1.0, factor)
                h_abs *= factor
                break
            h_abs *= max(MIN_FACTOR, SAFETY * error_norm ** ERR_EXP)
            step_rejected = True
        t, y, f = t_new, y_new, f_new
    return y, nfev
