"""Coefficients of the GELU polynomial in csrc/common.h (TFIMM_GELU_C0 .. C9).

GELU(v) = v Phi(v).  On the clamped argument t = clip(v, -c, c):  Phi(t) - 1/2 = t P(r),  r = t^2 - c^2 / 2,  and the kernel
computes  v * fma(t, P(r), 0.5).  P is the polynomial of degree n that minimises the maximum ABSOLUTE error of that GELU over
[-c, c] (a linear programme over a dense grid), subject to  c P(c^2 / 2) = 1/2  -- so that the factor is 1 (0) beyond the clamp
and the error there is v (1 - Phi(|v|)) <= c (1 - Phi(c)).  Prints the coefficients and the error of the float32 evaluation.

    python tools/fit_gelu_poly.py [degree=9] [clamp=4.5]
"""
import sys

import numpy as np
from scipy.optimize import linprog
from scipy.special import erf


def phi(v):
    return 0.5 * (1.0 + erf(v / np.sqrt(2.0)))


def fit(n, c, grid=4000):
    k = np.arange(grid)
    t = np.concatenate([c * np.sin(0.5 * np.pi * (k + 0.5) / grid), np.linspace(0.0, c, grid // 2)])
    u = t * t
    s = 2.0 * u / (c * c) - 1.0                       # fit in s in [-1, 1]; r = s c^2 / 2
    a = np.stack([u * s ** j for j in range(n + 1)], axis=1)       # GELU(t) - t / 2 = t^2 P
    b = t * phi(t) - 0.5 * t                                       # (the same equation for -t: the form is odd-symmetric)
    cost = np.zeros(n + 2)
    cost[-1] = 1.0
    one = np.ones((len(t), 1))
    res = linprog(cost, A_ub=np.block([[a, -one], [-a, -one]]), b_ub=np.concatenate([b, -b]),
                  A_eq=np.concatenate([np.ones(n + 1), [0.0]])[None, :], b_eq=[0.5 / c],
                  bounds=[(None, None)] * (n + 1) + [(0, None)], method="highs")
    assert res.status == 0, res.message
    return res.x[:n + 1] * (2.0 / (c * c)) ** np.arange(n + 1), res.x[-1]       # coefficients of r^j


def gelu_poly_f32(coef, v, c):
    """The kernel's arithmetic: float32 with fused multiply-adds."""
    def fma(x, y, z):
        return (x.astype(np.float64) * y.astype(np.float64) + np.float64(z)).astype(np.float32)
    v = np.asarray(v, dtype=np.float32)
    t = np.clip(v, np.float32(-c), np.float32(c))
    r = fma(t, t, np.float32(-c * c / 2))
    g = np.full_like(r, np.float32(coef[-1]))
    for cf in coef[-2::-1]:
        g = fma(g, r, np.float32(cf))
    return v * fma(t, g, np.float32(0.5))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    c = float(sys.argv[2]) if len(sys.argv) > 2 else 4.5
    coef, e = fit(n, c)
    v = np.linspace(-16, 16, 1280001)
    err = np.abs(gelu_poly_f32(coef, v, c).astype(np.float64) - v * phi(v))
    print(f"degree {n}, clamp {c}: minimax error {e:.3e}; float32 evaluation: max |error| {err.max():.3e} at v = {v[err.argmax()]:.3f}")
    for j, x in enumerate(coef):
        print(f"#define TFIMM_GELU_C{j} {np.float32(x):.9e}f")
