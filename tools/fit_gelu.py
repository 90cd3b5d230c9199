#!/usr/bin/env python
"""Fit of the GELU polynomial used by gcd_amd/csrc/common.h::gelu_fast.

Phi(x) - 1/2 is approximated on [-R, R] (R = 4.5, beyond which the argument is clamped) by the odd
polynomial x * P(u), u = 2 x^2 / R^2 - 1, P of degree 9, fitted in the Chebyshev basis with Lawson's
iteratively re-weighted least squares (-> near-minimax), then converted to monomials in u for Horner
evaluation.  Prints the coefficients (low -> high) and the fp32 evaluation error.
"""
from math import erf

import numpy as np
from numpy.polynomial import chebyshev as Ch


def phi(x):
    return 0.5 * (1 + np.vectorize(erf)(x / np.sqrt(2)))


def main(R=4.5, n=10, N=6000):
    x = np.cos((np.arange(N) + 0.5) / N * np.pi) * R / 2 + R / 2
    g = phi(x) - 0.5
    u = 2 * x * x / (R * R) - 1
    V = Ch.chebvander(u, n - 1) * x[:, None]
    w = np.ones(N)
    for _ in range(200):
        c, *_ = np.linalg.lstsq(V * w[:, None], g * w, rcond=None)
        e = np.abs(V @ c - g)
        w = w * (e / e.max() + 1e-3)
        w /= w.max()
    mono = Ch.cheb2poly(c)
    print("P(u) coefficients, low -> high:")
    print(", ".join(f"{v:.9e}f" for v in mono))
    xx = np.linspace(-8, 8, 400001).astype(np.float32)
    xc = np.clip(xx, -R, R).astype(np.float32)
    uu = ((xc * xc).astype(np.float32) * np.float32(2 / (R * R)) - np.float32(1)).astype(np.float32)
    p = np.full_like(uu, np.float32(mono[-1]))
    for k in range(n - 2, -1, -1):
        p = (p * uu + np.float32(mono[k])).astype(np.float32)
    ph = (np.float32(0.5) + (xc * p).astype(np.float32)).astype(np.float32)
    print("max |Phi error| (fp32 evaluation, incl. clamp):", np.abs(ph - phi(xx.astype(np.float64))).max())
    print("max |gelu error| on [-8, 8]:", np.abs(xx * ph - xx.astype(np.float64) * phi(xx.astype(np.float64))).max())


if __name__ == "__main__":
    main()
