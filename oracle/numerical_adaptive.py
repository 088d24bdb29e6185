#!/usr/bin/env python3
"""Adaptive-quadrature restatement of get_coal_ints(::NumericalCoalStyle, ...) -- TEST INFRASTRUCTURE ONLY.

The reference integrates the coalescence integrals of an arbitrary kernel function with nested adaptive
Gauss-Kronrod quadrature, quadgk(...; rtol = 1e-8, maxevals = 1000) (src/Sources/Coalescence.jl:503-622, integrands
:644-708, weighting_fn :624-642).  QuadGK.jl is not vendored and Julia is not installed, so this file restates the same
nested integrals with scipy.integrate.quad (QUADPACK's adaptive Gauss-Kronrod, epsrel = 1e-8), function by function.

It is far too slow for a batch (~1e5 density evaluations per integral, ~50 integrals per parcel) and exists for one
purpose: to measure the DISCRETISATION error of the fixed Gauss rule that the HIP quadrature-kernel plans and
oracle/cloudy_oracle_quad.c share.  `python oracle/numerical_adaptive.py` regenerates
tests/golden/numerical_adaptive.json (a few parcels; runs in this build container only: scipy.integrate).
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RTOL = 1e-8


def _quad(f, a, b, points=None):
    from scipy.integrate import quad

    return quad(f, a, b, epsrel=RTOL, epsabs=0.0, limit=400, points=points)[0]


# ---- densities, ParticleDistributions.jl:323-388 (Gamma / Exponential / Lognormal) ---------------------------
def density(d, x):
    t, n, th, k = d
    return n * normed_density(d, x)


def normed_density(d, x):
    t, n, th, k = d
    if x <= 0.0:
        return 0.0
    if t == 0:    # Exponential
        return math.exp(-x / th) / th
    if t == 1:    # Gamma: x^(k-1) / theta^k / Gamma(k) * exp(-x/theta), evaluated in logs against overflow
        return math.exp((k - 1.0) * math.log(x) - k * math.log(th) - math.lgamma(k) - x / th)
    if t == 3:    # Lognormal (theta = mu, k = sigma)
        l = math.log(x) - th
        return math.exp(-(l * l) / (2.0 * k * k)) / (x * k * math.sqrt(2.0 * math.pi))
    raise TypeError("no method normed_density_func for this distribution")


def _scale(d):
    """a length scale of the density (mean mass), to place the break points of the adaptive rule"""
    t, n, th, k = d
    return th if t == 0 else th * k if t == 1 else math.exp(th + 0.5 * k * k)


# ---- KernelFunctions.jl:94-116 ----------------------------------------------------------------------------------
def kernel(kf, x, y):
    kind, p = kf
    if kind == 0:
        return p[0]
    if kind == 1:
        return p[0] * (x + y)
    if kind == 2:
        r1 = (3.0 / 4.0 / math.pi * x) ** (1.0 / 3.0)
        r2 = (3.0 / 4.0 / math.pi * y) ** (1.0 / 3.0)
        return p[0] * (r1 + r2) ** 2 * abs(math.pi * r1 * r1 - math.pi * r2 * r2)
    if kind == 3:
        if x < p[0] and y < p[0]:
            return p[1] * (x * x + y * y)
        return p[2] * (x + y)
    raise ValueError(kind)


# ---- Coalescence.jl:624-642 -------------------------------------------------------------------------------------
def weighting_fn(x, k, pdists):
    if k > len(pdists):
        raise AssertionError("k out of range")
    denom = sum(normed_density(d, x) for d in pdists)
    num = sum(normed_density(d, x) for d in pdists[:k])
    return 0.0 if denom == 0.0 else num / denom


def _outer(f, scales):
    """int_0^inf f: split at multiples of the density scales so the adaptive rule sees the structure"""
    pts = sorted({s * m for s in scales for m in (0.05, 0.5, 2.0, 8.0, 40.0)})
    total, a = 0.0, 0.0
    for b in pts:
        total += _quad(f, a, b)
        a = b
    from scipy.integrate import quad

    total += quad(f, a, np.inf, epsrel=RTOL, epsabs=0.0, limit=400)[0]
    return total


# ---- integrands, Coalescence.jl:644-708 -------------------------------------------------------------------------
def q_integrand_outer(x, j, k, kf, pdists, m):
    inner = lambda y: 0.5 * kernel(kf, x - y, y) * (density(pdists[j], x - y) * density(pdists[k], y)
                                                    + density(pdists[k], x - y) * density(pdists[j], y))
    return x ** m * _quad(inner, 0.0, x, points=[0.5 * x])


def r_integrand_outer(x, j, k, kf, pdists, m):
    inner = lambda y: kernel(kf, x, y) * density(pdists[k], x) * density(pdists[j], y)
    return x ** m * _outer(inner, [_scale(pdists[j]), x])


def s_integrand_inner(x, k, kf, pdists, m):
    inner = lambda y: 0.5 * kernel(kf, x - y, y) * density(pdists[k], x - y) * density(pdists[k], y)
    return x ** m * _quad(inner, 0.0, x, points=[0.5 * x])


def get_coal_ints_numerical(pdists, kf):
    """Coalescence.jl:470-489 with the Q/R/S matrices of :503-622 (0-based indices here)."""
    N = len(pdists)
    np_ = [2 if d[0] in (0, 2) else 3 for d in pdists]
    orders = max(np_)
    sc = [_scale(d) for d in pdists]
    Q = np.zeros((orders, N, N))
    R = np.zeros((orders, N, N))
    S = np.zeros((orders, 2, N))
    for m in range(orders):
        for k in range(N):
            for j in range(N):
                if not (k <= j or np_[k] <= m):
                    Q[m, j, k] = _outer(lambda x: q_integrand_outer(x, j, k, kf, pdists, m), [sc[j], sc[k]])
                if not np_[k] <= m:
                    R[m, j, k] = _outer(lambda x: r_integrand_outer(x, j, k, kf, pdists, m), [sc[k]])
        for k in range(N):
            zero = (np_[k] <= m and np_[k + 1] <= m) if k < N - 1 else np_[k] <= m
            if zero:
                continue
            S[m, 0, k] = _outer(lambda x: weighting_fn(x, k + 1, pdists) * s_integrand_inner(x, k, kf, pdists, m), sc)
            S[m, 1, k] = _outer(lambda x: (1 - weighting_fn(x, k + 1, pdists)) * s_integrand_inner(x, k, kf, pdists, m), sc)
    out = []
    for k in range(N):
        for m in range(np_[k]):
            v = Q[m, :, k].sum() - R[m, :, k].sum() + S[m, 0, k]
            if k > 0:
                v += S[m, 1, k - 1]
            out.append(v)
    return np.array(out), Q, R, S


CASES = [
    # the reference's own NumericalCoalStyle test configuration (test_Sources_correctness.jl:175-263)
    dict(name="ref_test_3gamma_linear", kf=(1, [1.0]),
         pdists=[(1, 10.0, 10.0, 3.0), (1, 20.0, 100.0, 5.0), (1, 2.0, 500.0, 6.0)]),
    dict(name="3gamma_hydrodynamic", kf=(2, [1e2 * math.pi * 1e6 * (1e-9) ** (4.0 / 3.0)]),   # normalised E = 1e2 pi, norms (1e6, 1e-9)
         pdists=[(1, 120.0, 0.02, 2.5), (1, 3.0, 4.0, 3.0), (1, 0.05, 300.0, 4.0)]),
    dict(name="2gamma_long", kf=(3, [0.5236, 9.44e9 * 1e6 * 1e-18, 5.78 * 1e6 * 1e-9]),       # box_gamma_mixture_long.jl:20, normalised
         pdists=[(1, 100.0, 0.05, 2.0), (1, 1.0, 5.0, 3.0)]),
    dict(name="1gamma_hydrodynamic", kf=(2, [1e2 * math.pi * 1e6 * (1e-9) ** (4.0 / 3.0)]), pdists=[(1, 50.0, 0.3, 1.7)]),
]


def main():
    out = {"_comment": "generated by oracle/numerical_adaptive.py (scipy.integrate.quad, epsrel 1e-8): "
                       "get_coal_ints(::NumericalCoalStyle) of Coalescence.jl:470-708 by nested adaptive quadrature; "
                       "pdists are (type, n, theta, k) in normalised units, kernel (kind, params) normalised",
           "cases": []}
    for c in CASES:
        ci, Q, R, S = get_coal_ints_numerical(c["pdists"], c["kf"])
        out["cases"].append(dict(name=c["name"], kf=list(c["kf"]), pdists=[list(d) for d in c["pdists"]],
                                 coal_ints=ci.tolist(), Q=Q.tolist(), R=R.tolist(), S=S.tolist()))
        print(c["name"], ci, flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "numerical_adaptive.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
