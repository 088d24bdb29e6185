#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (round 6): a THIRD, cheap, independent opinion on EVERY entry of the Q, R and S matrices of
EVERY golden case of tests/golden/numerical_adaptive.json.

The golden values come from the builder's nested adaptive Gauss-Kronrod restatement (oracle/cloudy_oracle_adaptive.c); mpmath
recomputes 25 of the 47 cases (hours per case).  This script integrates the reference's formulas themselves
(src/Sources/Coalescence.jl:644-708, weighting_fn :624-642, densities ParticleDistributions.jl:323-388, kernel functions
KernelFunctions.jl:94-116) with a FIXED composite Gauss-Legendre rule -- no adaptivity, no error estimate, no special function
beyond exp / log / lgamma, nothing shared with the oracle's code:

  R[m][j,k] = int x^m f_k(x) int K(x, y) f_j(y) dy dx                               (:673-691)   a tensor rule in (ln x, ln y)
  Q[m][j,k] = int x^m int_0^x 1/2 K(x-y, y) [f_j(x-y) f_k(y) + f_k(x-y) f_j(y)] dy dx   (:644-671)
            = int int (u + v)^m K(u, v) f_j(u) f_k(v) du dv     (u = x - y, v = y; K symmetric)   the same tensor rule
  S_1/2[m][k] = int x^m {w_k(x), 1 - w_k(x)} int_0^x 1/2 K(x-y, y) f_k(x-y) f_k(y) dy dx        (:693-708)
            the inner integral over y in (0, x/2] (the integrand is symmetric about x/2) in ln y, the outer in ln x

Panel edges: per density a ladder that resolves its core and keeps |d ln f| per panel small out to e^-250 of its peak (uniform in
ln x through the bulk of a Gamma density and LINEAR in x along its exponential tail; uniform in (ln x - mu) / sigma for a
Lognormal density); the kernel's non-smooth sets are panel edges (Long: x_t, 2 x_t, and y = x - x_t in the inner rule) or are
integrated on the two triangles of the cells they cross (hydrodynamic: the diagonal u = v, by a Duffy map).  12 Gauss-Legendre
points per panel.  Entries are compared RELATIVELY, entry by entry (entries below 1e-30 of the largest of their matrix against
that floor); the worst figure of a case is stored in the golden file as `gl_max_rel_diff`.

usage: python oracle/gl_check.py [--write] [case-name ...]        (all cases: ~20 min on 8 cores)"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "numerical_adaptive.json")
NG = 12
GX, GW = np.polynomial.legendre.leggauss(NG)
LN_CUT = 250.0   # a density is followed down to e^-250 of its peak


# ---- densities n g(x) (ParticleDistributions.jl:323-388) and normed densities g(x) (:363-388), through logarithms ----------------
def ln_normed(d, x, lx):
    t, n, th, k = d
    if t == 0:
        return -x / th - math.log(th)
    if t == 1:
        return (k - 1.0) * lx - x / th - k * math.log(th) - math.lgamma(k)
    return -(lx - th) ** 2 / (2.0 * k * k) - lx - math.log(k * math.sqrt(2.0 * math.pi))


def dens(d, x, lx):
    return d[1] * np.exp(ln_normed(d, x, lx))


def kernel(kind, prm, x, y):   # KernelFunctions.jl:94-116
    if kind == 0:
        return prm[0] + 0.0 * (x + y)
    if kind == 1:
        return prm[0] * (x + y)
    if kind == 2:
        r1, r2 = np.cbrt(3.0 / 4.0 / math.pi * x), np.cbrt(3.0 / 4.0 / math.pi * y)
        return prm[0] * (r1 + r2) ** 2 * np.abs(math.pi * r1 * r1 - math.pi * r2 * r2)
    return np.where((x < prm[0]) & (y < prm[0]), prm[1] * (x * x + y * y), prm[2] * (x + y))


def weighting_fn(pd, k, x, lx):   # weighting_fn(x, k, pdists), :624-642 (k 1-based: modes 1..k in the numerator)
    lg = np.stack([ln_normed(d, x, lx) + 0.0 * x for d in pd])
    mx = lg.max(axis=0)
    g = np.exp(lg - mx)
    den = g.sum(axis=0)
    return g[:k].sum(axis=0) / den   # (den > 0: the largest term is 1)


# ---- panel edges in ln x ------------------------------------------------------------------------------------------------------
def support(d):
    """[ln x_lo, ln x_hi]: above, n g(x) x^3 is below e^-250 of its peak; below, n g(x) x is below e^-60 of it"""
    t, n, th, k = d
    if t == 3:
        s = k
        return th - 23.0 * s, th + 23.0 * s + 4.0 * s * s
    kk = 1.0 if t == 0 else k
    hi = math.log(th * (kk + 2.0 + LN_CUT + (kk + 2.0) * math.log((kk + 2.0 + LN_CUT) / (kk + 2.0)) + 10.0))
    lo = math.log(th * kk) + min(-4.0, -60.0 / kk)
    return lo, hi


def ladder(d):
    """edges in ln x that resolve density d: every panel has a modest change of ln f"""
    t, n, th, k = d
    lo, hi = support(d)
    if t == 3:
        return np.arange(lo, hi + 1e-12, k / 3.0)
    kk = 1.0 if t == 0 else k
    core = math.log(th * kk) - 4.0                                  # below: x g(x) ~ x^k, a pure exponential in ln x
    knee = math.log(th * max(kk, 1.0)) + 1.5                       # ~4.5 x the mean: the exponential tail takes over
    low = np.arange(lo, core, 0.5)
    bulk = np.arange(core, knee, 0.2 / max(1.0, math.sqrt(kk) / 2.0))
    tail = np.log(np.arange(math.exp(knee), math.exp(hi), 2.5 * th))   # linear spacing of 2.5 theta: |d ln f| = 2.5 per panel
    return np.concatenate([low, bulk, tail, [hi]])


def merge(edge_lists, lo, hi, extra=(), hmax=0.5, tol=1e-9):
    e = np.concatenate([np.asarray(x, dtype=float).ravel() for x in edge_lists] + [np.asarray(list(extra), dtype=float), [lo, hi]])
    e = np.unique(e[(e >= lo) & (e <= hi)])
    e = e[np.concatenate([[True], np.diff(e) > tol])]
    nsub = np.maximum(np.ceil(np.diff(e) / hmax), 1).astype(int)    # no panel wider than hmax (power-law factors of the kernel, x^m)
    if (nsub > 1).any():
        parts = [a + (b - a) * np.arange(n) / n for a, b, n in zip(e[:-1], e[1:], nsub)]
        e = np.concatenate(parts + [[e[-1]]])
    return e


def nodes(edges):
    a, b = edges[:-1], edges[1:]
    h = 0.5 * (b - a)
    t = (0.5 * (a + b))[:, None] + h[:, None] * GX[None, :]
    w = h[:, None] * GW[None, :]
    return t.ravel(), w.ravel()


# ---- R and Q: tensor rules over (ln u, ln v) ------------------------------------------------------------------------------------
def pair_integrals(kind, prm, du, dv, orders, weight):
    """int int W_m(u, v) K(u, v) f_du(u) f_dv(v) du dv for m < orders; weight(m, U, V) = u^m (R) or (u + v)^m (Q)"""
    lo_u, hi_u = support(du)
    lo_v, hi_v = support(dv)
    extra = [math.log(prm[0])] if kind == 3 else []
    # ONE edge list for both variables (cells on the diagonal are then squares: the hydrodynamic kink u = v crosses no other cell);
    # each variable walks the panels that meet its density's support
    E = merge([ladder(du), ladder(dv)], min(lo_u, lo_v), max(hi_u, hi_v), extra)
    pu = np.flatnonzero((E[1:] > lo_u) & (E[:-1] < hi_u))
    pv = np.flatnonzero((E[1:] > lo_v) & (E[:-1] < hi_v))
    eu, ev = E[pu[0]:pu[-1] + 2], E[pv[0]:pv[-1] + 2]
    tu, wu = nodes(eu)
    tv, wv = nodes(ev)
    U, V = np.exp(tu), np.exp(tv)
    fu = dens(du, U, tu) * U * wu          # du = u d(ln u)
    fv = dens(dv, V, tv) * V * wv
    out = np.zeros(orders)
    diag = []
    if kind == 2:   # panels both variables walk: their square cells hold the kink and are done on their two triangles below
        for p_ in np.intersect1d(pu, pv):
            diag.append((int(p_ - pu[0]), int(p_ - pv[0]), E[p_], E[p_ + 1]))
    blk = 256
    for i0 in range(0, U.size, blk):
        Ub, fb = U[i0:i0 + blk, None], fu[i0:i0 + blk, None]
        Kb = kernel(kind, prm, Ub, V[None, :]) * fb * fv[None, :]
        for m in range(orders):
            out[m] += (Kb * weight(m, Ub, V[None, :])).sum()
    for iu, jv, a, b in diag:
        su_, sv_ = slice(iu * NG, (iu + 1) * NG), slice(jv * NG, (jv + 1) * NG)
        Kc = kernel(kind, prm, U[su_, None], V[None, sv_]) * fu[su_, None] * fv[None, sv_]
        h = b - a
        al, wa = 0.5 * (GX + 1.0), 0.5 * GW          # Duffy: t_hi = a + h alpha, t_lo = a + h alpha beta, Jacobian h^2 alpha
        A, B = np.meshgrid(al, al, indexing="ij")
        WA = np.outer(wa, wa) * (h * h) * A
        thi, tlo = a + h * A, a + h * A * B
        for m in range(orders):
            out[m] -= (Kc * weight(m, U[su_, None], V[None, sv_])).sum()
            for (t1, t2) in ((thi, tlo), (tlo, thi)):   # u > v and u < v
                X, Y = np.exp(t1), np.exp(t2)
                out[m] += (WA * kernel(kind, prm, X, Y) * dens(du, X, t1) * X * dens(dv, Y, t2) * Y * weight(m, X, Y)).sum()
    return out


# ---- S: outer in ln x, inner over y in (0, x/2] in ln y -------------------------------------------------------------------------
def s_integrals(kind, prm, pd, k, orders):
    d = pd[k]
    lo, hi = support(d)
    hi_x = hi + math.log(2.0)
    extra = [math.log(prm[0]), math.log(2.0 * prm[0])] if kind == 3 else []
    if kind == 3:
        # the inner integral behaves like (x - x_t)^k just above x_t (the branch of the kernel changes on y < x - x_t, and the
        # density is y^(k-1) there): an algebraic end point of the OUTER variable -- panels graded geometrically towards both
        # sides of x_t and 2 x_t (a shape of 0.75 left 1e-8 in S_2 of 3gamma_long_scales_apart without them)
        for c0 in (prm[0], 2.0 * prm[0]):
            extra += [math.log(c0 * (1.0 + sg * 2.0 ** -i)) for i in range(1, 48) for sg in (1.0, -1.0)]
    lads = [ladder(q) for q in pd]                                   # weighting_fn turns where the densities cross
    if d[0] == 3:                                                    # the self-sum of a Lognormal mode: a peak near 2 e^mu
        lads.append(d[2] + math.log(2.0) + np.arange(-23.0, 23.01, 1.0 / 3.0) * d[3] / math.sqrt(2.0))
    ex = merge(lads, lo, hi_x, extra, hmax=0.25)
    tx, wx = nodes(ex)
    own = ladder(d)
    S1, S2 = np.zeros(orders), np.zeros(orders)
    for t, w in zip(tx, wx):
        x = math.exp(t)
        lx2 = math.log(0.5 * x)
        if lx2 <= lo:
            continue
        # inner edges: the mode's own ladder in y, and in x - y (mapped), the Long jump at y = x - x_t
        inner = [own]
        big = np.exp(own)
        m_ = x - big
        inner.append(np.log(m_[(m_ > 0)]))
        ext = []
        if kind == 3 and prm[0] < x < 2.0 * prm[0]:
            ext.append(math.log(x - prm[0]))
        ey = merge(inner, lo, lx2, ext, hmax=0.5)
        ty, wy = nodes(ey)
        Y = np.exp(ty)
        XmY = x - Y
        val = (0.5 * kernel(kind, prm, XmY, Y) * dens(d, XmY, np.log(XmY)) * dens(d, Y, ty) * Y * wy).sum() * 2.0   # (0, x/2] twice
        wk = float(weighting_fn(pd, k + 1, np.array([x]), np.array([t]))[0])
        # 1 - w from the densities of the modes ABOVE k (no cancellation where w is within rounding of 1)
        lg = np.array([ln_normed(q, x, t) for q in pd])
        g = np.exp(lg - lg.max())
        omw = float(g[k + 1:].sum() / g.sum())
        for m in range(orders):
            c = w * x * x ** m * val
            S1[m] += wk * c
            S2[m] += omw * c
    return S1, S2


def matrices(case):
    pd = [tuple(d) for d in case["pdists"]]
    kind, prm = case["kf"][0], list(case["kf"][1])
    N = len(pd)
    npm = [2 if int(d[0]) == 0 else 3 for d in pd]
    orders = max(npm)
    Q, R, S = np.zeros((orders, N, N)), np.zeros((orders, N, N)), np.zeros((orders, 2, N))
    for k in range(N):
        for j in range(N):
            r = pair_integrals(kind, prm, pd[k], pd[j], orders, lambda m, u, v: u ** m)        # x: mode k, y: mode j
            for m in range(orders):
                if not npm[k] <= m:
                    R[m, j, k] = r[m]
            if k > j:
                q = pair_integrals(kind, prm, pd[j], pd[k], orders, lambda m, u, v: (u + v) ** m)
                for m in range(orders):
                    if not npm[k] <= m:
                        Q[m, j, k] = q[m]
        s1, s2 = s_integrals(kind, prm, pd, k, orders)
        for m in range(orders):
            skip = (npm[k] <= m and npm[k + 1] <= m) if k < N - 1 else npm[k] <= m
            if not skip:
                S[m, 0, k], S[m, 1, k] = s1[m], s2[m]
    return Q, R, S


def rel_diff(a, b, floor_rel=1e-30):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    floor = floor_rel * max(float(np.max(np.abs(b))), 1e-250)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def check_case(c):
    t0 = time.time()
    Q, R, S = matrices(c)
    per = {x: rel_diff(c[x], m) for x, m in (("Q", Q), ("R", R), ("S", S))}
    return dict(name=c["name"], gl_max_rel_diff=max(per.values()), per_matrix=per, seconds=time.time() - t0)


def main():
    write = "--write" in sys.argv
    names = [a for a in sys.argv[1:] if not a.startswith("--")]
    with open(GOLDEN) as f:
        g = json.load(f)
    cases = [c for c in g["cases"] if not names or c["name"] in names]
    from multiprocessing import Pool

    with Pool(min(8, len(cases))) as pool:
        for r in pool.imap_unordered(check_case, cases):
            print(f"{r['name']:36s} gl max rel diff {r['gl_max_rel_diff']:.2e}   (Q {r['per_matrix']['Q']:.1e}  R {r['per_matrix']['R']:.1e}  "
                  f"S {r['per_matrix']['S']:.1e})   {r['seconds']:.0f} s", flush=True)
            if write:
                next(c for c in g["cases"] if c["name"] == r["name"])["gl_max_rel_diff"] = r["gl_max_rel_diff"]
    if write:
        with open(GOLDEN, "w") as f:
            json.dump(g, f, indent=1)


if __name__ == "__main__":
    main()
