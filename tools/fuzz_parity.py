#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: random plans (N, P, closure families, fixed / moving thresholds, sparse symmetric
tensors, plane types) -- HIP (plan-time compiled and ahead-of-time kernels) against the CPU oracle on small batches.
usage: python tools/fuzz_parity.py [--configs 40] [--seed 1] [--parcels 400]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from __graft_entry__ import load_package
from oracle import cloudy_oracle as O

INF = float("inf")


def random_config(rng, wild=False, big=False):
    N = int(rng.integers(1, 5))
    P = int(rng.integers(1, 6))
    if big:  # beyond the ahead-of-time families (CLOUDY_AOT_MAX_MODES = 4, CLOUDY_AOT_MAX_P = 5): plan-time compiled kernels only
        if rng.random() < 0.5:
            N, P = int(rng.integers(5, 9)), int(rng.integers(1, 4))
        else:
            N, P = int(rng.integers(1, 4)), int(rng.integers(6, 9))
    moving = bool(rng.random() < 0.25) and N > 1
    dist = []
    for i in range(N):
        allowed = [0, 1] if moving and i < N - 1 else [0, 1, 1, 1, 2, 3]
        dist.append(int(rng.choice(allowed)))
    if moving:
        thr = tuple(float(rng.choice([0.5, 0.9, 0.97, 0.99])) for _ in range(N - 1)) + (1.0,)
    else:
        thr = []
        for i in range(N):
            finite = i < N - 1 and rng.random() < 0.6  # (any closure family, Lognormal included)
            thr.append(float(10.0 ** rng.uniform(-11.0, -6.5)) if finite else INF)
        thr = tuple(thr)
    kc = np.zeros((N, N, P, P))
    for j in range(N):
        for k in range(j, N):
            c = rng.uniform(0.1, 1.0, (P, P)) * (rng.random((P, P)) < 0.5)
            c = np.triu(c) + np.triu(c, 1).T
            kc[j, k] = kc[k, j] = c * 1e-3 * (1e9 ** np.add.outer(np.arange(P), np.arange(P)))
    dtype = int(rng.choice([0, 0, 0, 1]))
    norms, k_range = bench.NORMS, (float(np.finfo(np.float64).eps), 10.0)
    if wild:  # other unit systems and clamp ranges: the eps rules, the Simpson grids (x_t / m0) and the k clamp move
        norms = (float(10.0 ** rng.uniform(3, 9)), float(10.0 ** rng.uniform(-12, -7)))
        k_range = (float(rng.choice([np.finfo(np.float64).eps, 1e-3, 0.1])), float(rng.choice([5.0, 10.0, 25.0])))
        dtype = 0
    return dict(N=N, P=P, moving=moving, dist=dist, thr=thr, kc=kc, dtype=dtype, norms=norms, k_range=k_range)


def moments_for(dist, n, seed):
    if len(dist) > 4:  # more size classes than bench.synth_moments has: one per mode between 1e-12 and 1e-4 kg
        rng = np.random.Generator(np.random.Philox(key=seed))
        edges = np.logspace(-12, -4, len(dist) + 1)
        full = np.concatenate([bench._gamma_mode(rng, n, 1e6 * 10.0 ** (-12.0 * i / len(dist)), 1e9 * 10.0 ** (-12.0 * i / len(dist)),
                                                 0.5 if i == 0 else 1.0, 7.0, edges[i], edges[i + 1]) for i in range(len(dist))])
        z = rng.choice(n, max(n // 50, 1), replace=False)
        full[:, z[: len(z) // 2]] = 0.0
        full[2::3, z[len(z) // 2:]] = full[1::3, z[len(z) // 2:]] ** 2 / full[0::3, z[len(z) // 2:]]
    else:
        full = bench.synth_moments(len(dist), n, seed)
    rows = []
    for i, t in enumerate(dist):
        rows += [full[3 * i], full[3 * i + 1]] + ([full[3 * i + 2]] if t in (1, 3) else [])
    return np.ascontiguousarray(np.stack(rows))


def run(pkg, plan, mom, tio):
    m = pkg.DeviceArray.from_numpy(mom.astype(tio))
    dm = pkg.DeviceArray.zeros(mom.shape[0], mom.shape[1], tio)
    pkg._lib.check(pkg.lib().cloudy_coal_rhs(plan.handle, mom.shape[1], mom.shape[1], m.ptr, dm.ptr, None))
    return dm.to_numpy().astype(np.float64)


def check_config(pkg, cfg, n, seed):
    """-> (worst |hip - oracle| / scale, worst |jit - aot| / scale); raises AssertionError on a parity failure"""
    N = cfg["N"]
    kernels = tuple(tuple(pkg.CoalescenceTensor(cfg["kc"][j, k]) for k in range(N)) for j in range(N))
    npm = tuple({0: 2, 1: 3, 2: 2, 3: 3}[t] for t in cfg["dist"])
    ts = pkg.MovingThreshold() if cfg["moving"] else pkg.FixedThreshold()
    norms, k_range = cfg.get("norms", bench.NORMS), cfg.get("k_range", (float(np.finfo(np.float64).eps), 10.0))
    cd = pkg.CoalescenceData(kernels, npm, cfg["thr"], norms, ts)
    op = O.make_params(cfg["dist"], cfg["kc"], cfg["thr"], norms=norms, k_range=k_range,
                       threshold_style=O.MOVING_THRESHOLD if cfg["moving"] else O.FIXED_THRESHOLD)
    mom = moments_for(cfg["dist"], n, seed)
    tio = np.float64 if cfg["dtype"] == 0 else np.float32
    mom_in = mom.astype(tio).astype(np.float64)
    jit = cd.plan(cfg["dist"], k_range=k_range, dtype=cfg["dtype"], specialize=1)
    a = run(pkg, jit, mom_in, tio)
    if N > 4 or cfg["P"] > 5:
        b = a   # (no ahead-of-time kernels for this family)
    else:
        b = run(pkg, cd.plan(cfg["dist"], k_range=k_range, dtype=cfg["dtype"], specialize=-1), mom_in, tio)
    want, scale = O.rhs_coal_batch(op, mom_in, with_scale=True)
    with np.errstate(over="ignore"):
        fin = np.isfinite(want) & np.isfinite(want.astype(tio))
    quad = cfg["moving"] or any(np.isfinite(cfg["thr"]))
    tol = 1e-8 if quad else 1e-12
    if cfg["dtype"] == 0:
        bound = tol * scale
    else:
        bound = 6.0e-8 * np.abs(want) + tol * scale + 1.5e-45
    with np.errstate(invalid="ignore"):
        err = np.abs(a - want)
    # (float planes: where the allowed error exceeds FLT_MAX -- clamped closures with terms of 1e90 that cancel --
    # the rounding residual of the fp64 arithmetic overflows the float output; any value is within tolerance there)
    bad = fin & ~(err <= bound) & (bound < 3.0e38)
    err = np.where(bound < 3.0e38, err, 0.0)
    assert not bad.any(), f"{bad.sum()} entries beyond tolerance, worst {np.max(err[fin] / np.maximum(scale[fin], 1e-300)):.3e}"
    # plan-time compiled vs ahead-of-time kernels: same arithmetic; -ffp-contract=fast may fuse a multiply-add differently
    # in the two compilations, so agreement is to rounding (bit-identical in most families)
    both = np.isfinite(a) & np.isfinite(b)
    dj = float(np.max(np.abs(a - b)[both] / np.maximum(scale[both], 1e-300))) if both.any() else 0.0
    assert dj <= (1e-14 if cfg["dtype"] == 0 else 1e-6), f"plan-time compiled and ahead-of-time kernels differ by {dj:.2e} of scale"
    worst = float(np.max(err[fin] / np.maximum(scale[fin], 1e-300))) if fin.any() else 0.0
    if cfg["dtype"] == 0:
        check_callers(pkg, cfg, cd, op, mom_in, jit, k_range, scale)
    return worst, dj


def check_callers(pkg, cfg, cd, op, mom, plan, k_range, scale):
    """the callers either side of the operator on the same random plan (fp64 planes): closure inversion bit for bit,
    sedimentation flux, condensation, the rainshaft cell body"""
    m = pkg.DeviceArray.from_numpy(mom)
    # (n, theta, k): bit-equal to the oracle
    got = pkg.update_dist_from_moments(plan, m).to_numpy()
    want = O.update_dist_batch(op, mom)
    N = cfg["N"]
    for i, t in enumerate(cfg["dist"]):
        rows = [3 * i, 3 * i + 1] + ([3 * i + 2] if t in (1, 3) else [])
        g, w = got[rows], want[rows]
        ok = np.isfinite(w)
        if t == 3:  # Lognormal: log / sqrt of the device and the host library differ in the last place, and
            # sigma^2 = log(M0 M2 / M1^2) is ill-conditioned for narrow distributions: compare where sigma > 0.05
            # (mu = ln(M1^2 / sqrt(M0^3 M2)) is the logarithm of a number of order one: an ABSOLUTE accuracy of a few ulps of
            # one -- a parcel whose mu happens to be -9e-6 in the plan's units showed 7e-11 RELATIVE, seed 32 --wild #14)
            wide = ok & (w[2] > 0.05)[None, :]
            assert np.allclose(g[wide], w[wide], rtol=1e-11, atol=1e-14), f"update_dist_from_moments (Lognormal) mode {i}"
        else:
            assert np.array_equal(g[ok], w[ok]), f"update_dist_from_moments differs for mode {i}"
    # parcels for the time-stepping checks: closures away from the k clamps (at a clamp one rounding of a stage value
    # flips the closure and with it tendencies of 1e40: what such a parcel does after a step is not a parity question)
    regular = np.ones(mom.shape[1], dtype=bool)
    for i, t in enumerate(cfg["dist"]):
        regular &= want[3 * i] > 0.0
        if t == 1:
            regular &= (want[3 * i + 2] > max(0.05, 2.0 * k_range[0])) & (want[3 * i + 2] < 0.95 * k_range[1])
        if t == 3:
            regular &= want[3 * i + 2] > 0.05
    # sedimentation flux and the rainshaft cell body (FixedThreshold only, as in the reference)
    vel = ((50.0, 1.0 / 6), (3.0, 0.0))  # same sign: no cancellation between the two terms
    opv = O.make_params(cfg["dist"], cfg["kc"], cfg["thr"], norms=cfg.get("norms", bench.NORMS), k_range=k_range,
                        threshold_style=O.MOVING_THRESHOLD if cfg["moving"] else O.FIXED_THRESHOLD, vel=vel)
    pv = cd.plan(cfg["dist"], k_range=k_range, vel=vel)
    if not cfg["moving"]:
        cs, sf = pkg.rainshaft_sources(pv, m)
        wcs, wsf = O.rainshaft_cell_batch(opv, mom)
        okf = np.isfinite(wsf)
        assert np.allclose(sf.to_numpy()[okf], wsf[okf], rtol=1e-10, atol=0), "sedimentation flux"
        okc = np.isfinite(wcs)
        tol = 1e-8 if any(np.isfinite(cfg["thr"])) else 1e-12
        assert np.all(np.abs(cs.to_numpy() - wcs)[okc] <= tol * np.maximum(scale[okc], 1e-300)), "rainshaft coalescence source"
    # the rainshaft column integrator (FixedThreshold): 12 columns of 10 cells, 2 steps, against numpy stepping of the
    # oracle's cell body + upwind divergence
    if not cfg["moving"]:
        nz, ncol, dz = 10, 12, 100.0
        cells = np.flatnonzero(regular & np.all(np.isfinite(wcs), axis=0) & np.all(np.isfinite(wsf), axis=0) &
                               np.all(mom > 0.0, axis=0))
        if cells.size >= nz * ncol:
            c0 = np.ascontiguousarray(mom[:, cells[:nz * ncol]])

            def f_col(x):
                np.maximum(x, 0.0, out=x)
                a, fl = O.rainshaft_cell_batch(opv, x)
                out = np.empty_like(x)
                for c in range(ncol):
                    sl = slice(c * nz, (c + 1) * nz)
                    fx = np.concatenate([fl[:, sl], np.zeros((x.shape[0], 1))], axis=1)
                    out[:, sl] = a[:, sl] + (-(fx[:, 1:] - fx[:, :-1]) / dz)
                return out

            with np.errstate(divide="ignore", invalid="ignore"):
                fc = f_col(c0.copy())
                dtc = 1e-3 * float(np.nanmin(np.where(fc != 0.0, np.abs(c0 / fc), np.inf)))
            if np.isfinite(dtc) and dtc > 0.0:
                u = c0.copy()
                for _ in range(2):
                    k = f_col(u)
                    up = u
                    u = up + dtc * k
                    k = f_col(u)
                    u = (3.0 * up + u + dtc * k) / 4.0
                    k = f_col(u)
                    u = (up + 2.0 * u + 2.0 * dtc * k) / 3.0
                    np.maximum(u, 0.0, out=u)
                d_in, d_out = pkg.DeviceArray.from_numpy(c0), pkg.DeviceArray.zeros(*c0.shape)
                pkg._lib.check(pkg.lib().cloudy_rainshaft_ssprk33_steps(pv.handle, nz, ncol, nz * ncol, d_in.ptr, d_out.ptr, dz,
                                                                        dtc, 2, None))
                got = d_out.to_numpy()
                okc = np.all(np.isfinite(u), axis=0) & np.all(np.abs(u) < 10.0 * np.abs(c0) + 1e-300, axis=0)
                # a column is only comparable if all its cells stayed sane (they are coupled through the flux)
                okc = np.repeat(okc.reshape(ncol, nz).all(axis=1), nz)
                if okc.any():
                    ref = np.abs(u[:, okc]).max(axis=1, keepdims=True) + 1e-300
                    tolq = 1e-8 if any(np.isfinite(cfg["thr"])) else 1e-12
                    slack = 6.0 * dtc * tolq * scale[:, cells[:nz * ncol]][:, okc]
                    assert np.all(np.abs(got[:, okc] - u[:, okc]) <= 1e-9 * ref + slack), \
                        "cloudy_rainshaft_ssprk33_steps vs oracle stepping"
    # fused SSPRK33 (2 steps) against the oracle stepped by numpy, on the parcels whose tendencies are finite
    f0 = O.rhs_coal_batch(op, mom)
    reg = np.flatnonzero(regular & np.all(np.isfinite(f0), axis=0) & np.all(mom > 0.0, axis=0))[:128]
    if reg.size >= 8:
        u0 = np.ascontiguousarray(mom[:, reg])
        with np.errstate(divide="ignore", invalid="ignore"):
            dt = 1e-3 * float(np.nanmin(np.where(f0[:, reg] != 0.0, np.abs(u0 / f0[:, reg]), np.inf)))
        if np.isfinite(dt) and dt > 0.0:
            u = u0.copy()
            for _ in range(2):  # OrdinaryDiffEq SSPRK33
                up = u
                u = up + dt * O.rhs_coal_batch(op, up)
                u = (3.0 * up + u + dt * O.rhs_coal_batch(op, u)) / 4.0
                u = (up + 2.0 * u + 2.0 * dt * O.rhs_coal_batch(op, u)) / 3.0
            d_in, d_out = pkg.DeviceArray.from_numpy(u0), pkg.DeviceArray.zeros(*u0.shape)
            pkg._lib.check(pkg.lib().cloudy_ssprk33_steps(plan.handle, u0.shape[1], u0.shape[1], d_in.ptr, d_out.ptr, dt, 2,
                                                          None))
            got = d_out.to_numpy()
            # (the explicit scheme blows up on some parcels -- one huge rate dominates the step size of the rest of the
            # state -- and what a blown-up state cancels to is not a parity question)
            okp = np.all(np.isfinite(u), axis=0) & np.all(u > 0.0, axis=0) & np.all(np.abs(u) < 10.0 * np.abs(u0), axis=0)
            ref = np.abs(u[:, okp]).max(axis=1, keepdims=True) + 1e-300
            # a tendency is a difference of large terms: each of the 6 evaluations may differ by tol * scale
            tolq = 1e-8 if (cfg["moving"] or any(np.isfinite(cfg["thr"]))) else 1e-12
            slack = 6.0 * dt * tolq * scale[:, reg][:, okp]
            assert np.all(np.abs(got[:, okp] - u[:, okp]) <= 1e-9 * ref + slack), "cloudy_ssprk33_steps vs oracle stepping"
    # condensation
    dm = pkg.DeviceArray.zeros(*mom.shape)
    pkg.rhs_condensation(plan, dm, m, 1e-8, 0.03)
    wc = O.rhs_condensation_batch(op, 1e-8, 0.03, mom)
    okc = np.isfinite(wc)
    assert np.allclose(dm.to_numpy()[okc], wc[okc], rtol=1e-10, atol=0), "rhs_condensation"


def random_numerical_config(rng, wild=False, big=False):
    """NumericalCoalStyle plans: random N, closure families (Exponential / Gamma / Lognormal), kernel function family with
    random parameters, rule order, plane type.  big: 5 ... 8 modes (round 5: kernels compiled for the plan only; at most one
    Lognormal mode -- each one costs a 2-D rule on the CPU side too)."""
    N = int(rng.integers(5, 9)) if big else int(rng.integers(1, 5))
    dist = [int(rng.choice([0, 1, 1, 1, 3])) for _ in range(N)]
    if big:
        seen = False
        for i, t in enumerate(dist):
            if t == 3:
                dist[i] = 1 if seen else 3
                seen = True
    kind = int(rng.integers(0, 4))
    params = {0: (float(10.0 ** rng.uniform(-6, -2)),), 1: (float(10.0 ** rng.uniform(-1, 1.5)),),
              2: (float(10.0 ** rng.uniform(1, 3)),),
              3: (float(10.0 ** rng.uniform(-10.5, -8.0)), float(10.0 ** rng.uniform(9, 10.5)), float(10.0 ** rng.uniform(0, 1.5)))}[kind]
    nq = int(rng.choice([2, 3, 4, 6, 8, 10, 10, 10, 12, 16, 20, 32]))
    norms, k_range = bench.NORMS, (float(np.finfo(np.float64).eps), 10.0)
    if wild:
        norms = (float(10.0 ** rng.uniform(3, 9)), float(10.0 ** rng.uniform(-12, -7)))
        k_range = (float(rng.choice([np.finfo(np.float64).eps, 1e-3, 0.1])), float(rng.choice([5.0, 10.0, 25.0])))
    return dict(N=N, dist=dist, kind=kind, params=params, nq=nq, dtype=int(rng.choice([0, 0, 0, 1])), norms=norms,
                k_range=k_range)


def check_numerical_config(pkg, cfg, n, seed):
    """-> (worst excess of |hip - oracle| over the oracle's own (1 - w) rounding noise, in units of scale; |jit - aot|)"""
    kf_cls = [pkg.ConstantKernelFunction, pkg.LinearKernelFunction, pkg.HydrodynamicKernelFunction, pkg.LongKernelFunction]
    kfn = pkg.get_normalized_kernel_func(kf_cls[cfg["kind"]](*cfg["params"]), cfg["norms"])
    okf = O.get_normalized_kernel_func(O.kernel_func(cfg["kind"], *cfg["params"]), cfg["norms"])
    op = O.make_params(cfg["dist"], np.zeros((1, 1)), (INF,) * cfg["N"], norms=cfg["norms"], k_range=cfg["k_range"])
    mom = moments_for(cfg["dist"], n, seed)
    tio = np.float64 if cfg["dtype"] == 0 else np.float32
    mom_in = mom.astype(tio).astype(np.float64)
    jit = pkg.NumericalPlan(cfg["dist"], kfn, cfg["norms"], cfg["nq"], k_range=cfg["k_range"], dtype=cfg["dtype"], specialize=1, quad_mode=0)
    aot = jit if cfg["N"] > 4 else pkg.NumericalPlan(cfg["dist"], kfn, cfg["norms"], cfg["nq"], k_range=cfg["k_range"],
                                                     dtype=cfg["dtype"], specialize=-1, quad_mode=0)   # (no ahead-of-time kernel beyond 4 modes)
    a, b = run(pkg, jit, mom_in, tio), run(pkg, aot, mom_in, tio)
    want, scale, noise = O.rhs_coal_numerical_batch(op, okf, cfg["nq"], mom_in, with_noise=True)
    keep = np.ones(mom.shape[1], dtype=bool)
    if 3 in cfg["dist"]:
        # very narrow Lognormal modes (sigma -> eps clamp, or float-rounded moments): the Gauss-Hermite nodes exp(mu +
        # sqrt(2) sigma t) coincide to within sigma, and a kernel that vanishes on the diagonal (hydrodynamic:
        # |x^(2/3) - y^(2/3)|) amplifies the last-place difference between the device's cbrt and the host's pow by 1/sigma
        prm = O.update_dist_batch(op, mom_in)
        for i, t in enumerate(cfg["dist"]):
            if t == 3:
                keep &= prm[3 * i + 2] > 1e-3
    with np.errstate(over="ignore", invalid="ignore"):
        fin = np.isfinite(want) & np.isfinite(want.astype(tio)) & keep[None, :]
        err = np.abs(a - want) - 8.0 * noise
    bound = 1e-11 * scale if cfg["dtype"] == 0 else 6.0e-8 * np.abs(want) + 1e-11 * scale + 1.5e-45
    bad = fin & ~(err <= bound) & (bound < 3.0e38)
    err = np.where(bound < 3.0e38, np.maximum(err, 0.0), 0.0)
    assert not bad.any(), f"{bad.sum()} entries beyond tolerance, worst {np.max(err[fin] / np.maximum(scale[fin], 1e-300)):.3e}"
    both = np.isfinite(a) & np.isfinite(b) & keep[None, :]
    dj = float(np.max(np.abs(a - b)[both] / np.maximum(scale[both], 1e-300))) if both.any() else 0.0
    assert dj <= (1e-13 if cfg["dtype"] == 0 else 1e-6), f"plan-time compiled and ahead-of-time kernels differ by {dj:.2e} of scale"
    if cfg["dtype"] == 0:
        # cloudy_ssprk33_steps on the plan (quad_ssprk33_body), both kernels: one step against the same scheme staged on
        # the host with the device RHS (same arithmetic, different call structure)
        dt = 1e-3
        u = mom_in.copy()
        up = u
        u = up + dt * run(pkg, jit, up, tio)
        u = (3.0 * up + u + dt * run(pkg, jit, u, tio)) / 4.0
        u = (up + 2.0 * u + 2.0 * dt * run(pkg, jit, u, tio)) / 3.0
        with np.errstate(all="ignore"):
            okp = np.all(np.isfinite(u), axis=0) & np.all(np.abs(u) < 10.0 * np.abs(mom_in) + 1e-300, axis=0) & keep
        for plan in (jit, aot):
            d_in, d_out = pkg.DeviceArray.from_numpy(mom_in), pkg.DeviceArray.zeros(*mom_in.shape)
            pkg._lib.check(pkg.lib().cloudy_ssprk33_steps(plan.handle, mom_in.shape[1], mom_in.shape[1], d_in.ptr, d_out.ptr,
                                                          dt, 1, None))
            got = d_out.to_numpy()
            ref = np.abs(mom_in) + np.abs(u)
            e = np.abs(got - u)[:, okp] / np.maximum(ref[:, okp], 1e-300)
            assert e.size == 0 or e.max() <= 1e-11, f"cloudy_ssprk33_steps vs staged device RHS: {e.max():.2e}"
    return (float(np.max(err[fin] / np.maximum(scale[fin], 1e-300))) if fin.any() else 0.0), dj


def check_converged_config(pkg, cfg, n, seed):
    """CLOUDY_QUAD_CONVERGED plan of a random NumericalCoalStyle configuration against the same-rule oracle
    (co_rhs_coal_numerical_converged): plan-time compiled and ahead-of-time kernels, fp64 planes"""
    kf_cls = [pkg.ConstantKernelFunction, pkg.LinearKernelFunction, pkg.HydrodynamicKernelFunction, pkg.LongKernelFunction]
    kfn = pkg.get_normalized_kernel_func(kf_cls[cfg["kind"]](*cfg["params"]), cfg["norms"])
    okf = O.get_normalized_kernel_func(O.kernel_func(cfg["kind"], *cfg["params"]), cfg["norms"])
    op = O.make_params(cfg["dist"], np.zeros((1, 1)), (INF,) * cfg["N"], norms=cfg["norms"], k_range=cfg["k_range"])
    q = int(min(max(cfg["nq"], 4), 16))
    mom = moments_for(cfg["dist"], n, seed)
    jit = pkg.NumericalPlan(cfg["dist"], kfn, cfg["norms"], q, k_range=cfg["k_range"], specialize=1, quad_mode=1)
    aot = jit if cfg["N"] > 4 else pkg.NumericalPlan(cfg["dist"], kfn, cfg["norms"], q, k_range=cfg["k_range"], specialize=-1, quad_mode=1)
    a, b = run(pkg, jit, mom, np.float64), run(pkg, aot, mom, np.float64)
    # round 5: the second call of the plan ranks the parcels of a workgroup by the cost hints the first call left -- not a bit may change
    a2 = run(pkg, jit, mom, np.float64)
    assert np.array_equal(a, a2, equal_nan=True), "the cost hints of the first call changed the result of the second"
    # round 6: so does the ahead-of-time kernel (its 512-register build of a four-mode linear plan did not, tools/jit_aot_diff.py)
    assert aot is jit or np.array_equal(b, run(pkg, aot, mom, np.float64), equal_nan=True), "the ahead-of-time kernel does not repeat itself"
    want, scale = O.rhs_coal_numerical_converged_batch(op, okf, q, mom, with_scale=True)
    keep = np.ones(mom.shape[1], dtype=bool)
    if 3 in cfg["dist"]:
        prm = O.update_dist_batch(op, mom)
        for i, t in enumerate(cfg["dist"]):
            if t == 3:
                keep &= prm[3 * i + 2] > 1e-3
    assert np.array_equal(np.isnan(a[:, keep]), np.isnan(want[:, keep])), "NaN pattern differs"
    fin = np.isfinite(want) & keep[None, :]
    err = np.abs(a - want)
    bad = fin & ~(err <= 1e-11 * scale)
    worst = float(np.max(err[fin] / np.maximum(scale[fin], 1e-300))) if fin.any() else 0.0
    assert not bad.any(), f"{bad.sum()} entries beyond 1e-11 of scale, worst {worst:.3e}"
    both = np.isfinite(a) & np.isfinite(b) & keep[None, :]
    dj = float(np.max(np.abs(a - b)[both] / np.maximum(scale[both], 1e-300))) if both.any() else 0.0
    assert dj <= 1e-13, f"plan-time compiled and ahead-of-time kernels differ by {dj:.2e} of scale"
    return worst, dj


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--parcels", type=int, default=400)
    ap.add_argument("--wild", action="store_true", help="also randomise norms and the k clamp range")
    ap.add_argument("--numerical", action="store_true", help="NumericalCoalStyle (fixed Gauss rule) plans instead of tensor plans")
    ap.add_argument("--converged", action="store_true", help="NumericalCoalStyle plans in CLOUDY_QUAD_CONVERGED mode")
    ap.add_argument("--big", action="store_true", help="plans beyond the ahead-of-time families (tensor plans: 5...8 modes, or order 5...7; with --numerical / --converged: 5...8 modes)")
    a = ap.parse_args()
    pkg = load_package()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    fails = 0
    for c in range(a.configs):
        if a.converged:
            cfg = random_numerical_config(rng, a.wild, a.big)
            tag = (f"#{c} converged N={cfg['N']} dist={cfg['dist']} kernel={['constant', 'linear', 'hydro', 'long'][cfg['kind']]}"
                   f"{tuple(f'{v:.3g}' for v in cfg['params'])} q={min(max(cfg['nq'], 4), 16)}"
                   + (f" norms=({cfg['norms'][0]:.1e},{cfg['norms'][1]:.1e}) k_range={cfg['k_range']}" if a.wild else ""))
            try:
                worst, dj = check_converged_config(pkg, cfg, a.parcels, 5000 + c)
                print(f"ok   {tag}: max |hip-oracle|/scale {worst:.2e}, |jit-aot|/scale {dj:.1e}", flush=True)
            except (AssertionError, pkg.CloudyError) as e:
                fails += 1
                print(f"FAIL {tag}: {e}", flush=True)
            continue
        if a.numerical:
            cfg = random_numerical_config(rng, a.wild, a.big)
            tag = (f"#{c} numerical N={cfg['N']} dist={cfg['dist']} kernel={['constant', 'linear', 'hydro', 'long'][cfg['kind']]}"
                   f"{tuple(f'{v:.3g}' for v in cfg['params'])} nq={cfg['nq']} dtype={cfg['dtype']}"
                   + (f" norms=({cfg['norms'][0]:.1e},{cfg['norms'][1]:.1e}) k_range={cfg['k_range']}" if a.wild else ""))
            try:
                worst, dj = check_numerical_config(pkg, cfg, a.parcels, 3000 + c)
                print(f"ok   {tag}: max (|hip-oracle| - noise)/scale {worst:.2e}, |jit-aot|/scale {dj:.1e}", flush=True)
            except (AssertionError, pkg.CloudyError) as e:
                fails += 1
                print(f"FAIL {tag}: {e}", flush=True)
            continue
        cfg = random_config(rng, a.wild, a.big)
        tag = (f"#{c} N={cfg['N']} P={cfg['P']} dist={cfg['dist']} {'moving' if cfg['moving'] else 'fixed'} "
               f"thr={tuple(f'{t:.2g}' for t in cfg['thr'])} dtype={cfg['dtype']}"
               + (f" norms=({cfg['norms'][0]:.1e},{cfg['norms'][1]:.1e}) k_range={cfg['k_range']}" if a.wild else ""))
        try:
            worst, dj = check_config(pkg, cfg, a.parcels, 1000 + c)
            print(f"ok   {tag}: max |hip-oracle|/scale {worst:.2e}, |jit-aot|/scale {dj:.1e}")
        except (AssertionError, pkg.CloudyError) as e:
            fails += 1
            print(f"FAIL {tag}: {e}", flush=True)
    print(f"{a.configs} configurations, {fails} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
