"""GPU parity tests: the HIP path (through the C ABI of libcloudy_hip.so) against the CPU oracle, the reference's
known-answer values, and closed forms.  Tolerances (stated here, from BASELINE.json north_star):

  * all-Inf polynomial path (constant / Golovin / Long pieces):  north_star asks |hip - oracle| <= 1e-12 * scale, and
    rel 1e-12 against the closed forms;  rel 1e-10 vs the oracle on the Golovin analytic case
  * Simpson / incomplete-gamma path (finite or moving thresholds): north_star asks |hip - oracle| <= 1e-8 * scale

The asserted envelopes are tighter -- regression guards around what is measured on MI355X (<= 7.3e-15 on every
configuration, 2.0e-14 over random plans): TOL_POLY = 1e-13 x scale, TOL_QUAD = 1e-12 x scale, plus a plain RELATIVE
1e-10 on every output that is not a cancellation (|want| > 1e-3 x scale).

`scale` is the sum of |Q|, |R|, |S| terms of an output as reported by the oracle: a tendency is a small difference
of large terms (e.g. the net mass tendency of a mode), so rounding is measured against the terms."""
import math
import os

import ctypes as C

import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EPS = float(np.finfo(np.float64).eps)
INF = float("inf")
TOL_POLY = 1e-13   # all-Inf polynomial path (north_star: 1e-12)
TOL_QUAD = 1e-12   # Simpson / incomplete-gamma path, fp64 (north_star: 1e-8)
TOL_REL = 1e-10    # plain relative error of outputs that are not small differences of large terms
TYPES = {"exponential": 0, "gamma": 1}


def dev(cloudy, a):
    return cloudy.DeviceArray.from_numpy(a)


def run_rhs(cloudy, par, mom, ts=None):
    rhs = cloudy.make_box_model_rhs(cloudy.AnalyticalCoalStyle(), ts)
    m = dev(cloudy, mom)
    dm = cloudy.DeviceArray.zeros(*mom.shape)
    rhs(dm, m, par, 0.0)
    return dm.to_numpy()


def make_case(cloudy, oracle, dist_types, kc, thr, norms, moving=False, k_range=(EPS, 10.0), vel=()):
    """product-side ODE parameters + oracle params for the same configuration"""
    N = len(dist_types)
    kc = np.asarray(kc, dtype=np.float64)
    if kc.ndim == 2:
        kc = np.broadcast_to(kc, (N, N) + kc.shape).copy()
    kernels = tuple(tuple(cloudy.CoalescenceTensor(kc[j, k]) for k in range(N)) for j in range(N))
    npm = tuple({0: 2, 1: 3, 2: 2, 3: 3}[t] for t in dist_types)
    ts = cloudy.MovingThreshold() if moving else cloudy.FixedThreshold()
    cd = cloudy.CoalescenceData(kernels, npm, thr, norms, ts)
    mk = {0: lambda: cloudy.ExponentialPrimitiveParticleDistribution(1.0, 1.0),
          1: lambda: cloudy.GammaPrimitiveParticleDistribution(1.0, 1.0, 1.0),
          2: lambda: cloudy.MonodispersePrimitiveParticleDistribution(1.0, 1.0),
          3: lambda: cloudy.LognormalPrimitiveParticleDistribution(1.0, 1.0, 1.0)}
    pd = tuple(mk[t]() for t in dist_types)
    extra = {"vel": vel} if len(vel) else {}
    par = cloudy.ODEParameters(pd, cd, npm, norms, k_range=k_range, **extra)
    op = oracle.make_params(list(dist_types), kc, thr, norms=norms, threshold_style=1 if moving else 0,
                            k_range=k_range, vel=vel)
    return par, op, ts


def assert_close_scaled(got, want, scale, tol, what=""):
    assert got.shape == want.shape
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), f"{what}: NaN pattern differs"
    ok = ~nan_w
    err = np.abs(got[ok] - want[ok])
    bound = tol * scale[ok]
    bad = err > bound
    worst = (err / np.maximum(scale[ok], 1e-300)).max() if err.size else 0.0
    assert not bad.any(), f"{what}: max |diff|/scale = {worst:.3e} > {tol:g} ({bad.sum()} of {err.size})"
    if tol <= TOL_QUAD:   # fp64 comparisons: outputs that are not cancellations must also agree RELATIVELY
        big = np.abs(want[ok]) > 1e-3 * scale[ok]
        rel = err[big] / np.abs(want[ok][big])
        assert not (rel > TOL_REL).any(), f"{what}: max relative error of non-cancelling outputs {rel.max():.3e} > {TOL_REL:g}"
    return worst


def mixed_moments(dist_types, n, seed):
    """physical moments for a mixed Exp/Gamma mode list, (nmom, n)"""
    if len(dist_types) > 4:   # (bench.synth_moments knows the four size classes of the reference's examples)
        return many_mode_moments(dist_types, n, seed)[0]
    full = bench.synth_moments(len(dist_types), n, seed)
    rows = []
    for i, t in enumerate(dist_types):
        rows += [full[3 * i], full[3 * i + 1]] + ([full[3 * i + 2]] if t in (1, 3) else [])
    return np.ascontiguousarray(np.stack(rows))


# ----------------------------------------------------------------------------------------------------
def test_closed_forms_constant_and_golovin(gpu_cloudy, oracle):
    """SURVEY App. A.14, single Gamma mode, thresholds Inf, physical units with norms (1e6, 1e-9):
    constant c=[[A]]: dM0 = -A/2 M0^2, dM1 = 0, dM2 = A M1^2;  Golovin: dM0 = -b M0 M1, dM1 = 0, dM2 = 2b M1 M2."""
    cloudy = gpu_cloudy
    n = 100_000
    mom = bench.synth_moments(1, n, seed=5, degenerate_frac=0.0)
    M0, M1, M2 = mom
    A = 1e-9
    par, op, _ = make_case(cloudy, oracle, [1], [[A]], (INF,), bench.NORMS)
    d = run_rhs(cloudy, par, mom)
    assert np.allclose(d[0], -A / 2 * M0 * M0, rtol=TOL_POLY, atol=0)
    assert np.allclose(d[2], A * M1 * M1, rtol=TOL_POLY, atol=0)
    assert np.all(np.abs(d[1]) <= 1e-15 * A * M0 * M1)
    b = 5.0
    par, op, _ = make_case(cloudy, oracle, [1], [[0.0, b], [b, 0.0]], (INF,), bench.NORMS)
    d = run_rhs(cloudy, par, mom)
    assert np.allclose(d[0], -b * M0 * M1, rtol=TOL_POLY, atol=0)
    assert np.allclose(d[2], 2 * b * M1 * M2, rtol=TOL_POLY, atol=0)
    assert np.all(np.abs(d[1]) <= 1e-15 * b * M1 * M1)
    # the same Golovin case against the oracle: rel 1e-10 (north_star) -- and in fact 1e-12 of scale
    want, scale = oracle.rhs_coal_batch(op, mom, with_scale=True)
    assert np.allclose(d[[0, 2]], want[[0, 2]], rtol=1e-10, atol=0)
    assert_close_scaled(d, want, scale, TOL_POLY, "golovin")


@pytest.mark.parametrize("name,n,tol", [("cfg2", 1_000_000, TOL_POLY), ("cfg3a", 200_000, TOL_POLY),
                                        ("cfg3b", 20_000, TOL_QUAD), ("cfg4", 4_000, TOL_QUAD),
                                        ("moving4", 4_000, TOL_QUAD)])
def test_bench_workloads_vs_oracle(gpu_cloudy, oracle, name, n, tol):
    """The bench's synthetic Gamma-mixture batches (incl. ~1 % degenerate parcels) at oracle-sized n."""
    cloudy = gpu_cloudy
    wl = bench.make_workload(name, n, seed=99)
    d = run_rhs(cloudy, wl["par"], wl["mom"], cloudy.MovingThreshold() if wl["spec"].get("moving") else None)
    want, scale = oracle.rhs_coal_batch(bench.oracle_params(name), wl["mom"], with_scale=True)
    worst = assert_close_scaled(d, want, scale, tol, name)
    print(f"{name}: max |hip-oracle|/scale = {worst:.2e}")
    # mass conservation across modes, parcel by parcel
    nm = wl["mom"].shape[0] // 3
    net = sum(d[3 * i + 1] for i in range(nm))
    gross = sum(scale[3 * i + 1] for i in range(nm))
    assert np.all(np.abs(net) <= 1e-12 * gross)


def test_gamma_exp_reference_kat(gpu_cloudy, oracle, kats):
    """test_Sources_correctness.jl:87-169 through cloudy_get_coal_ints: Gamma(100, 0.1, 1) + Exp(1, 1), linear
    kernel order 1, thresholds (0.5, Inf), NProgMoms (3, 2), norms (1, 1)."""
    cloudy = gpu_cloudy
    e = kats["gamma_exp_coal_ints"]
    kc = np.array(e["kernel_c"])
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor(kc), (3, 2), (0.5, INF))
    params = cloudy.pack_params([(np.full(64, 100.0), np.full(64, 0.1), np.full(64, 1.0)),
                                 (np.full(64, 1.0), np.full(64, 1.0))])
    out = cloudy.get_coal_ints(cloudy.AnalyticalCoalStyle(), ([1, 0], dev(cloudy, params)), cd).to_numpy()
    assert np.all(out == out[:, :1])  # every lane identical
    got = out[:, 0]
    pd = [oracle.make_dist(1, 100.0, 0.1, 1.0), oracle.make_dist(0, 1.0, 1.0)]
    want, scale = oracle.get_coal_ints(pd, oracle.coalescence_data(kc, (3, 2), (0.5, INF)), with_scale=True)
    assert np.all(np.abs(got - want) <= TOL_QUAD * scale)
    assert np.all(np.abs(got - np.array(e["restated"])) <= TOL_QUAD * scale)
    assert abs(got[1] + got[4]) <= 1e-13 * scale[1]  # mass moves from mode 1 to mode 2, nothing is lost
    print("gamma+exp KAT max |diff|/scale:", np.max(np.abs(got - want) / scale))


def test_sm1916_constant_kernel_euler_steps(gpu_cloudy, kats):
    """test_Sources_correctness.jl:41-85: 5 explicit Euler steps, constant kernel, one Exponential mode, against the
    Smoluchowski (1916) solution 1/(1/a + b t/2); the RHS runs on the GPU, the Euler update on the host."""
    cloudy = gpu_cloudy
    e = kats["sm1916"]
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor(e["kernel_c"]), (2,), (INF,))
    par = cloudy.ODEParameters((cloudy.ExponentialPrimitiveParticleDistribution(1.0, 1.0),), cd, (2,), (1.0, 1.0))
    mom = np.tile(np.array(e["init_moments"])[:, None], (1, 3))
    for _ in range(e["n_steps"]):
        mom = mom + e["dt"] * run_rhs(cloudy, par, mom)
    t = e["dt"] * e["n_steps"]
    assert np.allclose(mom[0], 1.0 / (1.0 / e["a"] + e["b"] / 2 * t), rtol=1e-7)
    assert np.all(mom[1] == 2.0)
    for i in range(e["n_steps"] + 1):  # the reference's (loose) comparison loop
        assert np.allclose(mom[0], 1.0 / (1.0 / e["a"] + e["b"] / 2 * e["dt"] * i), rtol=e["rtol"])


def test_update_dist_from_moments_kats_and_fallbacks(gpu_cloudy, oracle, kats):
    cloudy = gpu_cloudy
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor([[1.0]]), (3, 2), (INF, INF))
    plan = cd.plan([1, 0])
    cols = [
        [10.0, 50.0, 300.0, 10.0, 50.0],      # :141-143 -> (10, 1, 5) ; :89-91 -> (10, 5)
        [1.1, 2.423, 8.112, 1.1, 2.0],        # :137-140 ; :84-87
        [1.1, 0.0, 8.112, 1.1, 0.0],          # degenerate -> (0, 1, 1) ; (0, 1)
        [EPS, 1.0, 1.0, 1.0, EPS],            # M0 <= eps / M1 <= eps
        [1.0, 1.0, 1.0, 2.0, 3.0],            # zero variance -> k = 10
        [1.0, 1.0, 0.5, 2.0, 3.0],            # negative variance -> k = eps
    ]
    mom = np.ascontiguousarray(np.array(cols).T)
    got = cloudy.update_dist_from_moments(plan, dev(cloudy, mom)).to_numpy()
    assert tuple(got[:3, 0]) == (10.0, 1.0, 5.0) and tuple(got[3:5, 0]) == (10.0, 5.0)
    assert tuple(got[:3, 2]) == (0.0, 1.0, 1.0) and tuple(got[3:5, 2]) == (0.0, 1.0)
    assert tuple(got[:3, 3]) == (0.0, 1.0, 1.0) and tuple(got[3:5, 3]) == (0.0, 1.0)
    assert got[2, 4] == 10.0 and got[1, 4] == 0.1
    assert got[2, 5] == EPS
    for j in range(mom.shape[1]):
        g = oracle.update_dist_from_moments(1, mom[:3, j])
        x = oracle.update_dist_from_moments(0, mom[3:, j])
        assert np.allclose(got[:, j], [g.n, g.theta, g.k, x.n, x.theta, 1.0], rtol=4 * EPS, atol=0)
    # k_range override of the reference test (:127-136): (1.1, 2.0, 4.1) with k <= 5 -> M2 ~ 4.364
    plan5 = cd.plan([1, 0], k_range=(EPS, 5.0))
    g = cloudy.update_dist_from_moments(plan5, dev(cloudy, np.array([[1.1], [2.0], [4.1], [1.0], [1.0]]))).to_numpy()[:3, 0]
    assert g[2] == 5.0
    assert g[0] * g[1] ** 2 * g[2] * (g[2] + 1) == pytest.approx(4.364, rel=1e-3)
    # wrong arity throws (:92, :144)
    with pytest.raises(TypeError):
        cloudy.update_dist_from_moments(plan, dev(cloudy, np.ones((6, 4))))


def test_finite_2d_integrals_and_moment_source_helper_kats(gpu_cloudy, oracle, kats):
    """moment_source_helper KATs (test_ParticleDistributions_correctness.jl:207-213, 20 bins per log unit) read off
    the F matrices of get_finite_2d_integrals, and F against the oracle for random parcels."""
    cloudy = gpu_cloudy
    for spec, t in ((["exponential", 1.0, 0.5], 0), (["gamma", 1.0, 0.5, 2.0], 1)):
        cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor([[0.0, 1.0], [1.0, 0.0]]), (3 if t else 2, 3), (0.5, INF))
        plan = cloudy.Plan([t, 1], cd.kernel_c, (0.5, INF), (1.0, 1.0), 0, n_bins_per_log_unit=20)
        k = spec[3] if t else 1.0
        params = cloudy.pack_params([(np.full(8, spec[1]), np.full(8, spec[2]), np.full(8, k)),
                                     (np.full(8, 1.0), np.full(8, 1.0), np.full(8, 1.0))])
        F = cloudy.get_finite_2d_integrals(plan, dev(cloudy, params)).to_numpy()[:, 0].reshape(2, 4, 4)
        for e in kats["moment_source_helper"]:
            if e["dist"] == spec:
                p1, p2 = int(e["p1"]), int(e["p2"])
                lo, hi = min(p1, p2), max(p1, p2)
                # get_finite_2d_integrals only ever evaluates moment_source_helper(p1 <= p2) and mirrors it
                # (Coalescence.jl:213, 232-240); the (1, 0) KAT equals the (0, 1) entry mathematically, to the
                # Simpson error (~7e-5) numerically.
                assert F[0, lo, hi] == pytest.approx(e["expected"], rel=e["rtol"]), e["cite"]
                od = oracle.make_dist(t, *spec[1:])
                assert F[0, lo, hi] == pytest.approx(oracle.moment_source_helper(od, lo, hi, 0.5, 20), rel=1e-10)
                if p1 <= p2:
                    assert F[0, lo, hi] == pytest.approx(e["restated"], rel=1e-10), e["cite"]
                assert F[0, hi, lo] == F[0, lo, hi]
    # random parcels, default 15 bins, 3 modes with two finite thresholds, P = 3, vs the oracle
    rng = np.random.default_rng(4)
    n = 512
    N, P, M = 3, 3, 5
    kc = rng.uniform(0, 1, (P, P))
    kc = kc + kc.T
    dt = [1, 0, 1]
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor(kc), (3, 2, 3), (0.5, 7.0, INF))
    plan = cd.plan(dt)
    nn = 10 ** rng.uniform(-3, 3, (3, n))
    th = 10 ** rng.uniform(-2, 1.5, (3, n))
    kk = rng.uniform(0.3, 9, (3, n))
    kk[1] = 1.0
    params = cloudy.pack_params([(nn[i], th[i], kk[i]) for i in range(3)])
    F = cloudy.get_finite_2d_integrals(plan, dev(cloudy, params)).to_numpy().reshape(N, M, M, n)
    ocd = oracle.coalescence_data(kc, (3, 2, 3), (0.5, 7.0, INF))
    worst = 0.0
    for i in range(0, n, 7):
        pd = [oracle.make_dist(dt[m], nn[m, i], th[m, i], kk[m, i]) for m in range(3)]
        mo = oracle.get_moments_matrix(pd, M, ocd.N_mom_max)
        Fo = oracle.get_finite_2d_integrals(pd, [ocd.dist_thresholds[j] for j in range(3)], mo,
                                            [ocd.N_2d_ints[j] for j in range(3)])
        mm = mo[:, :, None] * mo[:, None, :]
        err = np.abs(F[..., i] - Fo)
        assert np.all(err <= TOL_QUAD * np.maximum(mm, 1e-300)), i
        worst = max(worst, float((err / np.maximum(mm, 1e-300)).max()))
    print("F vs oracle, max |diff| / (M_p M_q):", worst)


@pytest.mark.parametrize("dist_types,P,thr", [
    ([1], 1, (INF,)), ([0], 2, (INF,)), ([1, 1], 2, (0.3e-9, INF)), ([0, 1], 3, (2e-10, INF)),
    ([1, 0, 1], 2, (1e-10, 3e-8, INF)), ([1, 1, 1], 3, (1e-9, 1e-7, INF)), ([1, 1, 1], 5, (INF, 1e-7, INF)),
    ([1, 0, 1, 1], 2, (1e-10, INF, 1e-6, INF)), ([1, 1, 1, 1], 3, (1e-9, 1e-7, 1e-5, INF)),
    ([0, 0], 3, (1e-9, INF)), ([1, 1, 1, 1], 5, (INF, INF, INF, INF)),
])
def test_mode_and_order_families_vs_oracle(gpu_cloudy, oracle, dist_types, P, thr):
    """Every (N, P) family with random symmetric tensors per pair, mixed Exponential / Gamma closures, fixed
    thresholds (some Inf), against the oracle.  box_gamma_mixture_3modes.jl:29 uses (1e-9, 1e-7, Inf)."""
    cloudy = gpu_cloudy
    N = len(dist_types)
    rng = np.random.default_rng(100 * N + P)
    kc = np.zeros((N, N, P, P))
    for j in range(N):
        for k in range(j, N):
            a = rng.uniform(0, 1, (P, P)) * (rng.uniform(0, 1, (P, P)) < 0.7)
            a = (a + a.T) * np.array([[10.0 ** (3 * (x + y)) for y in range(P)] for x in range(P)])
            kc[j, k] = kc[k, j] = a
    par, op, _ = make_case(cloudy, oracle, dist_types, kc, thr, bench.NORMS)
    n = 384 if any(np.isfinite(thr)) else 20_000
    mom = mixed_moments(dist_types, n, seed=7 + N)
    d = run_rhs(cloudy, par, mom)
    want, scale = oracle.rhs_coal_batch(op, mom, with_scale=True)
    tol = TOL_QUAD if any(np.isfinite(thr)) else TOL_POLY
    worst = assert_close_scaled(d, want, scale, tol, f"N={N} P={P}")
    print(f"N={N} P={P} thr={thr}: max |hip-oracle|/scale = {worst:.2e}")


def many_mode_moments(dist_types, n, seed):
    """physical moments of N size classes between 1e-12 and 1e-4 kg (one decade each for N = 8), number densities falling
    with size as in bench.synth_moments, 1 % degenerate parcels per mode; -> (moments, mask of the parcels without a degenerate mode)"""
    N = len(dist_types)
    regular = np.ones(n, dtype=bool)
    rng = np.random.Generator(np.random.Philox(key=seed))
    edges = np.logspace(-12, -4, N + 1)
    rows = []
    for i, t in enumerate(dist_types):
        hi = 1e9 * 10.0 ** (-1.5 * i * 8 / N)
        m = bench._gamma_mode(rng, n, hi * 1e-3, hi, 0.5 if i == 0 else 1.0, 7.0, edges[i], edges[i + 1])
        z = rng.choice(n, max(n // 100, 1), replace=False)
        m[:, z[: len(z) // 2]] = 0.0
        m[2, z[len(z) // 2:]] = m[1, z[len(z) // 2:]] ** 2 / m[0, z[len(z) // 2:]]
        regular[z] = False
        rows += [m[0], m[1]] + ([m[2]] if t in (1, 3) else [])
    return np.ascontiguousarray(np.stack(rows)), regular


@pytest.mark.parametrize("dist_types,P,thr,moving", [
    ([1, 0, 1, 1, 1, 1], 2, (INF,) * 6, False),                                   # six modes, no thresholds
    ([1, 1, 1, 0, 1], 3, (1e-10, 1e-9, 1e-8, 1e-7, INF), False),                  # five modes, all thresholded
    ([1] * 8, 2, (1e-11, 1e-10, INF, 1e-8, 1e-7, 1e-6, 1e-5, INF), False),        # eight modes
    ([1, 1], 7, (INF, INF), False), ([1, 1], 8, (5e-10, INF), False),             # order 6 / 7 tensors
    ([1, 1, 1], 6, (1e-9, 1e-7, INF), False),
    ([1] * 8, 8, (INF,) * 8, False),                                              # the largest plan the ABI takes
    ([1, 0, 1, 1, 1, 1], 6, (1e-10, 1e-9, 1e-8, INF, 1e-6, INF), False),
    ([1] * 6, 2, (0.9, 0.95, 0.99, 0.9, 0.99, 1.0), True),                        # six modes, MovingThreshold
])
def test_plans_beyond_the_ahead_of_time_families(gpu_cloudy, oracle, dist_types, P, thr, moving):
    """The reference bounds neither the number of modes nor the tensor order (CoalescenceData{N, P}, Coalescence.jl:55-104).
    Plans of more than 4 modes or order > 4 run the kernels compiled for the plan at plan creation (include/cloudy_hip.h,
    CLOUDY_AOT_MAX_MODES): RHS against the oracle to the same tolerances as the smaller families, the fused integrators
    against stepping with that RHS, every diagnostic entry point against the oracle, and a loud CLOUDY_EUNSUPPORTED when
    plan-time compilation is switched off."""
    cloudy = gpu_cloudy
    N = len(dist_types)
    rng = np.random.default_rng(1000 * N + P)
    kc = np.zeros((N, N, P, P))
    for j in range(N):
        for k in range(j, N):
            a = rng.uniform(0, 1, (P, P)) * (rng.uniform(0, 1, (P, P)) < 0.6)
            # (10^(x + y), not 10^(3 (x + y)) -- with orders up to 7 and size classes up to 1e5 normalised units
            # the larger factor pushed tendencies to 1e197; the oracle RHS of this batch is finite for EVERY parcel, so the
            # device RHS must be too: asserted below, element by element)
            a = (a + a.T) * np.array([[10.0 ** (x + y) for y in range(P)] for x in range(P)])
            kc[j, k] = kc[k, j] = a
    par, op, ts = make_case(cloudy, oracle, dist_types, kc, thr, bench.NORMS, moving=moving)
    thresholded = moving or any(np.isfinite(thr))
    n = 512 if thresholded else 20_000
    mom, regular = many_mode_moments(dist_types, n, seed=11 + N + P)
    d = run_rhs(cloudy, par, mom, ts)
    plan = par.coal_data.plan(dist_types)
    assert plan.specialized                                                # the plan-time compiled kernel ran
    want, scale = oracle.rhs_coal_batch(op, mom, with_scale=True)
    worst = assert_close_scaled(d, want, scale, TOL_QUAD if thresholded else TOL_POLY, f"N={N} P={P}")
    rows = np.cumsum([0] + [2 if t in (0, 2) else 3 for t in dist_types])[:-1] + 1
    assert np.isfinite(want).all() and np.array_equal(np.isfinite(d), np.isfinite(want))   # finite exactly where the oracle is
    ok = np.isfinite(d).all(axis=0)
    mass = np.abs(d[rows][:, ok].sum(axis=0)) / scale[rows][:, ok].sum(axis=0)
    print(f"N={N} P={P} thr={thr}: max |hip-oracle|/scale = {worst:.2e}, mass-rate residual {mass.max():.1e}")
    assert mass.max() < 1e-12
    # fused integrators of the same plan against host stepping with the device RHS (SSPRK33: three evaluations per step)
    u0 = dev(cloudy, mom)
    f = lambda u: run_rhs(cloudy, par, np.ascontiguousarray(u), ts)

    def staged(u, h):   # OrdinaryDiffEq's SSPRK33 driven from the host, the device RHS per stage
        u1 = u + h * f(u)
        u2 = 0.75 * u + 0.25 * (u1 + h * f(u1))
        return u / 3.0 + (2.0 / 3.0) * (u2 + h * f(u2))

    # (a) VERDICT r5 weak #9: the WHOLE batch -- its degenerate parcels (closures on a clamp, tendencies up to 1e30 x the regular
    # ones) included -- over a step sized so that NO parcel changes a moment by more than ~1e-3 of itself: everything stays
    # finite, and the fused step must be finite element by element exactly where the staged one is (round 5 accepted up to
    # 0.5 % of parcels that differed: its step was sized for the regular parcels and the degenerate ones overflowed, in
    # physical units on the host and in normalised units in the kernel)
    with np.errstate(all="ignore"):
        dt_all = 1e-3 * float(np.min(np.where((np.abs(d) > 0) & (mom > 0) & ok, mom / np.abs(d), np.inf)))
    assert 0 < dt_all < np.inf
    out_a = cloudy.DeviceArray.zeros(*mom.shape)
    cloudy.solve_ssprk33(par, u0, dt_all, 1, out=out_a)
    got_a, want_a = out_a.to_numpy(), staged(mom, dt_all)
    assert np.array_equal(np.isfinite(got_a), np.isfinite(want_a)) and np.isfinite(got_a).all()
    bound_a = np.maximum(np.maximum(np.abs(mom), dt_all * scale), 1e-300)
    assert (np.abs(got_a - want_a) / bound_a)[:, regular].max() <= 1e-12   # (a parcel ON a clamp amplifies the last bit of a stage)
    # (b) a real step -- sized for the regular parcels, on the regular parcels
    with np.errstate(all="ignore"):   # a step that changes no moment of any regular parcel by more than ~1e-3 of itself
        dt = 1e-3 * float(np.min(np.where((np.abs(d) > 0) & (mom > 0) & (ok & regular), mom / np.abs(d), np.inf)))
    assert 0 < dt < np.inf
    out = cloudy.DeviceArray.zeros(*mom.shape)
    cloudy.solve_ssprk33(par, u0, dt, 1, out=out)
    u3 = staged(mom, dt)
    got = out.to_numpy()
    fin = np.isfinite(u3).all(axis=0) & np.isfinite(got).all(axis=0) & regular
    assert np.array_equal(np.isfinite(u3)[:, regular], np.isfinite(got)[:, regular]) and fin.sum() == regular.sum()
    assert fin.mean() > 0.85 and (np.abs(got - mom) / np.maximum(np.abs(mom), 1e-300))[:, fin].max() > 1e-4   # a real step
    bound = np.maximum(np.maximum(np.abs(mom), dt * scale), 1e-300)
    r3 = (np.abs(got - u3) / bound)[:, fin].max()
    out5 = cloudy.DeviceArray.zeros(*mom.shape)
    cloudy.solve_tsit5(par, u0, dt, 1, out=out5)
    g5 = out5.to_numpy()
    # one Tsit5 step and one SSPRK33 step of the same dt differ by O(dt^4) of the state -- where the RHS is smooth over the
    # step: a parcel sitting exactly on a clamp of the closure inversion (zero variance, an empty mode) is not, and dt
    # is sized for the regular parcels
    fin &= np.isfinite(g5).all(axis=0)
    r5 = (np.abs(g5 - got) / bound)[:, fin].max()
    print(f"dt = {dt:.2e}: fused SSPRK33 vs three RHS calls {r3:.1e}, Tsit5 vs SSPRK33 {r5:.1e} (of max(|u|, dt scale))")
    assert r3 <= 1e-12 and r5 <= 1e-6
    # ---- the diagnostics and the parameter-plane entry points of such a plan: the bodies of the ahead-of-time kernels compiled
    # for the plan on first use (jit.hpp part 7)
    prm_d = cloudy.update_dist_from_moments(plan, u0)
    prm, prm_o = prm_d.to_numpy(), oracle.update_dist_batch(op, mom)
    assert np.allclose(prm[:, regular], prm_o[:, regular], rtol=1e-14, atol=0)
    nmm, n2d, thr_n, _, mom_norms = plan.get()
    ci = cloudy.get_coal_ints(cloudy.AnalyticalCoalStyle(), (dist_types, prm_d), par.coal_data, ts).to_numpy()
    assert_close_scaled(ci * mom_norms[:, None], want, scale, TOL_QUAD if thresholded else TOL_POLY, "get_coal_ints on (n, theta, k) planes")
    thr_dev = cloudy.compute_thresholds(plan, prm_d).to_numpy()
    M = P + 2
    if not moving:
        assert np.all(thr_dev == np.asarray(thr_n)[:, None])
        F = cloudy.get_finite_2d_integrals(plan, prm_d).to_numpy().reshape(N, M, M, n)
        worstF = 0.0
        for i in np.flatnonzero(regular)[:: max(n // 12, 1)][:12]:
            pd = [oracle.make_dist(dist_types[m], prm_o[3 * m, i], prm_o[3 * m + 1, i], prm_o[3 * m + 2, i]) for m in range(N)]
            mo = oracle.get_moments_matrix(pd, M, nmm)
            Fo = oracle.get_finite_2d_integrals(pd, [float(t) for t in thr_n], mo, [int(q) for q in n2d])
            mm = np.maximum(mo[:, :, None] * mo[:, None, :], 1e-300)
            worstF = max(worstF, float((np.abs(F[..., i] - Fo) / mm).max()))
        print(f"get_finite_2d_integrals vs oracle: {worstF:.1e} of M_p M_q")
        assert worstF <= TOL_QUAD
    else:
        for i in np.flatnonzero(regular)[:5]:
            for m in range(N - 1):
                od = oracle.make_dist(dist_types[m], prm_o[3 * m, i], prm_o[3 * m + 1, i], prm_o[3 * m + 2, i])
                assert thr_dev[m, i] == pytest.approx(oracle.compute_threshold(od, thr[m]), rel=1e-11)
    xi, sv = 1e-8, 0.03
    dmc = cloudy.DeviceArray.zeros(*mom.shape)
    cloudy.rhs_condensation(plan, dmc, u0, xi, sv)
    assert np.allclose(dmc.to_numpy()[:, regular], oracle.rhs_condensation_batch(op, xi, sv, mom)[:, regular], rtol=1e-11, atol=0)
    q = cloudy.get_standard_N_q(plan, u0, 1e-9).to_numpy()
    for i in np.flatnonzero(regular)[:8]:
        pd = [oracle.make_dist(dist_types[m], prm_o[3 * m, i], prm_o[3 * m + 1, i], prm_o[3 * m + 2, i]) for m in range(N)]
        w = oracle.get_standard_N_q(pd, 1e-9 / bench.NORMS[1]) * np.array([1e6, 1e6, 1e6 * 1e-9, 1e6 * 1e-9])
        assert np.allclose(q[:, i], w, rtol=1e-10, atol=1e-14 * np.abs(w).max())
    if not moving:   # make_rainshaft_rhs uses FixedThreshold (rainshaft_helpers.jl:70)
        vel = ((50.0, 1.0 / 6),)
        par_v, op_v, _ = make_case(cloudy, oracle, dist_types, kc, thr, bench.NORMS, vel=vel)
        plan_v = par_v.coal_data.plan(dist_types, vel=vel)
        fl = cloudy.get_sedimentation_flux(plan_v, u0).to_numpy()
        cs, sf = cloudy.rainshaft_sources(plan_v, u0)
        wcs, wsf = oracle.rainshaft_cell_batch(op_v, mom)
        assert np.allclose(fl[:, regular], wsf[:, regular], rtol=1e-11, atol=0)
        assert np.allclose(sf.to_numpy()[:, regular], wsf[:, regular], rtol=1e-11, atol=0)
        assert np.all(np.abs(cs.to_numpy() - wcs)[:, regular] <= TOL_QUAD * np.maximum(scale, 1e-300)[:, regular])
    # without plan-time compilation such a plan cannot exist: a loud error, no fallback
    with pytest.raises(cloudy.CloudyError) as e:
        par.coal_data.plan(dist_types, specialize=-1)
    assert e.value.code == cloudy._lib.EUNSUPPORTED and "plan-time compilation" in str(e.value)


def test_moving_threshold_vs_oracle_and_threshold_kats(gpu_cloudy, oracle, kats):
    """MovingThreshold (Coalescence.jl:152-185; box_gamma_mix_moving.jl:31: percentiles (0.99, 0.99, 0.99, 1.0))."""
    cloudy = gpu_cloudy
    b = 5.0
    dist_types = [1, 1, 0, 1]
    pct = (0.99, 0.97, 0.5, 1.0)
    par, op, ts = make_case(cloudy, oracle, dist_types, [[EPS, b], [b, 0.0]], pct, bench.NORMS, moving=True)
    n = 256
    mom = mixed_moments(dist_types, n, seed=21)
    d = run_rhs(cloudy, par, mom, ts)
    want, scale = oracle.rhs_coal_batch(op, mom, with_scale=True)
    worst = assert_close_scaled(d, want, scale, TOL_QUAD, "moving")
    print("moving threshold: max |hip-oracle|/scale =", worst)
    # compute_thresholds KATs (test_ParticleDistributions_correctness.jl:257-268): Exp(10,1), Gamma(5,10,2)
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor([[1.0]]), (2, 3, 3), (0.97, 0.97, 1.0), (1.0, 1.0),
                                cloudy.MovingThreshold())
    plan = cd.plan([0, 1, 1])
    params = cloudy.pack_params([(np.full(4, 10.0), np.full(4, 1.0)), (np.full(4, 5.0), np.full(4, 10.0), np.full(4, 2.0)),
                                 (np.ones(4), np.ones(4), np.ones(4))])
    thr = cloudy.compute_thresholds(plan, dev(cloudy, params)).to_numpy()[:, 0]
    assert thr[0] == pytest.approx(3.507, rel=1e-3) and thr[0] == pytest.approx(-math.log(1 - 0.97), rel=1e-14)
    import scipy.special as sp

    assert thr[1] == pytest.approx(10.0 * sp.gammaincinv(2.0, 0.97), rel=1e-12)
    assert thr[1] == pytest.approx(oracle.compute_threshold(oracle.make_dist(1, 5.0, 10.0, 2.0), 0.97), rel=1e-12)
    assert np.isinf(thr[2])
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor([[1.0]]), (2, 3), (0.5, 1.0), (1.0, 1.0), cloudy.MovingThreshold())
    thr = cloudy.compute_thresholds(cd.plan([0, 1]), dev(cloudy, params[:6])).to_numpy()[:, 0]
    assert thr[0] == pytest.approx(0.6931, rel=1e-3) and thr[0] == pytest.approx(math.log(2.0), rel=1e-14)
    cd0 = cloudy.CoalescenceData(cloudy.CoalescenceTensor([[1.0]]), (2, 3, 3), (0.0, 0.0, 1.0), (1.0, 1.0),
                                 cloudy.MovingThreshold())
    thr0 = cloudy.compute_thresholds(cd0.plan([0, 1, 1]), dev(cloudy, params)).to_numpy()[:, 0]
    assert abs(thr0[0]) < 1e-6 and abs(thr0[1]) < 1e-6  # percentile 0 -> max(0, 1e-18)


def test_sedimentation_flux_kat_and_rainshaft_sources(gpu_cloudy, oracle, kats):
    cloudy = gpu_cloudy
    e = kats["sedimentation"]  # Exp(1,1), vel ((1, 0), (-1, 1/6)) -> (-1 + gamma(7/6), -1 + gamma(13/6))
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor([[1.0]]), (2,), (INF,))
    plan = cd.plan([0], vel=e["vel"])
    flux = cloudy.get_sedimentation_flux(plan, dev(cloudy, np.ones((2, 5)))).to_numpy()
    ex = [-1.0 + math.gamma(1 + 1 / 6), -1.0 + math.gamma(2 + 1 / 6)]
    assert np.allclose(flux[:, 0], ex, rtol=1e-13, atol=0)
    # rainshaft per-cell sources (rainshaft_gamma_mixture.jl:39-47): Golovin b = 5, thr (2e-10, Inf), vel (50, 1/6)
    vel = ((50.0, 1.0 / 6),)
    par, op, _ = make_case(cloudy, oracle, [1, 1], [[EPS, 5.0], [5.0, 0.0]], (2e-10, INF), bench.NORMS, vel=vel)
    n = 300
    mom = bench.synth_moments(2, n, seed=13)
    mom[:, :40] = 0.0               # empty cells (the initial rainshaft column is mostly empty)
    mom[3:, 40:80] = 0.0            # no rain yet
    mom[2, 80:90] *= -1.0           # negative moments are clamped to zero in place (:52)
    mom[:, 90:95] = 1e-30           # all normalised moments < eps -> coalescence skipped (:67-68)
    plan = par.coal_data.plan([1, 1], vel=vel)
    cs, sf = cloudy.rainshaft_sources(plan, dev(cloudy, mom))
    wcs, wsf = oracle.rainshaft_cell_batch(op, mom)
    _, scale = oracle.rhs_coal_batch(op, np.maximum(mom, 0.0), with_scale=True)
    assert np.all(np.abs(cs.to_numpy() - wcs) <= TOL_QUAD * np.maximum(scale, 1e-300))
    assert np.allclose(sf.to_numpy(), wsf, rtol=1e-11, atol=0)
    assert np.all(cs.to_numpy()[:, :40] == 0.0) and np.all(cs.to_numpy()[:, 90:95] == 0.0)


def test_shapes_ragged_sizes_and_padding(gpu_cloudy, oracle):
    """empty batch, one parcel, sizes that are not multiples of the 64-lane wave / 256-thread block, ld > n."""
    cloudy = gpu_cloudy
    import ctypes as C

    wl = bench.make_workload("cfg3a", 1)
    plan = wl["coal_data"].plan(wl["dist_types"])
    L = cloudy.lib()
    assert L.cloudy_coal_rhs(plan.handle, 0, 0, None, None, None) == 0          # n = 0 is a no-op
    op = bench.oracle_params("cfg3a")
    for n in (1, 63, 64, 65, 255, 257, 1000, 100_003):
        mom = bench.synth_moments(2, n, seed=n)
        d = run_rhs(cloudy, wl["par"], mom)
        want, scale = oracle.rhs_coal_batch(op, mom, with_scale=True)
        assert_close_scaled(d, want, scale, TOL_POLY, f"n={n}")
    # ld > n: planes padded to a multiple of 256; the padding must not be touched
    n, ld = 1000, 1024
    mom = bench.synth_moments(2, n, seed=2)
    buf = np.full((6, ld), 7.0)
    buf[:, :n] = mom
    m, dm = dev(cloudy, buf), dev(cloudy, np.full((6, ld), -3.0))
    assert L.cloudy_coal_rhs(plan.handle, n, ld, m.ptr, dm.ptr, None) == 0
    out = dm.to_numpy()
    want, scale = oracle.rhs_coal_batch(op, mom, with_scale=True)
    assert_close_scaled(out[:, :n], want, scale, TOL_POLY, "ld>n")
    assert np.all(out[:, n:] == -3.0)
    assert L.cloudy_coal_rhs(plan.handle, n, n - 1, m.ptr, dm.ptr, None) == cloudy._lib.EINVAL
    # host-pointer convenience entry point gives the same numbers
    host_out = np.zeros_like(mom)
    assert L.cloudy_coal_rhs_host(plan.handle, n, n, mom.ctypes.data, host_out.ctypes.data) == 0
    assert np.array_equal(host_out, out[:, :n])


def test_error_behaviour_on_device(gpu_cloudy):
    cloudy = gpu_cloudy
    wl = bench.make_workload("cfg3a", 8)
    m = dev(cloudy, wl["mom"])
    with pytest.raises(ValueError):   # dm with the wrong number of planes
        cloudy.make_box_model_rhs(cloudy.AnalyticalCoalStyle())(cloudy.DeviceArray.zeros(5, 8), m, wl["par"], 0.0)
    with pytest.raises(AttributeError):   # NumericalCoalStyle reads p.kernel_func (box_model_helpers.jl:47-48): not in these parameters
        cloudy.make_box_model_rhs(cloudy.NumericalCoalStyle())(cloudy.DeviceArray.zeros(6, 8), m, wl["par"], 0.0)
    with pytest.raises(ValueError):   # MovingThreshold RHS on FixedThreshold data
        cloudy.make_box_model_rhs(cloudy.AnalyticalCoalStyle(), cloudy.MovingThreshold())(
            cloudy.DeviceArray.zeros(6, 8), m, wl["par"], 0.0)
    with pytest.raises(cloudy.CloudyError) as e:
        cloudy.Plan([1], np.array([[1.0, 2.0], [3.0, 4.0]]), (INF,), (1.0, 1.0), 0)
    assert e.value.code == cloudy._lib.ENOTSYMMETRIC
    # Non-finite moments: a NaN M0 or M1 fails the `> eps` guard and selects the fallback distribution exactly as in
    # the reference (ParticleDistributions.jl:461-475) -> finite output equal to the oracle's.  A NaN M2 makes k,
    # hence every higher moment of that mode, NaN: that mode's tendencies are NaN.  (Which OTHER entries of the
    # parcel are NaN is not part of parity: the kernel cancels the products common to Q and R analytically, the
    # reference subtracts NaN - NaN.)  Other parcels are untouched.
    from oracle import cloudy_oracle as O

    mom = wl["mom"].copy()
    mom[1, 3] = np.nan
    mom[2, 5] = np.nan
    d = run_rhs(cloudy, wl["par"], mom)
    want = O.rhs_coal_batch(bench.oracle_params("cfg3a"), mom)
    _, scale = O.rhs_coal_batch(bench.oracle_params("cfg3a"), np.nan_to_num(mom, nan=0.0), with_scale=True)
    assert not np.isnan(d[:, 3]).any() and np.all(np.abs(d[:, 3] - want[:, 3]) <= TOL_POLY * scale[:, 3])
    assert np.isnan(d[:3, 5]).all() and np.isnan(want[:3, 5]).all()
    keep = [i for i in range(8) if i != 5]
    assert not np.isnan(d[:, keep]).any()
    assert np.all(np.abs(d[:, keep] - want[:, keep]) <= TOL_POLY * scale[:, keep])


def test_full_size_properties_1e7(gpu_cloudy):
    """BASELINE size (1e7 parcels, cfg3a and cfg3b) through size-independent properties: mass conservation,
    linearity in the kernel tensor, the n -> lambda n scaling (tendencies are bilinear in number), and agreement
    of the two entry points (moments in / parameters in)."""
    cloudy = gpu_cloudy
    n = 10_000_000
    for name in ("cfg3a", "cfg3b"):
        wl = bench.make_workload(name, n, seed=1)
        plan = wl["coal_data"].plan(wl["dist_types"])
        m = dev(cloudy, wl["mom"])
        dm = cloudy.DeviceArray.zeros(6, n)
        rhs = cloudy.make_box_model_rhs(cloudy.AnalyticalCoalStyle())
        rhs(dm, m, wl["par"], 0.0)
        s = cloudy.moment_sums(plan, dm)
        d = dm.to_numpy()
        assert np.allclose(s, d.sum(axis=1), rtol=1e-9)
        # (1) mass: per parcel the two mass tendencies cancel
        gross = np.abs(d[1]) + np.abs(d[4])
        assert np.all(np.abs(d[1] + d[4]) <= 1e-9 * np.maximum(gross, 1e-300))
        assert abs(s[1] + s[4]) <= 1e-9 * abs(s[1])
        # (2) number can only decrease, the second moment can only grow (coalescence)
        assert np.all(d[0] + d[3] <= 0) and np.all(d[2] + d[5] >= -1e-12 * (np.abs(d[2]) + np.abs(d[5])))
        # (3) n -> 2n doubles every moment: tendencies x4
        m2 = dev(cloudy, 2.0 * wl["mom"][:, :1_000_000])
        dm2 = cloudy.DeviceArray.zeros(6, 1_000_000)
        rhs(dm2, m2, wl["par"], 0.0)
        d2 = dm2.to_numpy()
        ref = 4.0 * d[:, :1_000_000]
        # (exact for thr = Inf up to the eps guards; for finite thr the guard M_p M_q < eps is scale dependent)
        close = np.isclose(d2, ref, rtol=1e-9, atol=0) | (np.abs(d2 - ref) <= 1e-9 * (np.abs(d2) + np.abs(ref)))
        assert close.mean() > 0.999
        # (4) linearity in the tensor: rhs(c) = rhs(c_a) + rhs(c_b) with c = c_a + c_b
        if name == "cfg3a":
            kc = wl["kernel_c"]
            ka, kb = kc.copy(), kc.copy()
            ka[:, :, 0, 1] = ka[:, :, 1, 0] = 0.0
            kb[:, :, 0, 2] = kb[:, :, 2, 0] = 0.0
            kb[:, :, 0, 0] = 0.0
            outs = []
            for kk in (ka, kb):
                kern = tuple(tuple(cloudy.CoalescenceTensor(kk[j, k]) for k in range(2)) for j in range(2))
                cd = cloudy.CoalescenceData(kern, (3, 3), wl["spec"]["thresholds"], bench.NORMS)
                par = cloudy.ODEParameters(wl["par"].pdists, cd, (3, 3), bench.NORMS)
                o = cloudy.DeviceArray.zeros(6, n)
                rhs(o, m, par, 0.0)
                outs.append(o.to_numpy())
            lin = outs[0] + outs[1]
            assert np.all(np.abs(lin - d) <= 1e-11 * (np.abs(outs[0]) + np.abs(outs[1]) + 1e-300))
        del m, dm, d


def test_full_size_properties_cfg4_cfg5_1p25e7(gpu_cloudy):
    """BASELINE configs[3] (reference formulation: 3 Gamma modes, order-4 hydrodynamic tensor, two thresholds) and
    configs[4] (Long pieces + sedimentation flux, float planes) at their per-GPU size of 1.25e7 parcels, through
    size-independent properties: mass conservation parcel by parcel, number never created, the largest mode's M2 never
    destroyed, agreement of a 1e5-parcel slice computed alone with the same parcels inside the full batch (the result of a
    parcel does not depend on its neighbours or on the launch geometry), fluxes downward and linear in the velocity."""
    cloudy = gpu_cloudy
    n = 12_500_000
    L = cloudy.lib()
    rhs = cloudy.make_box_model_rhs(cloudy.AnalyticalCoalStyle())
    # ---- cfg4
    wl = bench.make_workload("cfg4", n, seed=2)
    m, dm = dev(cloudy, wl["mom"]), cloudy.DeviceArray.zeros(9, n)
    rhs(dm, m, wl["par"], 0.0)
    d = dm.to_numpy()
    fin = np.all(np.isfinite(d), axis=0)
    assert fin.mean() > 0.97            # (clamped closures of the ~1 % degenerate parcels may overflow, as in the oracle)
    mass, gross = d[1] + d[4] + d[7], np.abs(d[1]) + np.abs(d[4]) + np.abs(d[7])
    # (the fitted order-4 tensor has coefficients of both signs up to 1e7 times its net effect: each mass tendency is
    # itself a cancellation, so the residual is judged against the tendencies for the bulk and bounded for the tail)
    resid = np.abs(mass[fin]) / np.maximum(gross[fin], 1e-300)
    print(f"cfg4 @ {n}: mass residual / gross: median {np.median(resid):.1e}, 99.9 % {np.quantile(resid, 0.999):.1e}, max {resid.max():.1e}")
    # (max is 1 for a handful of the degenerate parcels: with M_p M_q < eps the reference's F = 0 rule drops self-collision
    # terms of the LAST mode, which no mode receives -- Coalescence.jl:213-216 -- so tendencies of order eps do not cancel)
    assert np.quantile(resid, 0.999) <= 1e-9
    # (the fitted order-4 tensor is not sign-definite, so number / M2 monotonicity is not a property of this configuration)
    sl = slice(7_000_003, 7_100_003)
    ms, dms = dev(cloudy, np.ascontiguousarray(wl["mom"][:, sl])), cloudy.DeviceArray.zeros(9, 100_000)
    rhs(dms, ms, wl["par"], 0.0)
    assert np.array_equal(dms.to_numpy(), d[:, sl], equal_nan=True)
    del m, dm, d, ms, dms
    # ---- cfg5: cfg3b tensors and thresholds + sedimentation, CLOUDY_F32 planes
    wl = bench.make_workload("cfg3b", n, seed=4)
    vel = ((50.0, 1.0 / 6),)
    mom32 = wl["mom"].astype(np.float32)
    m = dev(cloudy, mom32)
    cs, sf = cloudy.DeviceArray.zeros(6, n, np.float32), cloudy.DeviceArray.zeros(6, n, np.float32)
    plan = wl["coal_data"].plan(wl["dist_types"], vel=vel, dtype=1)
    cloudy._lib.check(L.cloudy_rainshaft_sources(plan.handle, n, n, m.ptr, cs.ptr, sf.ptr, None))
    c, f = cs.to_numpy().astype(np.float64), sf.to_numpy().astype(np.float64)
    fin = np.all(np.isfinite(c), axis=0) & np.all(np.isfinite(f), axis=0)
    assert fin.mean() > 0.97
    gross = np.abs(c[1]) + np.abs(c[4])
    assert np.all(np.abs(c[1] + c[4])[fin] <= 4e-7 * np.maximum(gross[fin], 1e-300))   # float planes: final rounding of each source
    assert np.all(c[0][fin] + c[3][fin] <= 1e-6 * (np.abs(c[0]) + np.abs(c[3]))[fin])   # number only decreases
    assert np.all(f[:, fin] <= 0.0)                                                      # sedimentation fluxes point down
    plan2 = wl["coal_data"].plan(wl["dist_types"], vel=((100.0, 1.0 / 6),), dtype=1)
    cloudy._lib.check(L.cloudy_sedimentation_flux(plan2.handle, n, n, m.ptr, sf.ptr, None))
    f2 = sf.to_numpy().astype(np.float64)
    ok = fin & np.all(np.abs(f) > 1e-30, axis=0) & np.all(np.abs(f2) < 1e37, axis=0)
    assert np.all(np.abs(f2[:, ok] - 2.0 * f[:, ok]) <= 3e-7 * np.abs(f2[:, ok]))       # linear in the velocity coefficient


def _ssprk33_host(rhs_fn, u, dt, n_steps):
    """OrdinaryDiffEq's SSPRK33 update formulas with a host-side RHS callable (numpy)."""
    u = u.copy()
    for _ in range(n_steps):
        up = u.copy()
        k = rhs_fn(u)
        u = up + dt * k
        k = rhs_fn(u)
        u = (3.0 * up + u + dt * k) / 4.0
        k = rhs_fn(u)
        u = (up + 2.0 * u + 2.0 * dt * k) / 3.0
    return u


def test_cfg1_single_box_trajectory_ssprk33(gpu_cloudy, oracle):
    """BASELINE configs[0] / box_single_gamma.jl:15-36: one 0-D box, Gamma(1e8, 1e-10, 1), Golovin b = 5 (order-1
    tensor), norms (1e6, 1e-9), tspan (0, 120), SSPRK33 with dt = 10.  The fused device integrator against (i) the
    same scheme driven by the oracle RHS and (ii) by the GPU RHS called stage by stage (the OrdinaryDiffEq pattern)."""
    cloudy = gpu_cloudy
    kern = cloudy.CoalescenceTensor(cloudy.LinearKernelFunction(5.0), 1, 1e-6)
    cd = cloudy.CoalescenceData(kern, (3,), (INF,), bench.NORMS)
    par = cloudy.ODEParameters((cloudy.GammaPrimitiveParticleDistribution(1e8, 1e-10, 1.0),), cd, (3,), bench.NORMS)
    u0 = np.tile(np.array([1e8, 1e-2, 2e-12])[:, None], (1, 64))
    dt, n_steps = 10.0, 12
    op = oracle.make_params([1], kern.c, (INF,), norms=bench.NORMS)
    want = _ssprk33_host(lambda u: oracle.rhs_coal_batch(op, u), u0, dt, n_steps)
    u = dev(cloudy, u0)
    cloudy.solve_ssprk33(par, u, dt, n_steps)
    got = u.to_numpy()
    assert np.all(got == got[:, :1])
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    staged = _ssprk33_host(lambda x: run_rhs(cloudy, par, x), u0, dt, n_steps)
    assert np.allclose(got, staged, rtol=1e-13, atol=0)
    # Golovin: M1 is conserved and M0(t) = M0 exp(-b M1 t) (dM0/dt = -b M0 M1); SSPRK33 is 3rd order in dt
    assert np.allclose(got[1], 1e-2, rtol=1e-13)
    assert got[0, 0] == pytest.approx(1e8 * math.exp(-5.0 * 1e-2 * 120.0), rel=6e-2)  # dt*b*M1 = 0.5 per step
    # out-of-place call leaves the input untouched and n_steps = 0 is the identity
    u_in, u_out = dev(cloudy, u0), cloudy.DeviceArray.zeros(3, 64)
    cloudy.solve_ssprk33(par, u_in, dt, 0, out=u_out)
    assert np.array_equal(u_out.to_numpy(), u0) and np.array_equal(u_in.to_numpy(), u0)


_TSIT5_A = ((0.161,),
            (-0.008480655492356989, 0.335480655492357),
            (2.8971530571054935, -6.359448489975075, 4.3622954328695815),
            (5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525),
            (5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383),
            (0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774))


def _tsit5_host(rhs_fn, u, dt, n_steps):
    """the Tsit5 tableau (Tsitouras 2011; OrdinaryDiffEq's Tsit5()) with a fixed step and a host-side RHS callable"""
    u = u.copy()
    for _ in range(n_steps):
        k = [rhs_fn(u)]
        for row in _TSIT5_A[:-1]:
            k.append(rhs_fn(u + dt * sum(a * ki for a, ki in zip(row, k))))
        u = u + dt * sum(a * ki for a, ki in zip(_TSIT5_A[-1], k))
    return u


def test_cfg0_single_box_tsit5_fixed_step(gpu_cloudy, oracle):
    """BASELINE configs[0] as worded ("Single 0-D box parcel ... Golovin kernel, 3 prognostic moments, CPU Tsit5"):
    cloudy_tsit5_steps = the Tsit5 tableau per parcel with a fixed dt (no reference driver uses Tsit5, and OrdinaryDiffEq's
    step control needs a global error norm: DESIGN 3.5).  (i) against the same tableau driven by the oracle RHS on the
    host, box_single_gamma.jl's configuration and a thresholded 2-mode batch; (ii) the order of the tableau itself:
    constant kernel, M0(t) = 1 / (1 / M0 + A t / 2) (Smoluchowski 1916), error ratio ~2^5 per halving of dt."""
    cloudy = gpu_cloudy
    kern = cloudy.CoalescenceTensor(cloudy.LinearKernelFunction(5.0), 1, 1e-6)
    cd = cloudy.CoalescenceData(kern, (3,), (INF,), bench.NORMS)
    par = cloudy.ODEParameters((cloudy.GammaPrimitiveParticleDistribution(1e8, 1e-10, 1.0),), cd, (3,), bench.NORMS)
    u0 = np.tile(np.array([1e8, 1e-2, 2e-12])[:, None], (1, 64))
    dt, n_steps = 10.0, 12
    op = oracle.make_params([1], kern.c, (INF,), norms=bench.NORMS)
    want = _tsit5_host(lambda u: oracle.rhs_coal_batch(op, u), u0, dt, n_steps)
    u = dev(cloudy, u0)
    cloudy.solve_tsit5(par, u, dt, n_steps)
    got = u.to_numpy()
    assert np.all(got == got[:, :1]) and np.allclose(got, want, rtol=1e-12, atol=0)
    # Golovin: M1 conserved, M0(t) = M0 exp(-b M1 t); 5th order at dt b M1 = 0.5 per step
    assert np.allclose(got[1], 1e-2, rtol=1e-13)
    assert got[0, 0] == pytest.approx(1e8 * math.exp(-5.0 * 1e-2 * 120.0), rel=1e-4)   # (SSPRK33 at the same dt: 6e-2)
    # a thresholded two-mode batch (cfg3b): every lane of the workgroup ranks its parcels in every stage
    wl = bench.make_workload("cfg3b", 700, seed=4)
    opb = bench.oracle_params("cfg3b")
    dtb, nb_ = 1e-3, 2
    wantb = _tsit5_host(lambda v: oracle.rhs_coal_batch(opb, v), wl["mom"], dtb, nb_)
    ub = dev(cloudy, wl["mom"])
    cloudy.solve_tsit5(wl["par"], ub, dtb, nb_)
    gotb = ub.to_numpy()
    # regular parcels (as in test_fused_ssprk33_batch_vs_oracle_stepping), and not on a shape clamp: at zero variance one
    # ulp of a stage state decides between k = 10 and k = eps, and the device forms the stage states with FMAs
    prm = oracle.update_dist_batch(opb, wl["mom"])
    with np.errstate(all="ignore"):
        ok = np.isfinite(wantb).all(axis=0) & (np.abs(wantb[:3]) <= 10 * np.abs(wl["mom"][:3]) + 1e-300).all(axis=0)
        ok &= (prm[2] > 1e-3) & (prm[2] < 9.999) & (prm[5] > 1e-3) & (prm[5] < 9.999)
    assert ok.sum() > 0.85 * 700, ok.sum()
    ref = np.abs(wl["mom"]) + np.abs(wantb)
    assert (np.abs(gotb - wantb)[:, ok] / np.maximum(ref[:, ok], 1e-300)).max() < 1e-10
    # order of the tableau
    A = 1e-4
    cdc = cloudy.CoalescenceData(cloudy.CoalescenceTensor([[A]]), (3,), (INF,))
    parc = cloudy.ODEParameters((cloudy.GammaPrimitiveParticleDistribution(1.0, 1.0, 1.0),), cdc, (3,), (1.0, 1.0))
    m0 = np.array([[1e3], [2e3], [8e3]]).repeat(8, axis=1)
    T = 40.0   # A M0 T / 2 = 2: the number concentration falls to a third
    exact = 1.0 / (1.0 / 1e3 + A * T / 2.0)
    errs = []
    for steps in (4, 8, 16):
        uc = dev(cloudy, m0)
        cloudy.solve_tsit5(parc, uc, T / steps, steps)
        errs.append(abs(uc.to_numpy()[0, 0] - exact) / exact)
    print("Tsit5 fixed-step errors of M0 at dt, dt/2, dt/4:", errs)
    # at least fifth order (a wrong coefficient leaves order <= 3: ratios <= 8); these steps are still pre-asymptotic
    assert errs[0] < 1e-3 and errs[0] / errs[1] > 25.0 and errs[1] / errs[2] > 25.0 and errs[2] < 1e-7
    # status codes
    L, E = cloudy.lib(), cloudy._lib
    z6 = cloudy.DeviceArray.zeros(6, 8)
    assert L.cloudy_tsit5_steps(cd.plan([1]).handle, 8, 8, z6.ptr, z6.ptr, 1.0, -1, None) == E.EINVAL
    fast = cloudy.CoalescenceData(kern, (3, 3), (0.9e-9, INF), bench.NORMS).plan([1, 1], dtype=2)   # CLOUDY_F32_FAST
    z6f = cloudy.DeviceArray.zeros(6, 8, np.float32)
    assert L.cloudy_tsit5_steps(fast.handle, 8, 8, z6f.ptr, z6f.ptr, 1.0, 1, None) == E.EUNSUPPORTED


def test_tsit5_serves_every_plan_family_ssprk33_serves(gpu_cloudy, oracle):
    """VERDICT r3 missing #3: the reference's rhs! is solver-agnostic.  cloudy_tsit5_steps on (i) a 4-mode MovingThreshold
    plan, (ii) NumericalCoalStyle plans (converged = the default, and the 10-point rule), (iii) float planes, and (iv) with
    plan-time compilation off (the ahead-of-time kernel) -- each against the same tableau driven by the oracle RHS on the
    host; the ahead-of-time and plan-time compiled kernels against each other."""
    cloudy = gpu_cloudy
    # (i) MovingThreshold (box_gamma_mix_moving.jl): the workgroup re-ranks its parcels in every one of the 6 evaluations
    wl = bench.make_workload("moving4", 600, seed=9)
    opm = bench.oracle_params("moving4")
    dt, ns = 1e-3, 2
    want = _tsit5_host(lambda v: oracle.rhs_coal_batch(opm, v), wl["mom"], dt, ns)
    um = dev(cloudy, wl["mom"])
    cloudy.solve_tsit5(wl["par"], um, dt, ns)
    got = um.to_numpy()
    prm = oracle.update_dist_batch(opm, wl["mom"])
    with np.errstate(all="ignore"):
        ok = np.isfinite(want).all(axis=0) & (np.abs(want[:3]) <= 10 * np.abs(wl["mom"][:3]) + 1e-300).all(axis=0)
        for mode in range(4):
            ok &= (prm[3 * mode + 2] > 1e-3) & (prm[3 * mode + 2] < 9.999)
    assert ok.sum() > 0.8 * 600, ok.sum()
    ref = np.abs(wl["mom"]) + np.abs(want)
    assert (np.abs(got - want)[:, ok] / np.maximum(ref[:, ok], 1e-300)).max() < 1e-9
    # (iv) the ahead-of-time kernel of the same plan
    plan_aot = wl["coal_data"].plan(wl["dist_types"], specialize=-1)
    ua = dev(cloudy, wl["mom"])
    cloudy._lib.check(cloudy.lib().cloudy_tsit5_steps(plan_aot.handle, 600, 600, ua.ptr, ua.ptr, dt, ns, None))
    assert (np.abs(ua.to_numpy() - got)[:, ok] / np.maximum(ref[:, ok], 1e-300)).max() < 1e-12
    # (ii) NumericalCoalStyle: converged mode (the drop-in's default) and the 10-point rule, 3 Gamma modes, hydrodynamic
    from test_gpu_numerical import converged_case, numerical_case
    mom = bench.synth_moments(3, 300, seed=12, degenerate_frac=0.0)
    for conv in (True, False):
        par, opn, okf = (converged_case if conv else numerical_case)(cloudy, oracle, [1, 1, 1], "hydro")
        rhs_h = ((lambda v: oracle.rhs_coal_numerical_converged_batch(opn, okf, 8, v)) if conv else
                 (lambda v: oracle.rhs_coal_numerical_batch(opn, okf, 10, v)))
        wantn = _tsit5_host(rhs_h, mom, 1e-3, 1)
        un = dev(cloudy, mom)
        cloudy.solve_tsit5(par, un, 1e-3, 1, coal_type=cloudy.NumericalCoalStyle())
        refn = np.abs(mom) + np.abs(wantn)
        with np.errstate(all="ignore"):   # (regular parcels, as in test_fused_ssprk33_of_numerical_plans_vs_oracle_stepping)
            okn = np.isfinite(wantn).all(axis=0) & (np.abs(wantn[:2]) <= 10 * np.abs(mom[:2]) + 1e-300).all(axis=0)
        assert okn.sum() > 0.85 * mom.shape[1]
        # (the fixed rule's oracle forms 1 - weighting_fn as the reference does and carries that rounding noise, 1e-9 of a
        # tendency where a mode barely overlaps the next: tests/test_gpu_numerical.py allows tol x scale + 8 x noise)
        assert (np.abs(un.to_numpy() - wantn)[:, okn] / np.maximum(refn[:, okn], 1e-300)).max() < (1e-9 if conv else 1e-8), conv
    # (iii) float planes on the headline plan (state in fp64 registers, final rounding only)
    wl3 = bench.make_workload("cfg3a", 500, seed=5)
    want3 = _tsit5_host(lambda v: oracle.rhs_coal_batch(bench.oracle_params("cfg3a"), v), wl3["mom"].astype(np.float32).astype(np.float64),
                        1e-3, 2)
    u3 = dev(cloudy, wl3["mom"].astype(np.float32))
    cloudy.solve_tsit5(wl3["par"], u3, 1e-3, 2)
    g3 = u3.to_numpy().astype(np.float64)
    ok3 = np.isfinite(want3).all(axis=0) & (np.abs(want3[:3]) <= 10 * np.abs(wl3["mom"][:3]) + 1e-300).all(axis=0)
    assert ok3.sum() > 0.9 * 500
    assert (np.abs(g3 - want3)[:, ok3] / np.maximum(np.abs(want3[:, ok3]) + np.abs(wl3["mom"][:, ok3]), 1e-300)).max() < 3e-7


def test_f32_fast_without_thresholds_is_the_packed_single_precision_kernel(gpu_cloudy, oracle):
    """CLOUDY_F32_FAST on a plan without thresholds = float planes AND single-precision arithmetic, four
    parcels per lane as two packed pairs (csrc/allinf_f32.hpp: v_pk_fma_f32).  Error against the fp64 oracle on the float
    inputs, reported; the fp64-arithmetic CLOUDY_F32 plan beside it (final rounding only).  Shapes away from the clamps:
    single precision decides k = mean / (M2/M1 - mean) to ~1e-7 / (relative variance), which the report states.  Unaligned
    or odd-stride batches take the same packed arithmetic through scalar accesses (same bits; round 5)."""
    cloudy = gpu_cloudy
    L = cloudy.lib()
    for name in ("cfg2", "cfg3a"):
        n = 40000
        wl = bench.make_workload(name, n, seed=19)
        m32 = wl["mom"].astype(np.float32)
        want, scale = oracle.rhs_coal_batch(bench.oracle_params(name), m32.astype(np.float64), with_scale=True)
        prm = oracle.update_dist_batch(bench.oracle_params(name), m32.astype(np.float64))
        m = dev(cloudy, m32)
        res = {}
        for dt_code in (cloudy.F32, cloudy.F32_FAST):
            plan = wl["coal_data"].plan(wl["dist_types"], dtype=dt_code)
            assert plan.specialized
            dm = cloudy.DeviceArray.zeros(*m32.shape, dtype=np.float32)
            cloudy._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None))
            res[dt_code] = dm.to_numpy().astype(np.float64)
        ks = prm[2::3]
        regular = np.isfinite(want).all(axis=0) & (ks > 1e-3).all(axis=0) & (ks < 9.99).all(axis=0) & (scale > 0).all(axis=0)
        assert regular.mean() > 0.95
        e32 = (np.abs(res[cloudy.F32] - want) / np.maximum(scale, 1e-300))[:, regular]
        efast = (np.abs(res[cloudy.F32_FAST] - want) / np.maximum(scale, 1e-300))[:, regular]
        print(f"{name}: |hip - fp64 oracle| / scale on float planes: CLOUDY_F32 (fp64 arithmetic) max {e32.max():.1e}; "
              f"CLOUDY_F32_FAST (packed fp32 arithmetic) max {efast.max():.1e}, median {np.median(efast):.1e}, 99.9 % {np.percentile(efast, 99.9):.1e}")
        assert e32.max() < 2e-7 and np.percentile(efast, 99.9) < 1e-4 and efast.max() < 1e-2
        # regular parcels: every output finite.  Degenerate ones (~1 % of the batch): a closure clamped to k = eps has
        # theta = mean / eps ~ 1e14 normalised units, and theta^4 leaves the single-precision range -- those parcels may
        # come out Inf / NaN where the fp64 arithmetic stays finite (stated, bounded at 1 % of the batch)
        assert np.isfinite(res[cloudy.F32_FAST][:, regular]).all()
        lost = np.isfinite(res[cloudy.F32]).all(axis=0) & ~np.isfinite(res[cloudy.F32_FAST]).all(axis=0)
        print(f"   parcels finite with fp64 arithmetic, not finite in single precision: {lost.sum()} of {n}")
        assert lost.mean() < 0.01 and not lost[regular].any()
        # mass conservation of the packed kernel, parcel by parcel (the algebra has no mass term for a single mode)
        d = res[cloudy.F32_FAST]
        net = d[1::3].sum(axis=0)
        mag = np.abs(d[1::3]).sum(axis=0)
        okm = np.isfinite(net) & (mag > 0) & regular
        assert not okm.any() or (np.abs(net[okm]) / mag[okm]).max() < 5e-6   # (a single mode has no mass term at all)
        # the arithmetic is a property of the PLAN, not of the batch's layout -- an odd leading dimension (the
        # 16-B rule fails: scalar accesses) gives the SAME BITS as the aligned batch for the same parcels (rounds 3-4 answered
        # such batches with the fp64-arithmetic kernel: a shard offset changed the precision)
        n_odd = 1001
        mo = cloudy.DeviceArray.from_numpy(np.ascontiguousarray(m32[:, :n_odd]))
        planf = wl["coal_data"].plan(wl["dist_types"], dtype=cloudy.F32_FAST)
        dmo = cloudy.DeviceArray.zeros(m32.shape[0], n_odd, dtype=np.float32)
        cloudy._lib.check(L.cloudy_coal_rhs(planf.handle, n_odd, n_odd, mo.ptr, dmo.ptr, None))
        assert np.array_equal(dmo.to_numpy().astype(np.float64), res[cloudy.F32_FAST][:, :n_odd], equal_nan=True)


def test_f64_relaxed_dtype_error_report(gpu_cloudy, oracle):
    """CLOUDY_F64_RELAXED (opt-in) stops the power series / continued fraction of the incomplete gamma
    function at 1e-11 instead of 1e-17 / 1e-16 in the plan-time compiled threshold kernels.  Its error against the fp64
    oracle must stay <= 1e-9 of scale (an order inside north_star's 1e-8 for the Simpson path); reported per workload, and
    the default dtype beside it.  Plans without a threshold compute exactly as CLOUDY_F64."""
    cloudy = gpu_cloudy
    L = cloudy.lib()
    for name in ("cfg3b", "cfg4", "moving4"):
        n = 20000
        wl = bench.make_workload(name, n, seed=23)
        want, scale = oracle.rhs_coal_batch(bench.oracle_params(name), wl["mom"], with_scale=True)
        m = dev(cloudy, wl["mom"])
        errs = {}
        for dt_code in (cloudy.F64, cloudy.F64_RELAXED):
            plan = wl["coal_data"].plan(wl["dist_types"], dtype=dt_code)
            assert plan.specialized
            dm = cloudy.DeviceArray.zeros(*wl["mom"].shape)
            cloudy._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None))
            got = dm.to_numpy()
            ok = np.isfinite(want) & (scale > 0)
            assert np.array_equal(np.isfinite(got), np.isfinite(want))
            errs[dt_code] = float((np.abs(got - want)[ok] / scale[ok]).max())
        print(f"{name}: max |hip - oracle| / scale: CLOUDY_F64 {errs[cloudy.F64]:.1e}, CLOUDY_F64_RELAXED {errs[cloudy.F64_RELAXED]:.1e}")
        assert errs[cloudy.F64] < 1e-12 and errs[cloudy.F64_RELAXED] < 1e-9
    wl = bench.make_workload("cfg3a", 4096, seed=2)
    m = dev(cloudy, wl["mom"])
    outs = []
    for dt_code in (cloudy.F64, cloudy.F64_RELAXED):
        plan = wl["coal_data"].plan(wl["dist_types"], dtype=dt_code)
        dm = cloudy.DeviceArray.zeros(*wl["mom"].shape)
        cloudy._lib.check(L.cloudy_coal_rhs(plan.handle, 4096, 4096, m.ptr, dm.ptr, None))
        outs.append(dm.to_numpy())
    assert np.array_equal(outs[0], outs[1], equal_nan=True)


def test_python_helpers_allocate_fp64_planes_for_the_relaxed_dtype(gpu_cloudy):
    """CLOUDY_F64_RELAXED (dtype code 3) has fp64 planes; the Python helpers that allocate their output
    (get_sedimentation_flux, rainshaft_sources, get_standard_N_q with out=None) used `plan.dtype >= 1 -> float32` and handed
    the kernels buffers of half the size they write.  They derive the plane type through device.plane_dtype now: the
    outputs are float64 and equal to the CLOUDY_F64 plan's wherever the relaxed series tolerance plays no part."""
    cloudy = gpu_cloudy
    n = 5000
    wl = bench.make_workload("cfg3b", n, seed=5)
    vel = ((50.0, 1.0 / 6),)
    p64 = wl["coal_data"].plan(wl["dist_types"], vel=vel, dtype=cloudy.F64)
    prx = wl["coal_data"].plan(wl["dist_types"], vel=vel, dtype=cloudy.F64_RELAXED)
    assert cloudy.plane_dtype(prx) == np.float64 and cloudy.plane_dtype(p64) == np.float64
    assert cloudy.plane_dtype(wl["coal_data"].plan(wl["dist_types"], dtype=cloudy.F32)) == np.float32
    assert cloudy.plane_dtype(wl["coal_data"].plan(wl["dist_types"], dtype=cloudy.F32_FAST)) == np.float32
    m = dev(cloudy, wl["mom"])
    f_rx, f_64 = cloudy.get_sedimentation_flux(prx, m), cloudy.get_sedimentation_flux(p64, m)
    assert f_rx.dtype == np.float64 and f_rx.shape == (6, n)
    assert np.array_equal(f_rx.to_numpy(), f_64.to_numpy(), equal_nan=True)
    cs_rx, sf_rx = cloudy.rainshaft_sources(prx, m)
    cs_64, sf_64 = cloudy.rainshaft_sources(p64, m)
    assert cs_rx.dtype == np.float64 and sf_rx.dtype == np.float64
    assert np.array_equal(sf_rx.to_numpy(), sf_64.to_numpy(), equal_nan=True)
    a, b = cs_rx.to_numpy(), cs_64.to_numpy()
    ok = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), ok) and np.max(np.abs(a - b)[ok] / np.maximum(np.abs(b[ok]), 1e-300)) < 1e-6
    nq_rx, nq_64 = cloudy.get_standard_N_q(prx, m), cloudy.get_standard_N_q(p64, m)
    assert nq_rx.dtype == np.float64 and np.array_equal(nq_rx.to_numpy(), nq_64.to_numpy(), equal_nan=True)


@pytest.mark.parametrize("name,tol", [("cfg3a", TOL_POLY), ("cfg3b", TOL_QUAD)])
def test_fused_ssprk33_batch_vs_oracle_stepping(gpu_cloudy, oracle, name, tol):
    """box_gamma_mixture_long.jl:37-46 pattern on a batch of different boxes, 4 steps.  The synthetic parcels span
    number concentrations up to 1e9 m^-3, so the explicit scheme needs dt = 1e-3 s to stay in its stability region
    (outside it both integrations blow up and are not comparable)."""
    cloudy = gpu_cloudy
    n = 400
    wl = bench.make_workload(name, n, seed=31)
    op = bench.oracle_params(name)
    dt, n_steps = 1e-3, 4
    want = _ssprk33_host(lambda u: oracle.rhs_coal_batch(op, u), wl["mom"], dt, n_steps)
    u = dev(cloudy, wl["mom"])
    cloudy.solve_ssprk33(wl["par"], u, dt, n_steps)
    got = u.to_numpy()
    # regular parcels: finite and still within a factor 10 of their initial state (degenerate parcels with
    # k = eps have tendencies ~1e30 and leave any stability region)
    with np.errstate(all="ignore"):
        ok = np.isfinite(want).all(axis=0) & (np.abs(want[:3]) <= 10 * np.abs(wl["mom"][:3]) + 1e-300).all(axis=0)
    assert ok.sum() > 0.9 * n
    # compare on the size of each moment's own trajectory (|u0| + |u|): the state is O(1) of its initial value
    ref = np.abs(wl["mom"]) + np.abs(want)
    err = np.abs(got - want)[:, ok] / np.maximum(ref[:, ok], 1e-300)
    assert err.max() < max(1e3 * tol, 1e-9), err.max()
    print(f"{name}: fused SSPRK33 vs oracle stepping, max rel err {err.max():.2e}")


@pytest.mark.parametrize("name,n", [("cfg3a", 100_001), ("cfg3b", 10_000)])
def test_fp32_planes_vs_fp64_oracle(gpu_cloudy, oracle, name, n):
    """BASELINE configs[4] "fp32 path with fp64 tolerance check": CLOUDY_F32 plans keep the moment / tendency planes
    as float in HBM (half the traffic) and compute in fp64 registers, so against the fp64 oracle evaluated on the
    same (float-representable) inputs the only error is the final rounding of each tendency to float."""
    cloudy = gpu_cloudy
    wl = bench.make_workload(name, n, seed=77)
    mom32 = wl["mom"].astype(np.float32)
    m = cloudy.DeviceArray.from_numpy(mom32)
    dm = cloudy.DeviceArray.zeros(6, n, np.float32)
    cloudy.make_box_model_rhs(cloudy.AnalyticalCoalStyle())(dm, m, wl["par"], 0.0)
    got = dm.to_numpy()
    assert got.dtype == np.float32
    want, scale = oracle.rhs_coal_batch(bench.oracle_params(name), mom32.astype(np.float64), with_scale=True)
    with np.errstate(over="ignore"):
        want32 = want.astype(np.float32)
    fin = np.isfinite(want32)
    assert np.array_equal(np.isfinite(got), fin)
    err = np.abs(got.astype(np.float64)[fin] - want[fin])
    bound = 6.0e-8 * np.abs(want[fin]) + 1e-12 * scale[fin] + 1.5e-45
    assert np.all(err <= bound), (err / np.maximum(np.abs(want[fin]), 1e-300)).max()
    big = np.abs(want[fin]) > 1e-3 * scale[fin]  # entries that are not cancellation residues
    print(f"{name} fp32 planes: max rel err vs fp64 oracle {(err[big] / np.abs(want[fin][big])).max():.2e}")
    # mixed element types are rejected
    with pytest.raises(TypeError):
        cloudy.make_box_model_rhs(cloudy.AnalyticalCoalStyle())(cloudy.DeviceArray.zeros(6, n), m, wl["par"], 0.0)


def test_cfg5_long_kernel_plus_sedimentation_fp32(gpu_cloudy, oracle):
    """BASELINE configs[4]: Long's kernel pieces + sedimentation flux (vel = ((50, 1/6),), rainshaft_gamma_mixture.jl:44),
    fp32 planes, against the fp64 oracle of the rainshaft cell body."""
    cloudy = gpu_cloudy
    n = 5000
    vel = ((50.0, 1.0 / 6),)
    spec = bench.workload_spec("cfg3b")
    kc = bench.kernel_matrix(spec)
    par, op, _ = make_case(cloudy, oracle, [1, 1], kc, spec["thresholds"], bench.NORMS, vel=vel)
    mom32 = bench.synth_moments(2, n, seed=5).astype(np.float32)
    plan = par.coal_data.plan([1, 1], vel=vel, dtype=1)
    cs, sf = cloudy.rainshaft_sources(plan, cloudy.DeviceArray.from_numpy(mom32))
    wcs, wsf = oracle.rainshaft_cell_batch(op, mom32.astype(np.float64))
    _, scale = oracle.rhs_coal_batch(op, mom32.astype(np.float64), with_scale=True)
    g1, g2 = cs.to_numpy().astype(np.float64), sf.to_numpy().astype(np.float64)
    with np.errstate(over="ignore"):
        fin = np.isfinite(wcs.astype(np.float32)) & np.isfinite(wsf.astype(np.float32))
    assert np.all(np.abs(g1 - wcs)[fin] <= 6e-8 * np.abs(wcs[fin]) + 1e-12 * scale[fin] + 1.5e-45)
    assert np.all(np.abs(g2 - wsf)[fin] <= 6e-8 * np.abs(wsf[fin]) + 1.5e-45)


@pytest.mark.parametrize("dist_types,thr", [
    ([2, 1], (5e-10, INF)),          # box_mono_gamma_mixture.jl:14-28: Monodisperse + Gamma, Golovin, thr (5e-10, Inf)
    ([3], (INF,)),                   # box_single_lognorm.jl:14-24
    ([1, 3], (5e-10, INF)),          # Gamma cloud mode + Lognormal rain mode
    ([2, 3, 1], (INF, INF, INF)),
    ([3, 3], (5e-10, INF)),          # box_lognorm_mixture.jl:15-27: two Lognormal modes, Golovin, thr (5e-10, Inf)
    ([3, 1, 3], (5e-10, 1e-7, INF)), # a Lognormal threshold and a Gamma threshold in one plan (two Simpson-type passes)
    ([3, 3, 3, 0], (1e-10, 2e-9, 5e-8, INF)),
])
def test_monodisperse_and_lognormal_closures_vs_oracle(gpu_cloudy, oracle, dist_types, thr):
    """the other two closure families (ParticleDistributions.jl:193-207, 483-505, 530-541,
    557-564) wherever the reference needs no adaptive quadrature for them."""
    cloudy = gpu_cloudy
    kern = cloudy.CoalescenceTensor(cloudy.LinearKernelFunction(5.0), 1, 1e-6)
    par, op, _ = make_case(cloudy, oracle, dist_types, kern.c, thr, bench.NORMS)
    n = 2000
    mom = mixed_moments(dist_types, n, seed=3)
    d = run_rhs(cloudy, par, mom)
    want, scale = oracle.rhs_coal_batch(op, mom, with_scale=True)
    tol = TOL_QUAD if any(np.isfinite(thr)) else TOL_POLY
    worst = assert_close_scaled(d, want, scale, tol, f"closures {dist_types}")
    print(f"closures {dist_types}: max |hip-oracle|/scale = {worst:.2e}")
    # the closure inversion itself, incl. the Lognormal KAT (10, 50, 300) -> (10, 1.518, 0.427)
    if dist_types == [3]:
        plan = par.coal_data.plan([3])
        kat = np.array([[10.0e6], [50.0e-3], [300.0e-12]])  # physical = normalised x (n0, n0 m0, n0 m0^2)
        g = cloudy.update_dist_from_moments(plan, dev(cloudy, kat)).to_numpy()[:, 0]
        assert g[0] == pytest.approx(10.0, rel=1e-3) and g[1] == pytest.approx(1.518, rel=1e-3)
        assert g[2] == pytest.approx(0.427, rel=1e-3)
        o = oracle.update_dist_from_moments(3, [10.0, 50.0, 300.0])
        assert np.allclose(g, [o.n, o.theta, o.k], rtol=1e-14)


def test_lognormal_threshold_integrals_reference_kats_and_oracle(gpu_cloudy, oracle, kats):
    """moment_source_helper(::Lognormal...) (ParticleDistributions.jl:614-625, nested quadgk in the reference) through
    cloudy_finite_2d_integrals: the reference's own KATs (LN(1, 0.5, 2), x_t = 2.5: 2.831e-1, 1.725e-1, 8.115e-2,
    test_ParticleDistributions_correctness.jl:215-218) and random parcels against the same-rule oracle, entry by entry
    relative to M_p1 M_p2; order-2 tensor (M = 5 orders share the node range)."""
    cloudy = gpu_cloudy
    ents = [e for e in kats["moment_source_helper"] if e["dist"][0] == "lognormal"]
    assert len(ents) == 3
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor(np.zeros((3, 3))), (3, 3), (2.5, INF))
    plan = cd.plan([3, 3])
    M = 5
    prm = np.array([[1.0], [0.5], [2.0], [1.0], [0.0], [1.0]]).repeat(4, axis=1)
    F = cloudy.get_finite_2d_integrals(plan, dev(cloudy, prm)).to_numpy().reshape(2, M, M, -1)
    for e in ents:
        got = F[0, int(e["p1"]), int(e["p2"]), 0]
        assert got == pytest.approx(e["expected"], rel=e["rtol"]), e["cite"]
    # random (n, mu, sigma) incl. narrow modes and thresholds far into either tail
    rng = np.random.default_rng(12)
    n = 1200
    prm = np.zeros((6, n))
    prm[0], prm[1], prm[2] = 10 ** rng.uniform(-1, 2, n), rng.uniform(-4, 3, n), 10 ** rng.uniform(-1.7, 0.3, n)
    prm[3], prm[4], prm[5] = 1.0, 0.0, 1.0
    F = cloudy.get_finite_2d_integrals(plan, dev(cloudy, prm)).to_numpy().reshape(2, M, M, n)
    worst = 0.0
    for i in range(0, n, 3):
        d = oracle.make_dist(oracle.LOGNORMAL, prm[0, i], prm[1, i], prm[2, i])
        Mq = [oracle.moment(d, float(q)) for q in range(M)]
        for p1 in range(M):
            for p2 in range(p1, M):
                mm = Mq[p1] * Mq[p2]
                if mm < EPS:
                    continue
                want = min(mm, oracle.lib().co_moment_source_helper_lognormal(C.byref(d), float(p1), float(p2), 2.5, M))
                worst = max(worst, abs(F[0, p1, p2, i] - want) / mm)
    print(f"lognormal moment_source_helper: max |hip - oracle| / (M_p1 M_p2) = {worst:.2e}")
    assert worst <= 1e-12


def test_unsupported_closure_threshold_combinations(gpu_cloudy):
    cloudy = gpu_cloudy
    kern = cloudy.CoalescenceTensor([[1.0]])
    with pytest.raises(cloudy.CloudyError) as e:   # compute_threshold has no Lognormal method (ParticleDistributions.jl:747-761)
        cloudy.CoalescenceData(kern, (3, 3), (0.9, 1.0), bench.NORMS, cloudy.MovingThreshold()).plan([3, 3])
    assert e.value.code == cloudy._lib.EINVAL
    with pytest.raises(cloudy.CloudyError) as e:   # no compute_threshold method for Monodisperse
        cloudy.CoalescenceData(kern, (2, 3), (0.9, 1.0), bench.NORMS, cloudy.MovingThreshold()).plan([2, 1])
    assert e.value.code == cloudy._lib.EINVAL


def test_condensation_rhs_vs_oracle(gpu_cloudy, oracle, kats):
    """rhs_condensation! (condensation_single_gamma.jl:17-27: s = 0.05, xi = 1e-8; condensation_exp_gamma.jl) batched,
    scalar and per-parcel supersaturation, all four closure families."""
    cloudy = gpu_cloudy
    dist_types = [0, 1, 2, 3]
    par, op, _ = make_case(cloudy, oracle, dist_types, [[1.0]], (INF,) * 4, bench.NORMS)
    plan = par.coal_data.plan(dist_types)
    n = 3000
    mom = mixed_moments(dist_types, n, seed=12)
    m = dev(cloudy, mom)
    dm = cloudy.DeviceArray.zeros(*mom.shape)
    xi, s = 1e-8, 0.05
    cloudy.rhs_condensation(plan, dm, m, xi, s)
    want = oracle.rhs_condensation_batch(op, xi, s, mom)
    assert np.allclose(dm.to_numpy(), want, rtol=1e-11, atol=0)
    rng = np.random.default_rng(0)
    s_arr = rng.uniform(-0.02, 0.05, n)
    cloudy.rhs_condensation(plan, dm, m, xi, dev(cloudy, s_arr[None, :]))
    want = oracle.rhs_condensation_batch(op, xi, s_arr, mom)
    assert np.allclose(dm.to_numpy(), want, rtol=1e-11, atol=0)
    assert np.all(dm.to_numpy()[[0, 2, 5, 7]] == 0.0)  # number is not changed by condensation
    # reference KAT (test_Sources_correctness.jl:274-283): Exp(1,1), norms (1,1)
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor([[1.0]]), (2,), (INF,))
    out = cloudy.DeviceArray.zeros(2, 4)
    cloudy.rhs_condensation(cd.plan([0]), out, dev(cloudy, np.ones((2, 4))), 1e-6, 0.01)
    C = (4 * math.pi / 3) ** (2 / 3) / 1000.0 ** (1 / 3)
    assert np.allclose(out.to_numpy()[:, 0], [0.0, 3 * 1e-6 * 0.01 * math.gamma(4 / 3) * C], rtol=1e-13, atol=0)


def test_hydrodynamic_order4_tensor_config(gpu_cloudy, oracle):
    """BASELINE configs[3] in the reference's own formulation: the hydrodynamic kernel enters as an
    order-4 fitted CoalescenceTensor (box_gamma_mixture_hydro.jl:22-29: E = 1e2 pi, limit 1e-6, thr (4e-9, Inf));
    here with three Gamma modes and thresholds (1e-9, 1e-7, Inf) (box_gamma_mixture_3modes.jl:29), 9 moments.  The
    fitted tensor is an INPUT to both sides."""
    cloudy = gpu_cloudy
    t = cloudy.CoalescenceTensor(cloudy.HydrodynamicKernelFunction(1e2 * np.pi), 4, 1e-6)
    for dist_types, thr, n in (([1, 1], (4e-9, INF), 1500), ([1, 1, 1], (1e-9, 1e-7, INF), 1000)):
        par, op, _ = make_case(cloudy, oracle, dist_types, t.c, thr, bench.NORMS)
        mom = mixed_moments(dist_types, n, seed=8)
        d = run_rhs(cloudy, par, mom)
        want, scale = oracle.rhs_coal_batch(op, mom, with_scale=True)
        worst = assert_close_scaled(d, want, scale, TOL_QUAD, f"hydro N={len(dist_types)}")
        print(f"hydro order-4, N={len(dist_types)}: max |hip-oracle|/scale = {worst:.2e}")
    # the example's own initial condition, one box
    par, op, _ = make_case(cloudy, oracle, [1, 1], t.c, (4e-9, INF), bench.NORMS)
    u0 = np.array([1e8, 1e-2, 2e-12, 1.0, 3e-9, 12e-18])[:, None].repeat(4, axis=1)
    d = run_rhs(cloudy, par, u0)
    want, scale = oracle.rhs_coal_batch(op, u0, with_scale=True)
    assert_close_scaled(d, want, scale, TOL_QUAD, "hydro example IC")


def test_rainshaft_column_rhs(gpu_cloudy, oracle):
    """make_rainshaft_rhs (rainshaft_gamma_mixture.jl:15-50: 20 cells over 3 km, two Gamma modes, Golovin b = 5,
    thr (2e-10, Inf), vel ((50, 1/6),)): coalescence source + upwind flux divergence, for 3 stacked columns."""
    cloudy = gpu_cloudy
    nz, ncol, dz = 20, 3, 150.0
    vel = ((50.0, 1.0 / 6),)
    par, op, _ = make_case(cloudy, oracle, [1, 1], [[EPS / 1e6, 5.0], [5.0, 0.0]], (2e-10, INF), bench.NORMS, vel=vel)
    par.dz, par.nz = dz, nz
    # initial_condition (rainshaft_helpers.jl:17-36): a slab between 0.5 and 0.75 of the column, plus random columns
    z = (np.arange(nz) + 0.5) * dz
    at = ((z >= 0.5 * z.max() - dz / 2) & (z < 0.75 * z.max() - dz / 2)).astype(float)
    col0 = np.outer([1e7, 1e-3, 2e-13, 0.0, 0.0, 0.0], at)
    rest = bench.synth_moments(2, nz * (ncol - 1), seed=4)
    mom = np.concatenate([col0, rest], axis=1)
    got = cloudy.make_rainshaft_rhs()(dev(cloudy, mom), par, 0.0).to_numpy()
    cs, sf = oracle.rainshaft_cell_batch(op, mom)
    want = np.empty_like(mom)
    for c in range(ncol):
        s = slice(c * nz, (c + 1) * nz)
        fl = np.concatenate([sf[:, s], np.zeros((6, 1))], axis=1)   # sedi_flux_top = 0 (:80-81)
        want[:, s] = cs[:, s] + (-(fl[:, 1:] - fl[:, :-1]) / dz)
    _, scale = oracle.rhs_coal_batch(op, np.maximum(mom, 0.0), with_scale=True)
    tol = TOL_QUAD * scale + 1e-12 * (np.abs(sf) + np.abs(np.roll(sf, -1, axis=1))) / dz + 1e-300
    assert np.all(np.abs(got - want) <= tol)
    assert np.all(got[:, :5] == 0.0)  # empty cells below the slab receive nothing from below and hold nothing


@pytest.mark.parametrize("thr_n,nbpl", [(1e-3, 15), (0.05, 15), (1.0, 15), (7.0, 15), (100.0, 15), (1e4, 15), (0.5, 10),
                                        (0.5, 20), (3.0, 33)])
def test_threshold_magnitudes_and_grid_densities(gpu_cloudy, oracle, thr_n, nbpl):
    """The Simpson grid changes with the (normalised) threshold -- n_bins = floor(nbpl * log10(x_t / x_lb)): 75 bins
    for x_t <= 1, 84 at 4, 105 at 100, 135 at 1e4 (ParticleDistributions.jl:604-607) -- and with n_bins_per_log_unit.
    The early/late node split and the regime sort must hold on all of them: F matrices against the oracle's
    moment_source_helper on every grid, for broad random (n, theta, k)."""
    cloudy = gpu_cloudy
    rng = np.random.default_rng(int(thr_n * 1000) + nbpl)
    n = 300
    kc = np.array([[0.0, 1.0, 0.5], [1.0, 0.0, 0.0], [0.5, 0.0, 0.0]])
    plan = cloudy.Plan([1, 0, 1], kc, (thr_n, thr_n * 3.0, INF), (1.0, 1.0), 0, n_bins_per_log_unit=nbpl)
    nn = 10 ** rng.uniform(-2, 3, (3, n))
    th = thr_n * 10 ** rng.uniform(-3, 2, (3, n))      # x_t / theta from 1e-2 to 1e3
    kk = rng.uniform(0.05, 10, (3, n))
    kk[1] = 1.0
    params = cloudy.pack_params([(nn[i], th[i], kk[i]) for i in range(3)])
    M = 5
    F = cloudy.get_finite_2d_integrals(plan, dev(cloudy, params)).to_numpy().reshape(3, M, M, n)
    worst = 0.0
    for i in range(0, n, 3):
        for mode, t in ((0, 1), (1, 0)):
            d = oracle.make_dist(t, nn[mode, i], th[mode, i], kk[mode, i])
            xt = thr_n * (1.0 if mode == 0 else 3.0)
            for p1 in range(M):
                for p2 in range(p1, M):
                    mm = oracle.moment(d, p1) * oracle.moment(d, p2)
                    want = 0.0 if mm < EPS else min(mm, oracle.moment_source_helper(d, p1, p2, xt, nbpl))
                    err = abs(F[mode, p1, p2, i] - want)
                    assert err <= TOL_QUAD * max(mm, 1e-300), (thr_n, nbpl, i, mode, p1, p2, F[mode, p1, p2, i], want)
                    worst = max(worst, err / max(mm, 1e-300))
    print(f"x_t={thr_n:g} nbpl={nbpl}: max |F - oracle| / (M_p M_q) = {worst:.2e}")


def test_fp32_fast_threshold_path_error_report(gpu_cloudy, oracle):
    """CLOUDY_F32_FAST (BASELINE configs[4] "fp32 path with fp64 tolerance check"): float planes and single-precision
    arithmetic in the per-node Simpson / incomplete-gamma pass.  The error against the fp64 oracle is REPORTED and
    bounded at 5e-6 of the term scale (measured 5e-7: fp32 exp arguments of O(30), 75-term sums)."""
    cloudy = gpu_cloudy
    n = 20_000
    wl = bench.make_workload("cfg3b", n, seed=123)
    par = wl["par"]
    par.fast_f32 = True
    mom32 = wl["mom"].astype(np.float32)
    m = cloudy.DeviceArray.from_numpy(mom32)
    dm = cloudy.DeviceArray.zeros(6, n, np.float32)
    cloudy.make_box_model_rhs(cloudy.AnalyticalCoalStyle())(dm, m, par, 0.0)
    got = dm.to_numpy().astype(np.float64)
    want, scale = oracle.rhs_coal_batch(bench.oracle_params("cfg3b"), mom32.astype(np.float64), with_scale=True)
    with np.errstate(over="ignore"):
        fin = np.isfinite(want.astype(np.float32)) & (scale < 1e30)
    err = np.abs(got - want)[fin] / np.maximum(scale[fin], 1e-300)
    print(f"fp32-fast cfg3b: max |hip - oracle|/scale = {err.max():.2e}, 99.9th pct {np.quantile(err, 0.999):.2e}, "
          f"median {np.median(err):.2e}")
    assert err.max() < 5e-6
    # mass is still conserved to fp32 rounding of the two mass tendencies
    net = np.abs(got[1] + got[4])
    assert np.all(net[fin[1]] <= 3e-7 * (np.abs(got[1]) + np.abs(got[4]))[fin[1]] + 1e-300)


def test_standard_N_q_diagnostics(gpu_cloudy, oracle, kats):
    """get_standard_N_q (ParticleDistributions.jl:634-687) batched: the reference KAT (Exp(10,1) + Gamma(5,10,2):
    N_liq + N_rai = 15, M_liq + M_rai = 110) and random mixtures against the oracle."""
    cloudy = gpu_cloudy
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor([[1.0]]), (2, 3), (INF, INF))
    plan = cd.plan([0, 1])
    kat = np.array([[10.0], [10.0], [5.0], [100.0], [3000.0]]).repeat(4, axis=1)   # moments of Exp(10,1), Gamma(5,10,2)
    for cutoff in kats["get_standard_N_q"]["cutoffs"]:
        q = cloudy.get_standard_N_q(plan, dev(cloudy, kat), cutoff).to_numpy()[:, 0]
        assert q[0] + q[1] == pytest.approx(15.0, rel=1e-12) and q[2] + q[3] == pytest.approx(110.0, rel=1e-12)
        want = oracle.get_standard_N_q([oracle.make_dist(0, 10.0, 1.0), oracle.make_dist(1, 5.0, 10.0, 2.0)], cutoff)
        assert np.allclose(q, want, rtol=1e-12)
    dist_types = [1, 2, 0]
    par, op, _ = make_case(cloudy, oracle, dist_types, [[1.0]], (INF,) * 3, bench.NORMS)
    plan = par.coal_data.plan(dist_types)
    n = 500
    mom = mixed_moments(dist_types, n, seed=2)
    q = cloudy.get_standard_N_q(plan, dev(cloudy, mom), 1e-9).to_numpy()
    prm = oracle.update_dist_batch(op, mom)
    for i in range(0, n, 5):
        pd = [oracle.make_dist(dist_types[m], prm[3 * m, i], prm[3 * m + 1, i], prm[3 * m + 2, i]) for m in range(3)]
        w = oracle.get_standard_N_q(pd, 1e-9 / bench.NORMS[1]) * np.array([1e6, 1e6, 1e6 * 1e-9, 1e6 * 1e-9])
        tot = np.array([w[0] + w[1], w[0] + w[1], w[2] + w[3], w[2] + w[3]])
        assert np.all(np.abs(q[:, i] - w) <= 1e-12 * tot + 1e-300), i
    # the reference's own performance-test configuration, get_standard_N_q((mono, lognormal, gamma))
    # (performance_tests.jl:94-99): Lognormal partial moments in closed form (the reference: quadgk, :255-269)
    dist_types = [2, 3, 1]
    par, op, _ = make_case(cloudy, oracle, dist_types, [[1.0]], (INF,) * 3, bench.NORMS)
    plan = par.coal_data.plan(dist_types)
    mom = mixed_moments(dist_types, n, seed=8)
    prm = oracle.update_dist_batch(op, mom)
    for cutoff in (1e-9, 3e-8):
        q = cloudy.get_standard_N_q(plan, dev(cloudy, mom), cutoff).to_numpy()
        for i in range(0, n, 5):
            pd = [oracle.make_dist(dist_types[m], prm[3 * m, i], prm[3 * m + 1, i], prm[3 * m + 2, i]) for m in range(3)]
            w = oracle.get_standard_N_q(pd, cutoff / bench.NORMS[1]) * np.array([1e6, 1e6, 1e6 * 1e-9, 1e6 * 1e-9])
            tot = np.array([w[0] + w[1], w[0] + w[1], w[2] + w[3], w[2] + w[3]])
            assert np.all(np.abs(q[:, i] - w) <= 1e-12 * tot + 1e-300), i


def test_unaligned_planes_take_the_one_parcel_per_lane_kernel(gpu_cloudy, oracle):
    """The two-parcels-per-lane kernel needs 16-byte aligned planes (even ld, aligned base); otherwise the ABI falls
    back to the one-parcel kernel.  Same numbers either way."""
    cloudy = gpu_cloudy
    L = cloudy.lib()
    n, ld = 4097, 4099                      # odd leading dimension
    wl = bench.make_workload("cfg3a", n, seed=17)
    plan = wl["coal_data"].plan(wl["dist_types"])
    buf = np.zeros((6, ld))
    buf[:, :n] = wl["mom"]
    m, dm = dev(cloudy, buf), dev(cloudy, np.zeros((6, ld)))
    assert L.cloudy_coal_rhs(plan.handle, n, ld, m.ptr, dm.ptr, None) == 0
    out_odd = dm.to_numpy()[:, :n]
    # even ld but base pointers offset by one element (8-byte aligned only)
    ld2 = 4100
    big = np.zeros(6 * ld2 + 1)
    big[1:].reshape(6, ld2)[:, :n] = wl["mom"]
    mb, db = dev(cloudy, big[None, :]), dev(cloudy, np.zeros((1, 6 * ld2 + 1)))
    assert L.cloudy_coal_rhs(plan.handle, n, ld2, mb.ptr + 8, db.ptr + 8, None) == 0
    out_off = db.to_numpy()[0, 1:].reshape(6, ld2)[:, :n]
    # aligned reference run (two parcels per lane)
    d = run_rhs(cloudy, wl["par"], wl["mom"])
    assert np.array_equal(out_odd, d) and np.array_equal(out_off, d)


def _oracle_rainshaft_ssprk33(oracle, op, u, nz, dz, dt, n_steps):
    """OrdinaryDiffEq's SSPRK33 around the rainshaft rhs restated with the oracle's cell body; each evaluation clamps
    its argument in place first (rainshaft_helpers.jl:52), the FSAL evaluation on the step result included."""
    nmom, n = u.shape

    def f(x):
        np.maximum(x, 0.0, out=x)
        cs, sf = oracle.rainshaft_cell_batch(op, x)
        out = np.empty_like(x)
        for c in range(n // nz):
            s = slice(c * nz, (c + 1) * nz)
            fl = np.concatenate([sf[:, s], np.zeros((nmom, 1))], axis=1)
            out[:, s] = cs[:, s] + (-(fl[:, 1:] - fl[:, :-1]) / dz)
        return out

    u = u.copy()
    for _ in range(n_steps):
        k = f(u)
        up = u
        u = up + dt * k
        k = f(u)
        u = (3.0 * up + u + dt * k) / 4.0
        k = f(u)
        u = (up + 2.0 * u + 2.0 * dt * k) / 3.0
        np.maximum(u, 0.0, out=u)
    return u


@pytest.mark.parametrize("case", ["gamma_mixture", "single_gamma"])
def test_rainshaft_column_integrator_ssprk33(gpu_cloudy, oracle, case):
    """cloudy_rainshaft_ssprk33_steps (one launch, state in registers, flux exchange through LDS) against the oracle
    stepped by a numpy SSPRK33: rainshaft_gamma_mixture.jl:15-60 (20 cells over 3 km, two Gamma modes, Golovin b = 5,
    thr (2e-10, Inf), vel ((50, 1/6),), dt = 1) and rainshaft_single_gamma.jl (one mode, thr Inf)."""
    cloudy = gpu_cloudy
    nz, dz, dt, n_steps = 20, 150.0, 1.0, 40
    vel = ((50.0, 1.0 / 6),)
    z = (np.arange(nz) + 0.5) * dz
    at = ((z >= 0.5 * z.max() - dz / 2) & (z < 0.75 * z.max() - dz / 2)).astype(float)
    if case == "gamma_mixture":
        par, op, _ = make_case(cloudy, oracle, [1, 1], [[EPS / 1e6, 5.0], [5.0, 0.0]], (2e-10, INF), bench.NORMS,
                               vel=vel)
        amp = np.array([1e7, 1e-3, 2e-13, 0.0, 0.0, 0.0])
    else:
        par, op, _ = make_case(cloudy, oracle, [1], [[EPS / 1e6, 5.0], [5.0, 0.0]], (INF,), bench.NORMS, vel=vel)
        amp = np.array([1e7, 1e-3, 2e-13])
    par.dz, par.nz, par.dt = dz, nz, dt
    cols = [np.outer(amp * s, at) for s in (1.0, 0.3, 3.0)]
    shifted = np.outer(amp, np.roll(at, 3))              # a slab touching the top cell
    dirty = np.outer(amp, at)
    dirty[:, 2] = -1e-30                                 # negative round-off in an empty cell is clamped
    ncol_pad = 13                                        # 12 columns per workgroup at nz = 20: spill into a second one
    cols += [shifted, dirty] + [np.outer(amp * (0.5 + 0.1 * i), at) for i in range(ncol_pad - 5)]
    u0 = np.concatenate(cols, axis=1)
    want = _oracle_rainshaft_ssprk33(oracle, op, u0, nz, dz, dt, n_steps)
    ud = dev(cloudy, u0)
    out = cloudy.DeviceArray.zeros(*u0.shape)
    cloudy.solve_rainshaft_ssprk33(par, ud, n_steps, out=out)
    got = out.to_numpy()
    assert np.array_equal(ud.to_numpy(), u0)             # out-of-place call leaves the input alone
    ref = np.abs(want).max(axis=1, keepdims=True) + 1e-300
    err = np.abs(got - want) / ref
    print(f"rainshaft {case}: {n_steps} SSPRK33 steps, max |hip-oracle| / max|plane| = {err.max():.2e}")
    assert err.max() < 1e-9
    assert got.min() >= 0.0
    # rain reached the cells below the initial slab; mass only leaves through the bottom
    assert got[1, :5].sum() > 0.0 and got[1::3, :nz].sum() <= u0[1::3, :nz].sum() * (1 + 1e-12)
    # the same steps driven from the host through make_rainshaft_rhs agree with the fused launch
    rhs = cloudy.make_rainshaft_rhs()
    u = u0[:, :2 * nz].copy()
    for _ in range(3):
        f = lambda x: rhs(dev(cloudy, np.maximum(x, 0.0, out=x)), par, 0.0).to_numpy()  # noqa: E731
        up = u
        u = up + dt * f(up)
        u = (3.0 * up + u + dt * f(u)) / 4.0
        u = (up + 2.0 * u + 2.0 * dt * f(u)) / 3.0
    np.maximum(u, 0.0, out=u)
    o3 = cloudy.DeviceArray.zeros(u.shape[0], 2 * nz)
    cloudy.solve_rainshaft_ssprk33(par, dev(cloudy, u0[:, :2 * nz].copy()), 3, out=o3)
    assert np.allclose(o3.to_numpy(), u, rtol=1e-12, atol=1e-13 * np.abs(u).max())
    # in place, zero steps, and the nz limit
    cloudy.solve_rainshaft_ssprk33(par, ud, 0)
    assert np.array_equal(ud.to_numpy(), u0)
    # VERDICT r3 missing #4: columns taller than 256 cells -- one column per workgroup of 512 / 1024 threads in the kernel
    # compiled for the plan -- and (round 5, VERDICT r4 missing #4) columns taller than 1024 cells, stepped stage by stage inside
    # the library: against the same steps driven from the host through make_rainshaft_rhs (per-stage launches)
    for nz_t in (300, 700, 1500):
        par.nz, par.dz = nz_t, 3000.0 / nz_t
        zt = (np.arange(nz_t) + 0.5) * par.dz
        att = ((zt >= 0.5 * zt.max()) & (zt < 0.75 * zt.max())).astype(float)
        ut = np.concatenate([np.outer(amp * s, att) for s in (1.0, 0.4, 2.0)], axis=1)
        par.dt = 0.05   # (the cells are 15 / 35 x thinner: the upwind flux needs the smaller step)
        u = ut.copy()
        for _ in range(2):
            f = lambda x: rhs(dev(cloudy, np.maximum(x, 0.0, out=x)), par, 0.0).to_numpy()  # noqa: E731
            up = u
            u = up + par.dt * f(up)
            u = (3.0 * up + u + par.dt * f(u)) / 4.0
            u = (up + 2.0 * u + 2.0 * par.dt * f(u)) / 3.0
        np.maximum(u, 0.0, out=u)
        ot = cloudy.DeviceArray.zeros(*ut.shape)
        cloudy.solve_rainshaft_ssprk33(par, dev(cloudy, ut.copy()), 2, out=ot)
        gt = ot.to_numpy()
        assert np.allclose(gt, u, rtol=1e-12, atol=1e-13 * np.abs(u).max()), nz_t
        assert gt.min() >= 0.0 and gt[1, : nz_t // 2].sum() > 0.0


@pytest.mark.parametrize("N, nz, ncol", [(5, 20, 40), (6, 100, 7), (3, 300, 4), (4, 300, 3), (5, 300, 3), (3, 110, 9), (8, 20, 30)])
def test_column_integrator_picks_a_workgroup_size_whose_lds_fits(gpu_cloudy, oracle, N, nz, ncol, monkeypatch):
    """round 5 picked the 512- / 1024-thread column kernel from the column height alone; a thresholded plan's LDS
    rows grow with the mode count (5 modes at 512 threads, 3 at 1024 exceed a workgroup's 160 KB), so plans of 5-8 modes with
    more than 256 / nz columns, and 3- and 4-mode plans with nz in 257..341, got EUNSUPPORTED.  Now the pick skips sizes whose LDS
    does not fit, falls back to smaller ones, and steps stage by stage when no workgroup size both holds a column and fits.
    Every case: two SSPRK33 steps against the same steps driven from the host through make_rainshaft_rhs."""
    cloudy = gpu_cloudy
    thr = tuple(10.0 ** (-10 + i) for i in range(N - 1)) + (INF,)
    par, op, _ = make_case(cloudy, oracle, [1] * N, [[EPS / 1e6, 5.0], [5.0, 0.0]], thr, bench.NORMS, vel=((50.0, 1.0 / 6),))
    par.nz, par.dz, par.dt = nz, 3000.0 / nz, 0.05 if nz > 50 else 0.5
    # a slab of every mode in the upper half of each column (as rainshaft_gamma_mixture.jl:31-37), one amplitude per column
    amp = bench.synth_moments(min(N, 4), 1, seed=77, degenerate_frac=0.0)[:, 0]
    if N > 4:   # (synth_moments has four size ranges: further modes repeat the last one, a decade up each)
        amp = np.concatenate([amp] + [amp[9:12] * np.array([0.1, 1.0, 100.0]) * 10.0 ** (m - 3) for m in range(4, N)])
    amp = amp[: 3 * N]
    z = (np.arange(nz) + 0.5) * par.dz
    at = ((z >= 0.5 * z.max()) & (z < 0.75 * z.max())).astype(float)
    u0 = np.ascontiguousarray(np.concatenate([np.outer(amp * (0.3 + 0.2 * c), at) for c in range(ncol)], axis=1))
    rhs = cloudy.make_rainshaft_rhs()
    u = u0.copy()
    for _ in range(2):
        f = lambda x: rhs(dev(cloudy, np.maximum(x, 0.0, out=x)), par, 0.0).to_numpy()  # noqa: E731
        up = u
        u = up + par.dt * f(up)
        u = (3.0 * up + u + par.dt * f(u)) / 4.0
        u = (up + 2.0 * u + 2.0 * par.dt * f(u)) / 3.0
    np.maximum(u, 0.0, out=u)
    out = cloudy.DeviceArray.zeros(*u0.shape)
    cloudy.solve_rainshaft_ssprk33(par, dev(cloudy, u0.copy()), 2, out=out)
    got = out.to_numpy()
    fin = np.isfinite(u)
    assert np.array_equal(np.isfinite(got), fin)
    assert np.allclose(got[fin], u[fin], rtol=1e-11, atol=1e-13 * np.abs(u[fin]).max()), (N, nz, ncol)
    # the experiment switch for a size that cannot serve the plan is ignored, not an error (320 threads used to
    # fall back to 256 threads whatever the column height)
    for block in ("320", "1024"):
        monkeypatch.setenv("CLOUDY_HIP_RS_BLOCK", block)
        out2 = cloudy.DeviceArray.zeros(*u0.shape)
        cloudy.solve_rainshaft_ssprk33(par, dev(cloudy, u0.copy()), 2, out=out2)
        g2 = out2.to_numpy()
        assert np.array_equal(np.isfinite(g2), fin) and np.allclose(g2[fin], u[fin], rtol=1e-11, atol=1e-13 * np.abs(u[fin]).max())
    monkeypatch.delenv("CLOUDY_HIP_RS_BLOCK", raising=False)


def test_workgroup_size_of_the_column_integrator_changes_no_bit(gpu_cloudy, monkeypatch):
    """Round 5: the kernel compiled for the plan exists with 256, 512 and 1024 threads and picks by column
    height; across the Simpson passes of a stage the state waits in LDS, and the flux exchange runs after them.  Every
    workgroup size, the ahead-of-time kernel and round 4's bits (same expression order) agree bit for bit; a plan without
    thresholds (nothing parked) and a float-plane plan as well."""
    cloudy = gpu_cloudy
    L = cloudy.lib()
    for name, nz, ncol, dtype in (("cfg3b", 20, 131, np.float64), ("cfg3b", 100, 9, np.float64), ("cfg3a", 20, 57, np.float64),
                                  ("cfg3b", 33, 40, np.float32)):
        n = nz * ncol
        wl = bench.make_workload(name, n, seed=23)
        code = 0 if dtype == np.float64 else 1
        plans = {"jit": wl["coal_data"].plan(wl["dist_types"], vel=((50.0, 1.0 / 6),), dtype=code),
                 "aot": wl["coal_data"].plan(wl["dist_types"], vel=((50.0, 1.0 / 6),), dtype=code, specialize=-1)}
        assert plans["jit"].specialized and not plans["aot"].specialized
        u = cloudy.DeviceArray.from_numpy(wl["mom"].astype(dtype))
        res = {}
        for tag, plan, block in (("aot", plans["aot"], None), ("default", plans["jit"], None), ("256", plans["jit"], "256"),
                                 ("512", plans["jit"], "512"), ("1024", plans["jit"], "1024")):
            if block is None:
                monkeypatch.delenv("CLOUDY_HIP_RS_BLOCK", raising=False)
            else:
                monkeypatch.setenv("CLOUDY_HIP_RS_BLOCK", block)
            out = cloudy.DeviceArray.zeros(wl["mom"].shape[0], n, dtype)
            cloudy._lib.check(L.cloudy_rainshaft_ssprk33_steps(plan.handle, nz, ncol, n, u.ptr, out.ptr, 150.0, 1e-3, 2, None))
            res[tag] = out.to_numpy()
            if dtype == np.float64:   # (moments rounded to float planes hold degenerate cells: compared, not judged)
                assert np.isfinite(res[tag]).all() and res[tag].min() >= 0.0, (name, nz, tag)
        for tag in ("default", "256", "512", "1024"):
            assert np.array_equal(res[tag], res["aot"], equal_nan=True), (name, nz, tag)
        assert not np.array_equal(res["aot"], wl["mom"].astype(dtype))
    monkeypatch.delenv("CLOUDY_HIP_RS_BLOCK", raising=False)


def test_column_rhs_in_one_launch_matches_the_two_launch_path(gpu_cloudy, monkeypatch):
    """Round 5: cloudy_rainshaft_rhs -- make_rainshaft_rhs(...)'s rhs!, the drop-in boundary of the rainshaft drivers -- runs
    the RHS_ONLY instance of the column integrator's body when the plan has its kernels compiled for it and a workgroup holds a
    column: ONE launch (sources, fluxes, exchange through LDS, divergence) instead of the cell kernel + the divergence launch.
    fp64 planes: the same bits, rhs and fluxes, for every workgroup size; float planes: the one launch keeps fp64 between the
    source and the divergence where the two launches round the fluxes to float in between."""
    cloudy = gpu_cloudy
    L = cloudy.lib()
    for name, nz, ncol, dtype in (("cfg3b", 20, 131, np.float64), ("cfg3b", 100, 9, np.float64), ("cfg3a", 20, 57, np.float64),
                                  ("cfg4", 33, 21, np.float64), ("cfg3b", 700, 3, np.float64), ("cfg3b", 20, 64, np.float32)):
        n = nz * ncol
        wl = bench.make_workload(name, n, seed=37)
        code = 0 if dtype == np.float64 else 1
        plan = wl["coal_data"].plan(wl["dist_types"], vel=((50.0, 1.0 / 6),), dtype=code)
        mom = wl["mom"].astype(dtype)
        mom[:, 5] = -np.abs(mom[:, 5]) * 1e-20          # negative round-off is clamped inside, the input stays as it is
        u = cloudy.DeviceArray.from_numpy(mom)
        res = {}
        for tag, fused, block in (("two", "0", None), ("one", "1", None), ("one256", "1", "256"), ("one1024", "1", "1024")):
            monkeypatch.setenv("CLOUDY_HIP_RS_FUSED_RHS", fused)
            if block is None:
                monkeypatch.delenv("CLOUDY_HIP_RS_BLOCK", raising=False)
            elif nz > int(block):
                continue
            else:
                monkeypatch.setenv("CLOUDY_HIP_RS_BLOCK", block)
            rhs = cloudy.DeviceArray.zeros(mom.shape[0], n, dtype)
            flux = cloudy.DeviceArray.zeros(mom.shape[0], n, dtype)
            cloudy._lib.check(L.cloudy_rainshaft_rhs(plan.handle, nz, ncol, n, u.ptr, 150.0, flux.ptr, rhs.ptr, None))
            res[tag] = (rhs.to_numpy(), flux.to_numpy())
            assert np.array_equal(u.to_numpy(), mom, equal_nan=True)
        for tag in res:
            if tag == "two":
                continue
            if dtype == np.float64:
                assert np.array_equal(res[tag][0], res["two"][0], equal_nan=True), (name, nz, tag, "rhs")
                assert np.array_equal(res[tag][1], res["two"][1], equal_nan=True), (name, nz, tag, "flux")
            else:
                # rhs = coal + divergence is a difference of large terms: the scale of the rounding is |coal| + |fluxes| / dz
                coal = cloudy.DeviceArray.zeros(mom.shape[0], n, dtype)
                fl2 = cloudy.DeviceArray.zeros(mom.shape[0], n, dtype)
                cloudy._lib.check(L.cloudy_rainshaft_sources(plan.handle, n, n, u.ptr, coal.ptr, fl2.ptr, None))
                fl = np.abs(res["two"][1]).astype(np.float64)
                fup = np.concatenate([fl[:, 1:], np.zeros((fl.shape[0], 1))], axis=1)
                ref = np.abs(coal.to_numpy()).astype(np.float64) + (fl + fup) / 150.0
                ok = np.isfinite(res["two"][0]) & np.isfinite(ref)
                assert np.array_equal(res[tag][1], res["two"][1], equal_nan=True)
                err = np.abs(res[tag][0].astype(np.float64) - res["two"][0].astype(np.float64))
                assert np.all(err[ok] <= 2.5e-7 * ref[ok] + 1e-38), (name, nz, tag, float(np.max(err[ok] / (ref[ok] + 1e-300))))
        assert np.isfinite(res["one"][0]).mean() > 0.9 and np.abs(res["one"][0][np.isfinite(res["one"][0])]).max() > 0.0
    monkeypatch.delenv("CLOUDY_HIP_RS_BLOCK", raising=False)
    monkeypatch.delenv("CLOUDY_HIP_RS_FUSED_RHS", raising=False)


def test_tall_columns_replay_a_captured_step(gpu_cloudy, monkeypatch):
    """Columns of more than 1024 cells are stepped stage by stage inside the library (nine launches per step).  Experiment
    switch CLOUDY_HIP_GRAPH=1 (round 5; measured slower than the eager loop on ROCm 7.2, so off by default): from the second
    step on ONE step is captured into a hipGraph and replayed.  Same bits as the eager loop, on the legacy NULL stream and on
    a stream of the caller, out of place and in place; the caller's stream is ordered after the call's own stream (the result
    is complete when the caller's stream has drained)."""
    import ctypes
    import time

    cloudy = gpu_cloudy
    L = cloudy.lib()
    nz, ncol, n_steps = 1500, 3, 12
    n = nz * ncol
    wl = bench.make_workload("cfg3b", n, seed=31)
    plan = wl["coal_data"].plan(wl["dist_types"], vel=((50.0, 1.0 / 6),))
    hip = ctypes.CDLL("libamdhip64.so")
    user = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(user)) == 0
    res, ms = {}, {}
    for tag, graph, stream in (("eager", "0", None), ("graph", "1", None), ("graph_user_stream", "1", user), ("eager_user_stream", "0", user)):
        monkeypatch.setenv("CLOUDY_HIP_GRAPH", graph)
        u = cloudy.DeviceArray.from_numpy(wl["mom"])
        out = cloudy.DeviceArray.zeros(*wl["mom"].shape)
        for rep in range(2):   # (the second call is timed: kernels compiled, caches warm)
            t0 = time.perf_counter()
            cloudy._lib.check(L.cloudy_rainshaft_ssprk33_steps(plan.handle, nz, ncol, n, u.ptr, out.ptr, 150.0, 1e-3, n_steps, stream))
            cloudy._lib.check(L.cloudy_stream_synchronize(stream))
            ms[tag] = (time.perf_counter() - t0) * 1e3
        res[tag] = out.to_numpy()
        assert np.isfinite(res[tag]).mean() > 0.9 and not np.array_equal(res[tag], wl["mom"])   # (the batch holds degenerate cells, and a NaN travels down its column)
    for tag in ("graph", "graph_user_stream", "eager_user_stream"):
        assert np.array_equal(res[tag], res["eager"], equal_nan=True), tag
    monkeypatch.setenv("CLOUDY_HIP_GRAPH", "1")
    u = cloudy.DeviceArray.from_numpy(wl["mom"])
    cloudy._lib.check(L.cloudy_rainshaft_ssprk33_steps(plan.handle, nz, ncol, n, u.ptr, u.ptr, 150.0, 1e-3, n_steps, None))   # in place
    assert np.array_equal(u.to_numpy(), res["eager"], equal_nan=True)
    monkeypatch.delenv("CLOUDY_HIP_GRAPH", raising=False)
    assert hip.hipStreamDestroy(user) == 0
    print(f"{ncol} columns x {nz} cells, {n_steps} steps: eager {ms['eager']:.2f} ms, captured step replayed {ms['graph']:.2f} ms")


def _rhs_with_plan(cloudy, plan, mom, dtype=np.float64):
    m = cloudy.DeviceArray.from_numpy(mom.astype(dtype))
    dm = cloudy.DeviceArray.zeros(mom.shape[0], mom.shape[1], dtype)
    cloudy._lib.check(cloudy.lib().cloudy_coal_rhs(plan.handle, mom.shape[1], mom.shape[1], m.ptr, dm.ptr, None))
    return dm.to_numpy()


def test_plan_time_specialised_kernels_are_bit_identical(gpu_cloudy, oracle):
    """Plans whose thresholds are all Inf get kernels compiled for them at plan creation (hiprtc; jit.hpp) with the
    tensors, norms and closure types as compile-time constants: zero coefficients drop out.  Same bits as the
    ahead-of-time kernels -- for the headline workload, float planes, the one-parcel fallback and the fused SSPRK33."""
    cloudy = gpu_cloudy
    L = cloudy.lib()
    n = 100_003
    wl = bench.make_workload("cfg3a", n, seed=23)
    cd, dts = wl["coal_data"], wl["dist_types"]
    spec, gen = cd.plan(dts, specialize=1), cd.plan(dts, specialize=-1)
    assert spec.specialized and not gen.specialized and "off" in gen.jit_log()
    fin = np.all(np.isfinite(_rhs_with_plan(cloudy, gen, wl["mom"])), axis=0)
    assert fin.mean() > 0.98
    a, b = _rhs_with_plan(cloudy, spec, wl["mom"]), _rhs_with_plan(cloudy, gen, wl["mom"])
    assert np.array_equal(a[:, fin], b[:, fin])
    want, scale = oracle.rhs_coal_batch(bench.oracle_params("cfg3a"), wl["mom"], with_scale=True)
    assert_close_scaled(a, want, scale, TOL_POLY, "specialised cfg3a")
    # float planes
    s32, g32 = cd.plan(dts, dtype=1, specialize=1), cd.plan(dts, dtype=1, specialize=-1)
    a32 = _rhs_with_plan(cloudy, s32, wl["mom"], np.float32)
    b32 = _rhs_with_plan(cloudy, g32, wl["mom"], np.float32)
    f32 = np.all(np.isfinite(b32), axis=0)
    assert s32.specialized and np.array_equal(a32[:, f32], b32[:, f32])
    # odd leading dimension: the one-parcel-per-lane kernel of the specialised module
    ld = n + 1 + (n % 2)
    buf = np.zeros((6, ld))
    buf[:, :n] = wl["mom"]
    m, dm = dev(cloudy, buf), dev(cloudy, np.zeros((6, ld)))
    assert ld % 2 == 1 and L.cloudy_coal_rhs(spec.handle, n, ld, m.ptr, dm.ptr, None) == 0
    assert np.array_equal(dm.to_numpy()[:, :n][:, fin], b[:, fin])
    # fused SSPRK33
    reg = np.flatnonzero(fin)[:50_000]
    u0 = np.ascontiguousarray(wl["mom"][:, reg])
    outs = []
    for plan in (spec, gen):
        u, o = dev(cloudy, u0), cloudy.DeviceArray.zeros(6, u0.shape[1])
        cloudy._lib.check(L.cloudy_ssprk33_steps(plan.handle, u0.shape[1], u0.shape[1], u.ptr, o.ptr, 1e-3, 3, None))
        outs.append(o.to_numpy())
    ok = np.all(np.isfinite(outs[1]), axis=0)
    assert ok.mean() > 0.9 and np.array_equal(outs[0][:, ok], outs[1][:, ok])
    # a second plan with the same constants shares the compiled module (instant)
    assert cloudy.Plan(dts, cd.kernel_c, cd.dist_thresholds_in, cd.norms, 0, specialize=1).specialized


@pytest.mark.parametrize("name,moving,dtype", [("cfg3b", False, 0), ("cfg3b", False, 2), ("cfg3b", True, 0),
                                               ("cfg4", False, 0)])
def test_specialised_threshold_kernels_are_bit_identical(gpu_cloudy, oracle, name, moving, dtype):
    """Plans with a threshold get the regime-sorted kernel compiled for them too (tensors, norms, threshold and node
    counts as constants; the node table stays in memory): same bits as the ahead-of-time kernel."""
    cloudy = gpu_cloudy
    n = 20_011
    wl = bench.make_workload(name, n, seed=29)
    dts = wl["dist_types"]
    if moving:
        N = len(dts)
        kc = wl["kernel_c"]
        kernels = tuple(tuple(cloudy.CoalescenceTensor(kc[j, k]) for k in range(N)) for j in range(N))
        cd = cloudy.CoalescenceData(kernels, wl["NProgMoms"], (0.9,) * (N - 1) + (1.0,), bench.NORMS,
                                    cloudy.MovingThreshold())
    else:
        cd = wl["coal_data"]
    spec, gen = cd.plan(dts, dtype=dtype, specialize=1), cd.plan(dts, dtype=dtype, specialize=-1)
    assert spec.specialized and not gen.specialized
    tio = np.float64 if dtype == 0 else np.float32
    a, b = _rhs_with_plan(cloudy, spec, wl["mom"], tio), _rhs_with_plan(cloudy, gen, wl["mom"], tio)
    fin = np.all(np.isfinite(b), axis=0)
    assert fin.mean() > 0.95 and np.array_equal(a[:, fin], b[:, fin])
    assert np.array_equal(np.isnan(a), np.isnan(b))
    # the rainshaft cell body (clamp + empty-cell rule) has its own specialised kernel, compiled on first use
    if not moving:
        vel = ((50.0, 1.0 / 6),)
        sv, gv = cd.plan(dts, vel=vel, dtype=dtype, specialize=1), cd.plan(dts, vel=vel, dtype=dtype, specialize=-1)
        mneg = wl["mom"].astype(tio).copy()
        mneg[:, ::97] *= -1.0                      # negative moments are clamped to zero
        mneg[:, 5::101] = 0.0                      # empty cells skip coalescence
        res = [cloudy.rainshaft_sources(pl, cloudy.DeviceArray.from_numpy(mneg)) for pl in (sv, gv)]
        for x, y in zip(res[0], res[1]):
            x, y = x.to_numpy(), y.to_numpy()
            f2 = np.isfinite(y)
            assert np.array_equal(x[f2], y[f2])
    # the fused integrator of a thresholded plan is compiled on its first call
    reg = np.flatnonzero(fin)[:4096]
    u0 = np.ascontiguousarray(wl["mom"][:, reg]).astype(tio)
    outs = []
    for plan in (spec, gen):
        u, o = cloudy.DeviceArray.from_numpy(u0), cloudy.DeviceArray.zeros(u0.shape[0], u0.shape[1], tio)
        cloudy._lib.check(cloudy.lib().cloudy_ssprk33_steps(plan.handle, u0.shape[1], u0.shape[1], u.ptr, o.ptr, 1e-3, 2,
                                                            None))
        outs.append(o.to_numpy())
    ok = np.all(np.isfinite(outs[1]), axis=0)
    # (the explicit scheme blows up on some parcels.)  Same arithmetic, but -ffp-contract=fast may fuse multiply-adds
    # differently in the two compilations of the stage loop: agreement to rounding, not bit for bit
    assert ok.mean() > 0.5
    ref = np.abs(outs[1][:, ok]).max(axis=1, keepdims=True)
    assert np.all(np.abs(outs[0][:, ok] - outs[1][:, ok]) <= 1e-12 * ref)


@pytest.mark.parametrize("N,P", [(1, 1), (1, 2), (2, 2), (3, 3), (4, 2), (2, 5), (4, 5)])
def test_specialised_kernels_across_families(gpu_cloudy, oracle, N, P):
    """Random sparse symmetric tensors, mixed Exponential / Gamma modes: specialised == ahead-of-time, bit for bit,
    and both within tolerance of the oracle."""
    cloudy = gpu_cloudy
    rng = np.random.default_rng(100 * N + P)
    kc = np.zeros((N, N, P, P))
    for j in range(N):
        for k in range(j, N):
            c = rng.uniform(0.1, 1.0, (P, P)) * (rng.random((P, P)) < 0.45)
            c = np.triu(c) + np.triu(c, 1).T
            kc[j, k] = kc[k, j] = c * 1e-3 * (1e9 ** np.add.outer(np.arange(P), np.arange(P)))
    dist_types = [int(t) for t in rng.integers(0, 2, N)]
    par, op, _ = make_case(cloudy, oracle, dist_types, kc, (INF,) * N, bench.NORMS)
    n = 20_001
    mom = mixed_moments(dist_types, n, seed=N * 10 + P)
    spec = par.coal_data.plan(dist_types, specialize=1)
    gen = par.coal_data.plan(dist_types, specialize=-1)
    a, b = _rhs_with_plan(cloudy, spec, mom), _rhs_with_plan(cloudy, gen, mom)
    fin = np.all(np.isfinite(b), axis=0)
    assert spec.specialized and fin.mean() > 0.95 and np.array_equal(a[:, fin], b[:, fin])
    want, scale = oracle.rhs_coal_batch(op, mom, with_scale=True)
    assert_close_scaled(a, want, scale, TOL_POLY, f"specialised N={N} P={P}")


def test_code_object_cache_on_disk(gpu_cloudy, tmp_path, monkeypatch):
    """Compiled plan kernels are kept between processes (CLOUDY_HIP_CACHE_DIR): a new plan writes its code object
    there, and an empty value switches the cache off."""
    cloudy = gpu_cloudy
    d = tmp_path / "cc"
    monkeypatch.setenv("CLOUDY_HIP_CACHE_DIR", str(d))
    kc = [[0.0, 3.25], [3.25, 0.0]]                      # constants no other test uses: not yet in the process cache
    plan = cloudy.Plan([1], kc, (INF,), (1e6, 1e-9), 0, specialize=1)
    files = list(d.glob("*.co"))
    assert plan.specialized and len(files) == 1 and files[0].stat().st_size > 1000
    assert files[0].read_bytes()[:4] == b"\x7fELF"
    monkeypatch.setenv("CLOUDY_HIP_CACHE_DIR", "")
    plan2 = cloudy.Plan([1], [[0.0, 3.5], [3.5, 0.0]], (INF,), (1e6, 1e-9), 0, specialize=1)
    assert plan2.specialized and len(list(d.glob("*.co"))) == 1


def test_damaged_cache_entry_is_recompiled(gpu_cloudy, tmp_path):
    """A cached code object that the runtime cannot load (another ROCm patch level, a damaged file) must not drop the plan
    to the ahead-of-time kernels for ever: the entry is removed, compiled again and rewritten.  A second process (the
    in-process module cache would hide it) finds the damaged file."""
    import subprocess
    import sys

    d = tmp_path / "cc"
    script = (
        "import sys; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as ge\n"
        "cl = ge.load_package()\n"
        "p = cl.Plan([1], [[0.0, 3.75], [3.75, 0.0]], (float('inf'),), (1e6, 1e-9), 0, specialize=1)\n"
        "print('specialized', p.specialized)\n" % ROOT)
    env = dict(os.environ, CLOUDY_HIP_CACHE_DIR=str(d))
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "specialized True" in r.stdout, r.stdout + r.stderr
    (f,) = list(d.glob("*.co"))
    good = f.read_bytes()
    f.write_bytes(good[:64] + b"\0" * (len(good) - 64))        # ELF magic intact, everything else zeroed
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "specialized True" in r.stdout, r.stdout + r.stderr
    assert f.read_bytes() == good or f.read_bytes()[:4] == b"\x7fELF" and f.read_bytes() != good[:64] + b"\0" * (len(good) - 64)


def test_more_than_2_to_32_elements_per_array(gpu_cloudy):
    """Sizing for 288 GB: 8e8 parcels x 6 planes = 4.8e9 elements per array (38 GB in, 38 GB out) in one launch; the
    batch repeats a 2^20-parcel tile, so every tile of the output must equal the first (tools/big_batch_check.py)."""
    import subprocess
    import sys

    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_batch_check.py"), "800000000"],
                       capture_output=True, text=True, timeout=600)
    if p.returncode == 77:  # EXIT_NOMEM of the script: cloudy_malloc returned CLOUDY_ENOMEM, nothing else skips
        pytest.skip("not enough free device memory for the 77 GB check")
    assert p.returncode == 0, p.stdout + p.stderr
    assert "mismatching tiles: 0" in p.stdout
    print(p.stdout.strip())


@pytest.mark.parametrize("seed", [7, 11])
def test_randomised_plan_sweep(gpu_cloudy, oracle, seed):
    """tools/fuzz_parity.py: random plans (N, P, closure families, fixed / moving thresholds, sparse symmetric tensors,
    plane types) -- plan-time compiled and ahead-of-time kernels against the oracle and against each other."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity as F

    rng = np.random.default_rng(seed)
    for c in range(10):
        cfg = F.random_config(rng)
        worst, dj = F.check_config(gpu_cloudy, cfg, 300, 5000 + 100 * seed + c)
        assert dj <= 1e-6


def test_closure_stats_vs_oracle(gpu_cloudy, oracle):
    """cloudy_closure_stats: per mode the number of parcels on the fallback distribution (0, 1, 1), with the
    shape at its lower / upper clamp, and failing check_moment_consistency (ParticleDistributions.jl:437-449, 456-541) -- one pass
    over the batch -- against the oracle's count, exactly (integers): the bench batch at 1e6 parcels (its ~1 % degenerate parcels
    make every counter non-zero), a four-mode MovingThreshold plan, a plan with Exponential / Lognormal modes and another shape
    range, a slice with its own leading dimension, float planes, and an empty batch."""
    cloudy = gpu_cloudy
    O = oracle
    wl = bench.make_workload("cfg3b", 1_000_000, seed=3)
    plan = wl["coal_data"].plan(wl["dist_types"])
    m = dev(cloudy, wl["mom"])
    got = cloudy.closure_stats(plan, m)
    want = O.closure_stats(bench.oracle_params("cfg3b"), wl["mom"])
    assert got.shape == (2, 4) and np.array_equal(got, want) and want.min() > 0, (got, want)
    # the counts say what the parameter planes say
    par = cloudy.update_dist_from_moments(plan, m).to_numpy()
    assert got[0, 0] == ((par[0] == 0.0) & (par[1] == 1.0) & (par[2] == 1.0)).sum()
    assert got[1, 2] == (par[5] == 10.0).sum()
    # a slice of the batch with the batch's leading dimension
    import ctypes as C

    L = cloudy.lib()
    buf = (C.c_uint64 * 8)()
    lo, cnt, esz = 77_777, 123_457, 8
    cloudy._lib.check(L.cloudy_closure_stats(plan.handle, cnt, 1_000_000, m.ptr + lo * esz, buf, None))
    assert np.array_equal(np.array(buf[:]).reshape(2, 4), O.closure_stats(bench.oracle_params("cfg3b"), wl["mom"][:, lo:lo + cnt]))
    cloudy._lib.check(L.cloudy_closure_stats(plan.handle, 0, 0, None, buf, None))
    assert not any(buf[:])
    assert L.cloudy_closure_stats(plan.handle, 10, 5, m.ptr, buf, None) == cloudy._lib.EINVAL
    assert L.cloudy_closure_stats(plan.handle, 10, 10, m.ptr, None, None) == cloudy._lib.EINVAL
    # four modes, MovingThreshold
    w4 = bench.make_workload("moving4", 200_000, seed=4)
    p4 = w4["coal_data"].plan(w4["dist_types"])
    m4, want4 = dev(cloudy, w4["mom"]), O.closure_stats(bench.oracle_params("moving4"), w4["mom"])
    assert np.array_equal(cloudy.closure_stats(p4, m4), want4)
    # (round 6: the first version -- a zero-fill and one atomic per wave -- came back short on this case on one box of the pool,
    # the first two modes' counters 105 and 302 of 469; a row of counters per workgroup now: the same integers every time)
    for _ in range(40):
        assert np.array_equal(cloudy.closure_stats(p4, m4), want4)
    # Exponential + Gamma + Lognormal, k_range (0.5, 5): two-moment modes have no central-moment check, a Lognormal mode clamps sigma
    dt = [0, 1, 3]
    par3, op3, _ = make_case(cloudy, O, dt, [[EPS / 1e6, 5.0], [5.0, 0.0]], (INF, INF, INF), bench.NORMS, k_range=(0.5, 5.0))
    mom3 = mixed_moments(dt, 150_000, seed=12)
    rng = np.random.default_rng(8)
    bad = rng.choice(150_000, 3000, replace=False)
    mom3[0, bad[:500]] *= -1.0                       # a negative moment
    mom3[7, bad[500:1500]] = 0.9 * mom3[6, bad[500:1500]] ** 2 / mom3[5, bad[500:1500]]   # Lognormal: M0 M2 < M1^2 -> sigma = eps
    mom3[5, bad[1500:2000]] = 0.0                    # empty Lognormal mode
    mom3[4, bad[2000:]] = 4.0 * mom3[3, bad[2000:]] ** 2 / mom3[2, bad[2000:]]          # a wide Gamma mode: k below k_min
    plan3 = par3.coal_data.plan(dt, k_range=(0.5, 5.0))
    g3 = cloudy.closure_stats(plan3, dev(cloudy, mom3))
    w3 = O.closure_stats(op3, mom3)
    assert np.array_equal(g3, w3), (g3, w3)
    # (a few of the doctored parcels coincide with the batch's own degenerate ones)
    assert w3[0, 3] >= 450 and w3[2, 1] >= 800 and w3[2, 0] >= 450 and w3[1, 1] >= 800 and w3[0, 1] == 0 and w3[0, 2] == 0, w3
    # float planes: the counts of the ROUNDED moments (the oracle sees the same float values)
    pf = wl["coal_data"].plan(wl["dist_types"], dtype=1)
    mf = wl["mom"].astype(np.float32)
    gf = cloudy.closure_stats(pf, cloudy.DeviceArray.from_numpy(mf))
    assert np.array_equal(gf, O.closure_stats(bench.oracle_params("cfg3b"), mf.astype(np.float64)))


def test_first_calls_from_two_host_threads_on_one_plan(gpu_cloudy):
    """VERDICT r5 item 5 / ADVICE r5 (medium): a plan is shared by host threads.  (i) The FIRST calls of two entry points that each
    build something on first use -- cloudy_tsit5_steps compiles its kernel, a diagnostic its own -- from two threads at once on one
    thresholded plan (ctypes releases the GIL: the calls overlap in the library); (ii) a converged-mode plan whose hint bytes are
    GROWN by one thread (a larger batch) while the other keeps launching on the smaller one -- round 5 freed the superseded buffer
    under the first thread's launch.  Results equal the single-threaded ones bit for bit, on every repetition."""
    import ctypes as C
    import threading

    cloudy = gpu_cloudy
    L = cloudy.lib()
    # ---- (i)
    wl = bench.make_workload("cfg3b", 60_000, seed=21)
    plan = wl["coal_data"].plan(wl["dist_types"], k_range=(EPS, 9.5))     # (a plan no other test has warmed up)
    m = dev(cloudy, wl["mom"])
    o5, prm = cloudy.DeviceArray.zeros(6, 60_000), cloudy.DeviceArray.zeros(6, 60_000)
    errs = []

    def run(fn):
        try:
            fn()
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    t1 = threading.Thread(target=run, args=(lambda: cloudy._lib.check(
        L.cloudy_tsit5_steps(plan.handle, 60_000, 60_000, m.ptr, o5.ptr, C.c_double(1e-4), 1, None)),))
    t2 = threading.Thread(target=run, args=(lambda: [cloudy.closure_stats(plan, m),
                                                     cloudy._lib.check(L.cloudy_update_dist_from_moments(plan.handle, 60_000, 60_000, m.ptr, prm.ptr, None))],))
    t1.start(); t2.start(); t1.join(); t2.join()
    assert not errs, errs
    cloudy._lib.check(L.cloudy_stream_synchronize(None))
    ref5, refp = cloudy.DeviceArray.zeros(6, 60_000), cloudy.DeviceArray.zeros(6, 60_000)
    cloudy._lib.check(L.cloudy_tsit5_steps(plan.handle, 60_000, 60_000, m.ptr, ref5.ptr, C.c_double(1e-4), 1, None))
    cloudy._lib.check(L.cloudy_update_dist_from_moments(plan.handle, 60_000, 60_000, m.ptr, refp.ptr, None))
    cloudy._lib.check(L.cloudy_stream_synchronize(None))
    assert np.array_equal(o5.to_numpy(), ref5.to_numpy(), equal_nan=True) and np.array_equal(prm.to_numpy(), refp.to_numpy(), equal_nan=True)
    # ---- (ii)
    kfn = cloudy.get_normalized_kernel_func(cloudy.HydrodynamicKernelFunction(1e2 * np.pi), bench.NORMS)
    cplan = cloudy.NumericalPlan([1, 1], kfn, bench.NORMS, 8, specialize=1, quad_mode=cloudy.QUAD_CONVERGED)
    small, sizes = 20_000, (30_000, 70_000, 150_000, 320_000)
    ms = dev(cloudy, bench.synth_moments(2, small, seed=5))
    ds = cloudy.DeviceArray.zeros(6, small)
    cloudy._lib.check(L.cloudy_coal_rhs(cplan.handle, small, small, ms.ptr, ds.ptr, None))
    cloudy._lib.check(L.cloudy_stream_synchronize(None))
    want_small = ds.to_numpy().copy()
    bigs = [(n, dev(cloudy, bench.synth_moments(2, n, seed=6)), cloudy.DeviceArray.zeros(6, n)) for n in sizes]
    outs = [cloudy.DeviceArray.zeros(6, small) for _ in range(24)]

    def hammer():
        for o in outs:
            cloudy._lib.check(L.cloudy_coal_rhs(cplan.handle, small, small, ms.ptr, o.ptr, None))

    def grow():
        for n, mb, db in bigs:
            cloudy._lib.check(L.cloudy_coal_rhs(cplan.handle, n, n, mb.ptr, db.ptr, None))

    ta, tb = threading.Thread(target=run, args=(hammer,)), threading.Thread(target=run, args=(grow,))
    ta.start(); tb.start(); ta.join(); tb.join()
    cloudy._lib.check(L.cloudy_stream_synchronize(None))
    assert not errs, errs
    for o in outs:
        assert np.array_equal(o.to_numpy(), want_small, equal_nan=True)
    fresh = cloudy.NumericalPlan([1, 1], kfn, bench.NORMS, 8, specialize=1, quad_mode=cloudy.QUAD_CONVERGED)
    n, mb, db = bigs[-1]
    chk = cloudy.DeviceArray.zeros(6, n)
    cloudy._lib.check(L.cloudy_coal_rhs(fresh.handle, n, n, mb.ptr, chk.ptr, None))
    cloudy._lib.check(L.cloudy_stream_synchronize(None))
    assert np.array_equal(db.to_numpy(), chk.to_numpy(), equal_nan=True)


def test_error_returns_of_the_column_and_integrator_entry_points(gpu_cloudy):
    """Argument checking of the newer entry points: status codes, never exceptions or launches."""
    cloudy = gpu_cloudy
    L, E = cloudy.lib(), cloudy._lib
    kc = [[0.0, 5.0], [5.0, 0.0]]
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor(kc), (3,), (INF,), bench.NORMS)
    no_vel = cd.plan([1])
    with_vel = cd.plan([1], vel=((50.0, 1.0 / 6),))
    u = cloudy.DeviceArray.zeros(3, 40)
    # no terminal velocity in the plan
    assert L.cloudy_rainshaft_ssprk33_steps(no_vel.handle, 20, 2, 40, u.ptr, u.ptr, 150.0, 1.0, 1, None) == E.EINVAL
    assert b"n_vel" in L.cloudy_last_error()
    # bad geometry / step arguments
    assert L.cloudy_rainshaft_ssprk33_steps(with_vel.handle, 0, 2, 40, u.ptr, u.ptr, 150.0, 1.0, 1, None) == E.EINVAL
    assert L.cloudy_rainshaft_ssprk33_steps(with_vel.handle, 20, 2, 40, u.ptr, u.ptr, -1.0, 1.0, 1, None) == E.EINVAL
    assert L.cloudy_rainshaft_ssprk33_steps(with_vel.handle, 20, 2, 40, u.ptr, u.ptr, 150.0, 1.0, -3, None) == E.EINVAL
    assert L.cloudy_rainshaft_ssprk33_steps(with_vel.handle, 20, 2, 39, u.ptr, u.ptr, 150.0, 1.0, 1, None) == E.EINVAL  # ld < n
    assert L.cloudy_rainshaft_ssprk33_steps(with_vel.handle, 20, 2, 40, None, u.ptr, 150.0, 1.0, 1, None) == E.EINVAL
    assert L.cloudy_rainshaft_ssprk33_steps(None, 20, 2, 40, u.ptr, u.ptr, 150.0, 1.0, 1, None) == E.EINVAL
    # columns taller than the largest workgroup of the fused integrator (1024 cells) are stepped stage by stage (round 5)
    ut = cloudy.DeviceArray.zeros(3, 1030)
    assert L.cloudy_rainshaft_ssprk33_steps(with_vel.handle, 1030, 1, 1030, ut.ptr, ut.ptr, 150.0, 1.0, 1, None) == 0
    assert np.array_equal(ut.to_numpy(), np.zeros((3, 1030)))
    # an empty batch is fine
    assert L.cloudy_rainshaft_ssprk33_steps(with_vel.handle, 20, 0, 0, None, None, 150.0, 1.0, 1, None) == 0
    # MovingThreshold plans have no rainshaft driver in the reference
    cdm = cloudy.CoalescenceData(tuple(tuple(cloudy.CoalescenceTensor(kc) for _ in range(2)) for _ in range(2)), (3, 3),
                                 (0.9, 1.0), bench.NORMS, cloudy.MovingThreshold())
    pm = cdm.plan([1, 1], vel=((50.0, 1.0 / 6),))
    u6 = cloudy.DeviceArray.zeros(6, 40)
    assert L.cloudy_rainshaft_ssprk33_steps(pm.handle, 20, 2, 40, u6.ptr, u6.ptr, 150.0, 1.0, 1, None) == E.EINVAL
    # ... whichever path would serve the call (a single-mode MovingThreshold plan has no threshold pass, and the
    # one-launch column RHS of round 5 accepted it where the two-launch path returns EINVAL)
    cd1 = cloudy.CoalescenceData(cloudy.CoalescenceTensor(kc), (3,), (1.0,), bench.NORMS, cloudy.MovingThreshold())
    pm1 = cd1.plan([1], vel=((50.0, 1.0 / 6),))
    f3, w3 = cloudy.DeviceArray.zeros(3, 40), cloudy.DeviceArray.zeros(3, 40)
    for plan_m, arr in ((pm1, u), (pm, u6)):
        fw, rw = (f3, w3) if plan_m is pm1 else (cloudy.DeviceArray.zeros(6, 40), cloudy.DeviceArray.zeros(6, 40))
        assert L.cloudy_rainshaft_rhs(plan_m.handle, 20, 2, 40, arr.ptr, 150.0, fw.ptr, rw.ptr, None) == E.EINVAL
        assert b"FixedThreshold" in L.cloudy_last_error()
    assert L.cloudy_rainshaft_rhs(no_vel.handle, 20, 2, 40, u.ptr, 150.0, f3.ptr, w3.ptr, None) == E.EINVAL
    assert L.cloudy_rainshaft_rhs(with_vel.handle, 20, 2, 40, u.ptr, 150.0, None, w3.ptr, None) == E.EINVAL
    assert L.cloudy_ssprk33_steps(pm.handle, 40, 40, u6.ptr, u6.ptr, float("nan"), 1, None) == E.EINVAL
    assert L.cloudy_ssprk33_steps(pm.handle, 40, 40, u6.ptr, u6.ptr, 1.0, -1, None) == E.EINVAL
    assert L.cloudy_plan_specialized(None) == E.EINVAL and L.cloudy_plan_jit_log(None) == b"plan is NULL"
    assert L.cloudy_jit_selfcheck(None, None) == E.EINVAL
