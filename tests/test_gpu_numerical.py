"""GPU parity of the NumericalCoalStyle (fixed Gauss rule) plans, through the C ABI, against the same-rule CPU oracle
(oracle/cloudy_oracle_quad.c) -- BASELINE configs[3] as worded: "3-mode mixture, hydrodynamic kernel via 10-pt Gauss
quadrature".

Tolerances (north_star): <= 1e-8 x scale against the same rule on the CPU for quadrature kernels (measured ~1e-14:
asserted at 1e-11 so that a regression shows); <= 1e-12 x scale against the ANALYTIC tensor path on the device for
polynomial kernels where the two closures coincide (one mode).  `scale` = sum of |Q|, |R|, |S| terms of an output as the
oracle reports it.  The discretisation error of the rule against adaptive quadrature is a CPU test
(tests/test_numerical_oracle.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import bench
from test_gpu_parity import INF, _ssprk33_host, assert_close_scaled, dev, mixed_moments

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOL_SAME_RULE = 1e-11   # north_star allows 1e-8; the two implementations of the rule agree to ~1e-14
NORMS = bench.NORMS


def assert_same_rule(got, want, scale, noise, what, tol=TOL_SAME_RULE):
    """|hip - oracle| <= tol * scale + 8 * noise.  `noise` is the rounding error of the ORACLE's (= the reference's)
    `(1 - weighting_fn) * inner` form of S_2 (Coalescence.jl:703-708): where a mode barely overlaps the next one the
    difference 1 - w keeps only a few digits, while the kernel forms 1 - w directly from density ratios."""
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN pattern differs"
    ok = ~np.isnan(want)
    err = np.abs(got[ok] - want[ok])
    excess = err - 8.0 * noise[ok]
    worst = float((np.maximum(excess, 0.0) / np.maximum(scale[ok], 1e-300)).max()) if err.size else 0.0
    assert worst <= tol, f"{what}: max (|diff| - 8 noise) / scale = {worst:.3e} > {tol:g}"
    return worst


def kernel_funcs(cloudy, oracle):
    return {
        "constant": (cloudy.ConstantKernelFunction(1e-4), oracle.kernel_func(oracle.KF_CONSTANT, 1e-4)),
        "linear": (cloudy.LinearKernelFunction(5.0), oracle.kernel_func(oracle.KF_LINEAR, 5.0)),
        "hydro": (cloudy.HydrodynamicKernelFunction(1e2 * np.pi), oracle.kernel_func(oracle.KF_HYDRODYNAMIC, 1e2 * np.pi)),
        "long": (cloudy.LongKernelFunction(5.236e-10, 9.44e9, 5.78), oracle.kernel_func(oracle.KF_LONG, 5.236e-10, 9.44e9, 5.78)),
    }


def numerical_case(cloudy, oracle, dist_types, kname, nq=10, k_range=None):
    """product-side ODE parameters (p.kernel_func normalised, as the reference's drivers pass it) + oracle params"""
    kf, okf = kernel_funcs(cloudy, oracle)[kname]
    mk = {0: lambda: cloudy.ExponentialPrimitiveParticleDistribution(1.0, 1.0),
          1: lambda: cloudy.GammaPrimitiveParticleDistribution(1.0, 1.0, 1.0),
          3: lambda: cloudy.LognormalPrimitiveParticleDistribution(1.0, 1.0, 1.0)}
    pd = tuple(mk[t]() for t in dist_types)
    npm = tuple({0: 2, 1: 3, 3: 3}[t] for t in dist_types)
    extra = {"k_range": k_range} if k_range else {}
    par = cloudy.ODEParameters(pd, None, npm, NORMS, kernel_func=cloudy.get_normalized_kernel_func(kf, NORMS),
                               quad_order=nq, quad_mode=cloudy.QUAD_FIXED, **extra)
    op = oracle.make_params(list(dist_types), np.zeros((1, 1)), (INF,) * len(dist_types), norms=NORMS,
                            **({"k_range": k_range} if k_range else {}))
    return par, op, oracle.get_normalized_kernel_func(okf, NORMS)


def run_numerical(cloudy, par, mom):
    rhs = cloudy.make_box_model_rhs(cloudy.NumericalCoalStyle())
    m = dev(cloudy, mom)
    dm = cloudy.DeviceArray.zeros(*mom.shape, dtype=mom.dtype)
    rhs(dm, m, par, 0.0)
    return dm.to_numpy()


def test_cfg4_three_modes_hydrodynamic_kernel_10pt_rule(gpu_cloudy, oracle):
    """BASELINE configs[3]: 3 Gamma modes (bench.synth_moments incl. ~1 % degenerate parcels), HydrodynamicKernelFunction
    (1e2 pi) (box_gamma_mixture_hydro.jl:22), 10-point rule, 9 moments -- against the same rule on the CPU"""
    cloudy = gpu_cloudy
    par, op, okf = numerical_case(cloudy, oracle, [1, 1, 1], "hydro")
    mom = bench.synth_moments(3, 6000, seed=41)
    got = run_numerical(cloudy, par, mom)
    want, scale, noise = oracle.rhs_coal_numerical_batch(op, okf, 10, mom, with_noise=True)
    worst = assert_same_rule(got, want, scale, noise, "cfg4q")
    print(f"cfg4q (3 Gamma modes, hydrodynamic, nq = 10): max |hip - oracle| / scale = {worst:.2e}")
    ok = np.all(np.isfinite(got), axis=0)
    mass = got[1] + got[4] + got[7]
    assert np.all(np.abs(mass[ok]) <= 1e-11 * (scale[1] + scale[4] + scale[7])[ok])     # mass is conserved by the rule
    assert np.all((got[0] + got[3] + got[6])[ok] <= 0.0)                               # coalescence only removes particles


@pytest.mark.parametrize("dist_types,kname,nq", [
    ([1], "hydro", 10), ([1], "long", 10), ([0], "linear", 10), ([3], "hydro", 10),
    ([1, 1], "hydro", 10), ([1, 1], "long", 10), ([0, 1], "hydro", 10), ([1, 0], "linear", 6), ([0, 0], "long", 10),
    ([1, 3], "hydro", 10), ([3, 3], "linear", 8),
    ([1, 1, 1], "long", 10), ([1, 1, 1], "constant", 4), ([1, 0, 1], "hydro", 12), ([1, 1, 1], "hydro", 16),
    ([1, 1, 1, 1], "hydro", 10), ([0, 1, 1, 1], "long", 8), ([1, 1], "hydro", 32), ([1, 1], "hydro", 2),
    # round 5 (VERDICT r4 missing #3): five to eight modes, through the kernels compiled for the plan
    ([1, 1, 0, 1, 1], "hydro", 10), ([1] * 8, "long", 6),
])
def test_numerical_families_vs_same_rule_oracle(gpu_cloudy, oracle, dist_types, kname, nq):
    cloudy = gpu_cloudy
    par, op, okf = numerical_case(cloudy, oracle, dist_types, kname, nq)
    mom = mixed_moments(dist_types, 1500, seed=100 + 7 * len(dist_types) + nq)
    got = run_numerical(cloudy, par, mom)
    want, scale, noise = oracle.rhs_coal_numerical_batch(op, okf, nq, mom, with_noise=True)
    if 3 in dist_types:
        # a very narrow Lognormal mode (the sigma = eps clamp of zero / negative variance inputs, ParticleDistributions.jl:
        # 483-505) puts its Gauss-Hermite nodes within sigma of each other: a kernel that vanishes on the diagonal
        # (hydrodynamic) then amplifies the last-place difference between the device's cbrt and the host's pow by
        # 1/sigma -- rounding noise on both sides, and so is the oracle's `scale`.  Such parcels are not compared.
        prm = oracle.update_dist_batch(op, mom)
        degenerate = np.zeros(mom.shape[1], dtype=bool)
        for i, t in enumerate(dist_types):
            if t == 3:
                degenerate |= ~(prm[3 * i + 2] > 1e-3)
        assert degenerate.mean() < 0.02
        got, want, scale, noise = got[:, ~degenerate], want[:, ~degenerate], scale[:, ~degenerate], noise[:, ~degenerate]
    worst = assert_same_rule(got, want, scale, noise, f"{dist_types} {kname} nq={nq}")
    print(f"{dist_types} {kname} nq={nq}: {worst:.2e}")


def test_numerical_plan_with_other_shape_clamp(gpu_cloudy, oracle):
    """param_range k <= 5 (test_ParticleDistributions_correctness.jl:128-132) and k <= 30: the plan stages a start-value
    table for its own k range; closures clamped at the bounds sit on the table's end points"""
    cloudy = gpu_cloudy
    for k_range in ((float(np.finfo(np.float64).eps), 5.0), (0.5, 30.0)):
        par, op, okf = numerical_case(cloudy, oracle, [1, 1], "hydro", 10, k_range=k_range)
        mom = bench.synth_moments(2, 3000, seed=77)
        got = run_numerical(cloudy, par, mom)
        want, scale, noise = oracle.rhs_coal_numerical_batch(op, okf, 10, mom, with_noise=True)
        prm = oracle.update_dist_batch(op, mom)
        assert prm[2].max() <= k_range[1] and (prm[2] == k_range[1]).any()
        assert_same_rule(got, want, scale, noise, f"k_range {k_range}")


@pytest.mark.parametrize("kname,kc", [("constant", [[1e-4]]), ("linear", [[0.0, 5.0], [5.0, 0.0]])])
def test_polynomial_kernels_match_the_analytic_path_on_the_device(gpu_cloudy, oracle, kname, kc):
    """one mode: weighting_fn == 1, the Numerical and Analytical closures coincide, and the rule integrates polynomial
    kernels exactly -> the quadrature plan and the tensor plan must agree to 1e-12 of the term scale (north_star)"""
    cloudy = gpu_cloudy
    par, op, okf = numerical_case(cloudy, oracle, [1], kname)
    mom = bench.synth_moments(1, 50_000, seed=9)
    got = run_numerical(cloudy, par, mom)
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor(np.array(kc)), (3,), (INF,), NORMS)
    par_a = cloudy.ODEParameters(par.pdists, cd, (3,), NORMS)
    rhs = cloudy.make_box_model_rhs(cloudy.AnalyticalCoalStyle())
    m, dm = dev(cloudy, mom), cloudy.DeviceArray.zeros(*mom.shape)
    rhs(dm, m, par_a, 0.0)
    ana = dm.to_numpy()
    _, scale = oracle.rhs_coal_numerical_batch(op, okf, 10, mom, with_scale=True)
    worst = assert_close_scaled(got, ana, scale, 1e-12, f"quadrature vs tensor plan, {kname}")
    print(f"{kname}: max |quadrature plan - tensor plan| / scale = {worst:.2e}")


def test_ahead_of_time_kernels_params_input_and_float_planes(gpu_cloudy, oracle):
    cloudy = gpu_cloudy
    L = cloudy.lib()
    par, op, okf = numerical_case(cloudy, oracle, [1, 1, 1], "hydro")
    mom = bench.synth_moments(3, 2000, seed=5)
    want, scale, noise = oracle.rhs_coal_numerical_batch(op, okf, 10, mom, with_noise=True)
    kf = par.kernel_func
    jit = cloudy.NumericalPlan([1, 1, 1], kf, NORMS, 10, specialize=1, quad_mode=cloudy.QUAD_FIXED)
    aot = cloudy.NumericalPlan([1, 1, 1], kf, NORMS, 10, specialize=-1, quad_mode=cloudy.QUAD_FIXED)
    assert jit.specialized and not aot.specialized
    m = dev(cloudy, mom)
    outs = []
    for plan in (jit, aot):
        dm = cloudy.DeviceArray.zeros(9, mom.shape[1])
        cloudy._lib.check(L.cloudy_coal_rhs(plan.handle, mom.shape[1], mom.shape[1], m.ptr, dm.ptr, None))
        outs.append(dm.to_numpy())
        assert_same_rule(outs[-1], want, scale, noise, "specialised" if plan is jit else "ahead of time")
    # get_coal_ints(::NumericalCoalStyle, pdists, kernel_func) on (n, theta, k) planes, normalised units out
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor(np.array([[0.0]])), (3, 3, 3), (INF,) * 3, NORMS)
    params = cloudy.DeviceArray.zeros(9, mom.shape[1])
    cloudy._lib.check(L.cloudy_update_dist_from_moments(cd.plan([1, 1, 1]).handle, mom.shape[1], mom.shape[1], m.ptr,
                                                        params.ptr, None))
    ci = cloudy.get_coal_ints(cloudy.NumericalCoalStyle(), ([1, 1, 1], params), kf, quad_mode=cloudy.QUAD_FIXED).to_numpy()
    norms9 = np.tile([NORMS[0], NORMS[0] * NORMS[1], NORMS[0] * NORMS[1] ** 2], 3)[:, None]
    assert_same_rule(ci * norms9, want, scale, noise, "get_coal_ints on parameter planes")
    # float planes: final rounding only
    got32 = run_numerical(cloudy, par, mom.astype(np.float32))
    want32, scale32 = oracle.rhs_coal_numerical_batch(op, okf, 10, mom.astype(np.float32).astype(np.float64), with_scale=True)
    ok = np.isfinite(want32) & (np.abs(want32) < 3e38)
    assert np.all(np.abs(got32[ok] - want32[ok]) <= 1e-6 * scale32[ok] + 1e-37)


def test_shape_parameter_outside_the_staged_rule_range_gives_nan_not_a_wrong_answer(gpu_cloudy):
    """cloudy_get_coal_ints takes (n, theta, k) as given.  The per-parcel Gauss-Laguerre rules are staged for
    0 < k <= max(k_range[1], 1) (include/cloudy_hip.h): a Gamma mode outside that range must yield NaN tendencies for
    its parcel -- extrapolated start values would converge to wrong nodes silently -- and leave the others untouched;
    with a wider k_range the same parcels are served."""
    cloudy = gpu_cloudy
    kf = cloudy.get_normalized_kernel_func(cloudy.HydrodynamicKernelFunction(1e2 * np.pi), NORMS)
    n = 256
    prm = np.zeros((6, n))
    rng = np.random.default_rng(3)
    prm[0], prm[1], prm[2] = 50.0, 0.3, rng.uniform(0.5, 9.5, n)
    prm[3], prm[4], prm[5] = 2.0, 40.0, rng.uniform(1.0, 9.5, n)
    bad1, bad2 = np.arange(n) % 7 == 3, np.arange(n) % 11 == 5
    prm[2, bad1] = rng.uniform(10.5, 50.0, bad1.sum())      # above k_range[1] = 10
    prm[5, bad2] = -1.0                                     # not a shape parameter at all
    pd = cloudy.DeviceArray.from_numpy(prm)
    got = cloudy.get_coal_ints(cloudy.NumericalCoalStyle(), ([1, 1], pd), kf, quad_mode=cloudy.QUAD_FIXED).to_numpy()
    bad = bad1 | bad2
    assert np.all(np.isnan(got[:, bad]).any(axis=0)) and np.all(np.isfinite(got[:, ~bad]))
    wide = cloudy.get_coal_ints(cloudy.NumericalCoalStyle(), ([1, 1], pd), kf, k_range=(2.3e-16, 50.0),
                                quad_mode=cloudy.QUAD_FIXED).to_numpy()
    assert np.all(np.isfinite(wide[:, ~bad2])) and np.all(np.isnan(wide[:, bad2]).any(axis=0))
    ok = ~bad
    assert np.allclose(wide[:, ok], got[:, ok], rtol=1e-9, atol=0.0)   # (two tables, same nodes after the Newton steps)


def test_numerical_plan_status_codes(gpu_cloudy):
    cloudy = gpu_cloudy
    L, E = cloudy.lib(), cloudy._lib
    plan = cloudy.NumericalPlan([1, 1], cloudy.LinearKernelFunction(5e-3), NORMS, 10, quad_mode=cloudy.QUAD_FIXED)
    u = cloudy.DeviceArray.zeros(6, 64)
    assert L.cloudy_ssprk33_steps(plan.handle, 64, 64, u.ptr, u.ptr, 1.0, 1, None) == 0   # served (empty boxes stay empty)
    assert L.cloudy_rainshaft_ssprk33_steps(plan.handle, 4, 16, 64, u.ptr, u.ptr, 100.0, 1.0, 1, None) == E.EINVAL  # no velocities
    assert L.cloudy_finite_2d_integrals(plan.handle, 64, 64, u.ptr, u.ptr, None) == E.EUNSUPPORTED
    assert b"NumericalCoalStyle" in L.cloudy_last_error()
    assert L.cloudy_compute_thresholds(plan.handle, 64, 64, u.ptr, u.ptr, None) == E.EUNSUPPORTED
    assert L.cloudy_coal_rhs(plan.handle, 0, 0, None, None, None) == 0          # empty batch
    assert L.cloudy_coal_rhs(plan.handle, 64, 32, u.ptr, u.ptr, None) == E.EINVAL  # ld < n
    with pytest.raises(cloudy.CloudyError) as ei:
        cloudy.NumericalPlan([1, 2], cloudy.LinearKernelFunction(5e-3), NORMS, 10)
    assert ei.value.code == E.EINVAL
    with pytest.raises(cloudy.CloudyError) as ei:
        cloudy.NumericalPlan([1], cloudy.LinearKernelFunction(5e-3), NORMS, 33)
    assert ei.value.code == E.EUNSUPPORTED


@pytest.mark.parametrize("dist_types,kname,specialize", [([1, 1, 1], "hydro", 0), ([1, 0], "long", 0), ([1, 1], "linear", -1)])
def test_fused_ssprk33_of_numerical_plans_vs_oracle_stepping(gpu_cloudy, oracle, dist_types, kname, specialize):
    """solve(prob, SSPRK33(), dt) of the Numerical drivers (n_particles_gamma.jl:39-40) fused on the device
    (quad_ssprk33_body) against SSPRK33 steps of the same-rule oracle RHS on the host; plan-time compiled kernel
    (through the host mirror) and, with specialize = -1, the ahead-of-time one (through the C ABI).  dt as in
    test_fused_ssprk33_batch_vs_oracle_stepping."""
    cloudy = gpu_cloudy
    par, op, okf = numerical_case(cloudy, oracle, dist_types, kname)
    n = 300
    mom = mixed_moments(dist_types, n, seed=17)
    dt, n_steps = 1e-3, 3
    want = _ssprk33_host(lambda u: oracle.rhs_coal_numerical_batch(op, okf, 10, u), mom, dt, n_steps)
    u = dev(cloudy, mom)
    if specialize == 0:
        cloudy.solve_ssprk33(par, u, dt, n_steps, coal_type=cloudy.NumericalCoalStyle())
    else:
        plan = cloudy.NumericalPlan(dist_types, par.kernel_func, NORMS, 10, specialize=specialize, quad_mode=cloudy.QUAD_FIXED)
        assert not plan.specialized
        cloudy._lib.check(cloudy.lib().cloudy_ssprk33_steps(plan.handle, n, n, u.ptr, u.ptr, dt, n_steps, None))
    got = u.to_numpy()
    with np.errstate(all="ignore"):
        ok = np.isfinite(want).all(axis=0) & (np.abs(want[:2]) <= 10 * np.abs(mom[:2]) + 1e-300).all(axis=0)
    assert ok.sum() > 0.85 * n
    ref = np.abs(mom) + np.abs(want)
    err = np.abs(got - want)[:, ok] / np.maximum(ref[:, ok], 1e-300)
    assert err.max() < 1e-9, err.max()
    # and the fused call equals staged stepping with the device RHS (same arithmetic, different call structure)
    staged = _ssprk33_host(lambda v: run_numerical(cloudy, par, v), mom, dt, n_steps)
    err2 = np.abs(got - staged)[:, ok] / np.maximum(ref[:, ok], 1e-300)
    assert err2.max() < 1e-12, err2.max()
    print(f"{dist_types} {kname}: fused SSPRK33 vs oracle stepping {err.max():.2e}, vs staged device RHS {err2.max():.2e}")


def test_cfg4q_full_size_properties(gpu_cloudy):
    """BASELINE configs[3] at its per-GPU size (1.25e7 parcels, 9 moments): size-independent properties -- mass
    conservation parcel by parcel, number never created, bilinearity in the number concentrations (all moments x s
    => tendencies x s^2) and linearity in the coalescence efficiency."""
    cloudy = gpu_cloudy
    n = 12_500_000
    mom = bench.synth_moments(3, n, seed=bench.SEED, degenerate_frac=0.0)
    kf = cloudy.HydrodynamicKernelFunction(1e2 * np.pi)
    mk = lambda e: cloudy.ODEParameters(tuple(cloudy.GammaPrimitiveParticleDistribution(1.0, 1.0, 1.0) for _ in range(3)),
                                        None, (3, 3, 3), NORMS, quad_mode=cloudy.QUAD_FIXED,
                                        kernel_func=cloudy.get_normalized_kernel_func(cloudy.HydrodynamicKernelFunction(e), NORMS))
    rhs = cloudy.make_box_model_rhs(cloudy.NumericalCoalStyle())
    m, dm = dev(cloudy, mom), cloudy.DeviceArray.zeros(9, n)
    rhs(dm, m, mk(1e2 * np.pi), 0.0)
    d = dm.to_numpy()
    assert np.all(np.isfinite(d))
    mass = d[1] + d[4] + d[7]
    mag = np.abs(d[1]) + np.abs(d[4]) + np.abs(d[7])
    assert np.max(np.abs(mass) / np.maximum(mag, 1e-300)) < 1e-9
    assert np.all(d[0] + d[3] + d[6] <= 0.0) and np.all(d[0] <= 0.0)
    assert np.all(d[8] >= 0.0)                                   # the largest mode's M2 only grows
    # bilinear in number: scale every moment by 2 (same theta, k) -> tendencies x 4
    m2 = dev(cloudy, 2.0 * mom)
    rhs(dm, m2, mk(1e2 * np.pi), 0.0)
    d2 = dm.to_numpy()
    assert np.max(np.abs(d2 - 4.0 * d) / np.maximum(np.abs(4.0 * d), 1e-300)) < 1e-12
    # linear in the efficiency
    rhs(dm, m, mk(3e2 * np.pi), 0.0)
    d3 = dm.to_numpy()
    assert np.max(np.abs(d3 - 3.0 * d) / np.maximum(np.abs(3.0 * d), 1e-300)) < 1e-12


def test_configs3_at_its_full_1e8_parcels_on_one_gpu_and_its_shards(gpu_cloudy):
    """BASELINE configs[3] -- 1e8 parcels, 3 modes, hydrodynamic kernel by the 10-point rule, 9
    moments -- is worded for 8 GPUs, 1.25e7 parcels each.  Its whole batch fits ONE MI355X (7.2 GB in, 7.2 GB out): the
    eight ranks' shards (seed + 1000 x rank, as bench.py draws them) are laid side by side, the batch is evaluated in one
    launch, and the shard rank g would own (shard_range) evaluated ALONE -- its own array, its own leading dimension, as on
    rank g's GPU -- equals the same parcels inside the full batch bit for bit: sharding the batch changes no result."""
    cloudy = gpu_cloudy
    L = cloudy.lib()
    world, n_shard = 8, 12_500_000
    n = world * n_shard
    kf = cloudy.get_normalized_kernel_func(cloudy.HydrodynamicKernelFunction(1e2 * np.pi), NORMS)
    plan = cloudy.numerical_plan([1, 1, 1], kf, NORMS, 10, quad_mode=cloudy.QUAD_FIXED)
    m, dm = cloudy.DeviceArray(9, n), cloudy.DeviceArray(9, n)
    shards = {}
    for g in range(world):
        lo, hi = cloudy.shard_range(n, g, world)
        assert hi - lo == n_shard
        blk = bench.synth_moments(3, n_shard, bench.SEED + 1000 * g)
        m.set_columns(lo, blk)
        if g in (0, 3, 7):
            shards[g] = blk
    cloudy._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None))
    cloudy._lib.check(L.cloudy_stream_synchronize(None))
    for g, blk in shards.items():
        lo, hi = cloudy.shard_range(n, g, world)
        inside = dm.columns_to_numpy(n_shard, lo)
        ms, ds = dev(cloudy, blk), cloudy.DeviceArray.zeros(9, n_shard)
        cloudy._lib.check(L.cloudy_coal_rhs(plan.handle, n_shard, n_shard, ms.ptr, ds.ptr, None))
        alone = ds.to_numpy()
        assert np.array_equal(inside, alone, equal_nan=True), g
        mass = alone[1] + alone[4] + alone[7]
        mag = np.abs(alone[1]) + np.abs(alone[4]) + np.abs(alone[7])
        okm = np.isfinite(mass)
        assert np.max(np.abs(mass[okm]) / np.maximum(mag[okm], 1e-300)) < 1e-9
        del ms, ds
    del m, dm


# ---- CONVERGED mode (quad_mode = CLOUDY_QUAD_CONVERGED, csrc/quad_conv.hpp) --------------------------------------------
TOL_CONVERGED = 1e-11   # HIP vs the same-rule oracle; the oracle vs adaptive quadrature is a CPU test (<= 1e-8 of scale)


def converged_case(cloudy, oracle, dist_types, kname, q=8):
    par, op, okf = numerical_case(cloudy, oracle, dist_types, kname, q)
    par.quad_mode = cloudy.QUAD_CONVERGED
    return par, op, okf


@pytest.mark.parametrize("dist_types,kname,q", [
    ([1], "hydro", 8), ([1], "long", 8), ([0], "linear", 8), ([1], "constant", 4),
    ([1, 1], "hydro", 8), ([1, 1], "long", 8), ([0, 1], "hydro", 8), ([1, 0], "linear", 6), ([0, 0], "long", 10),
    ([1, 1, 1], "hydro", 8), ([1, 1, 1], "long", 8), ([1, 1, 1], "linear", 8), ([1, 0, 1], "hydro", 12),
    ([1, 1, 1, 1], "hydro", 8), ([0, 1, 1, 1], "long", 6), ([1, 1], "hydro", 16), ([1, 1, 1], "constant", 2),
    # Lognormal modes: closed-form moments / partial moments, Phi for a Lognormal pair, the 1-D rule for a Gamma-Lognormal
    # pair of the hydrodynamic kernel, the (ln s, t) rule for T_m of a Lognormal mode
    ([3], "hydro", 8), ([3, 3], "linear", 8), ([3, 3], "hydro", 8), ([1, 3], "hydro", 8), ([3, 1], "hydro", 6),
    ([3, 1], "linear", 8), ([1, 3], "long", 8), ([3, 3, 3], "linear", 8), ([1, 3, 1], "hydro", 8), ([3, 0], "constant", 4),
    # round 5 (VERDICT r4 missing #3): the reference's NumericalCoalStyle is generic in the number of modes
    # (Coalescence.jl:470-489) -- five to eight modes through the kernels compiled for the plan
    # (eight modes: the golden case 8gamma_long below and tools/fuzz_parity.py --converged --big)
    ([1, 1, 1, 1, 1], "hydro", 8), ([1, 0, 1, 1, 1, 1], "long", 8), ([1, 1, 3, 1, 1, 1], "linear", 8),
])
def test_converged_mode_vs_same_rule_oracle(gpu_cloudy, oracle, dist_types, kname, q):
    """cloudy_coal_rhs of a CLOUDY_QUAD_CONVERGED plan (closed forms of the region integrals + one 1-D rule per mode for
    the weighting_fn split) against oracle/cloudy_oracle_quad.c's restatement of the same formulas, incl. the ~1 %
    degenerate parcels of the synthetic batch (clamped closures, empty modes)"""
    cloudy = gpu_cloudy
    par, op, okf = converged_case(cloudy, oracle, dist_types, kname, q)
    n = 300 if 3 in dist_types else 1500   # (the Lognormal T_m rule is ~5e4 nodes per parcel and mode on the CPU side too)
    mom = mixed_moments(dist_types, n, seed=300 + 7 * len(dist_types) + q)
    got = run_numerical(cloudy, par, mom)
    want, scale = oracle.rhs_coal_numerical_converged_batch(op, okf, q, mom, with_scale=True)
    if 3 in dist_types:   # narrow Lognormal modes (the sigma = eps clamp): see test_numerical_families_vs_same_rule_oracle
        prm = oracle.update_dist_batch(op, mom)
        degenerate = np.zeros(n, dtype=bool)
        for i, t in enumerate(dist_types):
            if t == 3:
                degenerate |= ~(prm[3 * i + 2] > 1e-3)
        assert degenerate.mean() < 0.03
        got, want, scale = got[:, ~degenerate], want[:, ~degenerate], scale[:, ~degenerate]
    worst = assert_same_rule(got, want, scale, np.zeros_like(scale), f"converged {dist_types} {kname}", tol=TOL_CONVERGED)
    print(f"converged {dist_types} {kname} q = {q}: max |hip - oracle| / scale = {worst:.2e}")
    ok = np.all(np.isfinite(got), axis=0)
    rows = np.cumsum([0] + [{0: 2, 1: 3, 3: 3}[t] for t in dist_types])[:-1] + 1
    mass, mag = got[rows].sum(axis=0), scale[rows].sum(axis=0)
    assert np.all(np.abs(mass[ok]) <= 1e-11 * mag[ok] + 1e-300)            # mass is conserved by the closed forms


@pytest.mark.parametrize("kname", ["hydro", "long", "linear", "constant"])
def test_converged_mode_extreme_parcels(gpu_cloudy, oracle, kname):
    """Hand-made parcels at the edges of the rule (a CPU sweep of round 4 against nested adaptive quadrature): modes
    4 ... 10 decades apart (where the reference's own Q - R + S loses the tendency to cancellation), identical and nearly
    identical modes, shapes at both clamps, a narrow Lognormal mode below a Gamma mode at ~2 e^mu -- against the same-rule
    oracle, with exact mass conservation."""
    cloudy = gpu_cloudy
    n0, m0 = NORMS
    rows = []
    def gamma_pair(n1, th1, k1, n2, th2, k2):
        return [n1, n1 * k1 * th1, n1 * k1 * (k1 + 1) * th1 ** 2, n2, n2 * k2 * th2, n2 * k2 * (k2 + 1) * th2 ** 2]
    for dec in (4, 6, 8, 10):
        rows.append(gamma_pair(10.0 ** (dec / 2), 10.0 ** (-dec / 2), 3.0, 10.0 ** (-dec / 2), 10.0 ** (dec / 2), 3.0))
    rows.append(gamma_pair(10.0, 1.0, 2.0, 10.0, 1.0, 2.0))              # identical modes
    rows.append(gamma_pair(10.0, 1.0, 2.0, 5.0, 1.0001, 2.0001))        # nearly identical
    rows.append(gamma_pair(50.0, 0.05, 9.99, 2.0, 3.0, 0.05))           # shapes next to the upper clamp / small
    rows.append(gamma_pair(50.0, 0.05, 1e-3, 2.0, 3.0, 9.0))            # a shape of 1e-3
    rows.append(gamma_pair(1.0, 1e-6, 2.0, 1.0, 1e6, 2.0))              # 12 decades, equal numbers
    mom = np.ascontiguousarray(np.array(rows).T * np.array([n0, n0 * m0, n0 * m0 ** 2] * 2)[:, None])
    par, op, okf = converged_case(cloudy, oracle, [1, 1], kname, 8)
    got = run_numerical(cloudy, par, mom)
    want, scale = oracle.rhs_coal_numerical_converged_batch(op, okf, 8, mom, with_scale=True)
    worst = assert_same_rule(got, want, scale, np.zeros_like(scale), f"extreme parcels {kname}", tol=TOL_CONVERGED)
    mass = np.abs(got[1] + got[4]) / (scale[1] + scale[4])
    print(f"extreme Gamma pairs, {kname}: max |hip - oracle| / scale = {worst:.2e}, mass residual {mass.max():.1e}")
    assert np.all(np.isfinite(got)) and mass.max() <= 1e-13
    # a narrow Lognormal mode below a Gamma mode sitting at ~2 e^mu (the inner rule of its T_m: panels of <= 3 sigma)
    lrows = []
    for sg in (0.05, 0.01, 0.004):
        mu = -1.0
        lrows.append([2.0, 2.0 * np.exp(mu + 0.5 * sg * sg), 2.0 * np.exp(2 * mu + 2 * sg * sg), 1.0, 1.0 * 2.0 * 0.9, 1.0 * 2.0 * 3.0 * 0.81])
    lmom = np.ascontiguousarray(np.array(lrows).T * np.array([n0, n0 * m0, n0 * m0 ** 2] * 2)[:, None])
    par, op, okf = converged_case(cloudy, oracle, [3, 1], kname, 8)
    got = run_numerical(cloudy, par, lmom)
    want, scale = oracle.rhs_coal_numerical_converged_batch(op, okf, 8, lmom, with_scale=True)
    worst = assert_same_rule(got, want, scale, np.zeros_like(scale), f"narrow Lognormal {kname}", tol=TOL_CONVERGED)
    print(f"narrow Lognormal below a Gamma mode, {kname}: max |hip - oracle| / scale = {worst:.2e}")


@pytest.mark.parametrize("kname", ["linear", "constant", "hydro", "long"])
def test_converged_mode_four_modes_ahead_of_time_kernel_repeats_itself(gpu_cloudy, oracle, kname):
    """Round 6: the ahead-of-time kernel of a FOUR-mode converged-mode plan (what runs when hiprtc is not available; 1 000+ spilled
    registers) built with 512 registers and the accumulation registers as spill space returned, for the linear kernel, different
    bits from run to run on identical input -- 1e-12 ... 1e-11 of scale on ~7 % of the parcels of a multi-scale batch
    (tools/jit_aot_diff.py; found by tools/fuzz_parity.py --converged).  The kernels are built for two waves per SIMD now, as the
    plan-time compiled ones; here: three calls and a fresh plan give the same bits, and they agree with the plan-time compiled
    kernel to 1e-13 of scale and with the same-rule oracle."""
    cloudy = gpu_cloudy
    par, op, okf = converged_case(cloudy, oracle, [1, 1, 1, 1], kname, 6)
    rng = np.random.Generator(np.random.Philox(key=4056))
    n = 1200
    rows = []
    for i in range(4):   # size classes 2.5 decades apart, shapes 1 ... 8, numbers falling with size
        k = rng.uniform(1.0, 8.0, n)
        th = 10.0 ** (-11.0 + 2.5 * i + rng.uniform(-0.8, 0.8, n)) / k
        nn = 10.0 ** (8.0 - 2.5 * i + rng.uniform(-1.0, 1.0, n))
        rows += [nn, nn * k * th, nn * k * (k + 1.0) * th * th]
    mom = np.ascontiguousarray(np.stack(rows))
    want, scale = oracle.rhs_coal_numerical_converged_batch(op, okf, 6, mom, with_scale=True)
    kf = par.kernel_func
    jit = cloudy.NumericalPlan([1, 1, 1, 1], kf, NORMS, 6, specialize=1, quad_mode=1)
    aot = cloudy.NumericalPlan([1, 1, 1, 1], kf, NORMS, 6, specialize=-1, quad_mode=1)
    assert jit.specialized and not aot.specialized
    L, m = cloudy.lib(), dev(cloudy, mom)

    def run_plan(plan):
        dm = cloudy.DeviceArray.zeros(*mom.shape)
        cloudy._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None))
        return dm.to_numpy()

    a = run_plan(jit)
    b = [run_plan(aot) for _ in range(3)]
    b.append(run_plan(cloudy.NumericalPlan([1, 1, 1, 1], kf, NORMS, 6, specialize=-1, quad_mode=1)))
    for other in b[1:]:
        assert np.array_equal(b[0], other, equal_nan=True), "the ahead-of-time kernel does not repeat itself"
    assert np.array_equal(a, run_plan(jit), equal_nan=True)

    def step_plan(plan):   # cloudy_ssprk33_steps of the plan: the fused integrator's ahead-of-time kernel (512 registers)
        u, o = dev(cloudy, mom), cloudy.DeviceArray.zeros(*mom.shape)
        cloudy._lib.check(L.cloudy_ssprk33_steps(plan.handle, n, n, u.ptr, o.ptr, 1e-3, 1, None))
        return o.to_numpy()

    s0 = step_plan(aot)
    assert np.array_equal(s0, step_plan(aot), equal_nan=True) and np.array_equal(s0, step_plan(aot), equal_nan=True)
    ok = np.isfinite(want) & (scale > 0)
    dj = float(np.max(np.abs(a - b[0])[ok] / scale[ok]))
    assert dj <= 1e-13, dj
    assert_same_rule(b[0], want, scale, np.zeros_like(scale), f"ahead of time, four modes, {kname}", tol=TOL_CONVERGED)
    print(f"four modes, {kname}: |plan-time compiled - ahead of time| / scale = {dj:.1e}")


def test_converged_mode_jit_aot_params_and_float_planes(gpu_cloudy, oracle):
    cloudy = gpu_cloudy
    L = cloudy.lib()
    par, op, okf = converged_case(cloudy, oracle, [1, 1, 1], "hydro")
    mom = bench.synth_moments(3, 2000, seed=5)
    want, scale = oracle.rhs_coal_numerical_converged_batch(op, okf, 8, mom, with_scale=True)
    kf = par.kernel_func
    jit = cloudy.NumericalPlan([1, 1, 1], kf, NORMS, 8, specialize=1, quad_mode=1)
    aot = cloudy.NumericalPlan([1, 1, 1], kf, NORMS, 8, specialize=-1, quad_mode=1)
    assert jit.specialized and not aot.specialized
    m = dev(cloudy, mom)
    for plan in (jit, aot):
        dm = cloudy.DeviceArray.zeros(9, mom.shape[1])
        cloudy._lib.check(L.cloudy_coal_rhs(plan.handle, mom.shape[1], mom.shape[1], m.ptr, dm.ptr, None))
        assert_same_rule(dm.to_numpy(), want, scale, np.zeros_like(scale), "specialised" if plan is jit else "ahead of time",
                         tol=TOL_CONVERGED)
    # get_coal_ints(::NumericalCoalStyle, pdists, kernel_func) on (n, theta, k) planes, normalised units out
    cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor(np.array([[0.0]])), (3, 3, 3), (INF,) * 3, NORMS)
    params = cloudy.DeviceArray.zeros(9, mom.shape[1])
    cloudy._lib.check(L.cloudy_update_dist_from_moments(cd.plan([1, 1, 1]).handle, mom.shape[1], mom.shape[1], m.ptr,
                                                        params.ptr, None))
    ci = cloudy.get_coal_ints(cloudy.NumericalCoalStyle(), ([1, 1, 1], params), kf, quad_order=8, quad_mode=1).to_numpy()
    norms9 = np.tile([NORMS[0], NORMS[0] * NORMS[1], NORMS[0] * NORMS[1] ** 2], 3)[:, None]
    assert_same_rule(ci * norms9, want, scale, np.zeros_like(scale), "get_coal_ints on parameter planes", tol=TOL_CONVERGED)
    got32 = run_numerical(cloudy, par, mom.astype(np.float32))
    want32, scale32 = oracle.rhs_coal_numerical_converged_batch(op, okf, 8, mom.astype(np.float32).astype(np.float64),
                                                                with_scale=True)
    ok = np.isfinite(want32) & (np.abs(want32) < 3e38)
    assert np.all(np.abs(got32[ok] - want32[ok]) <= 1e-6 * scale32[ok] + 1e-37)
    with pytest.raises(cloudy.CloudyError) as ei:
        cloudy.NumericalPlan([1, 1], kf, NORMS, 8, quad_mode=2)
    assert ei.value.code == cloudy._lib.EINVAL


def test_converged_mode_reaches_the_adaptive_golden_values_on_the_device(gpu_cloudy):
    """the device result itself (not only the oracle's) against nested adaptive quadrature of the reference integrals:
    every case of tests/golden/numerical_adaptive.json (Gamma, Exponential and Lognormal modes) through cloudy_get_coal_ints, <= 1e-8 of scale
    (north_star's tolerance for quadrature kernels against Coalescence.jl:503-708), with the 10-point rule beside it"""
    import json
    import os

    cloudy = gpu_cloudy
    with open(os.path.join(os.path.dirname(__file__), "golden", "numerical_adaptive.json")) as f:
        gold = json.load(f)
    mk = {0: cloudy.ConstantKernelFunction, 1: cloudy.LinearKernelFunction, 2: cloudy.HydrodynamicKernelFunction,
          3: cloudy.LongKernelFunction}
    n_cases, worst = 0, 0.0
    for c in gold["cases"]:
        types = [int(d[0]) for d in c["pdists"]]
        n_cases += 1
        kf = mk[c["kf"][0]](*c["kf"][1])
        prm = np.array([v for d in c["pdists"] for v in (d[1], d[2], d[3])])[:, None].repeat(4, axis=1)
        Q, R, S = (np.abs(np.array(c[x])) for x in "QRS")
        npm = [2 if t == 0 else 3 for t in types]
        scale = np.concatenate([[Q[m, :, k].sum() + R[m, :, k].sum() + S[m, 0, k] + (S[m, 1, k - 1] if k else 0.0)
                                 for m in range(npm[k])] for k in range(len(types))])
        res = {}
        # (plans of more than four modes are compiled per plan, 15-40 s a kernel: the 10-point rule is not run beside them)
        for mode, q in ((1, 8), (0, 10)) if len(types) <= 4 else ((1, 8),):
            ci = cloudy.get_coal_ints(cloudy.NumericalCoalStyle(), (types, dev(cloudy, prm)), kf, quad_order=q,
                                      quad_mode=mode).to_numpy()[:, 0]
            res[mode] = float(np.max(np.abs(ci - np.array(c["coal_ints"])) / scale))
        worst = max(worst, res[1])
        print(f"{c['name']:36s} converged {res[1]:.1e}   10-point rule {res.get(0, float('nan')):.1e}   (of scale, vs adaptive)")
    assert n_cases >= 32 and worst <= 1e-8, worst


def test_default_numerical_drop_in_is_the_converged_mode_and_meets_the_reference_tolerance(gpu_cloudy, oracle):
    """the line the shim promises stays unchanged -- make_box_model_rhs(NumericalCoalStyle()) with the
    reference's own ODE parameters, no quad_* field anywhere -- must answer within north_star's 1e-8 of the reference's nested
    quadgk.  Every golden case goes through the DEFAULT operator as moments (norms = (1, 1)); the plan behind it is the
    converged one; cloudy_plan_desc_init itself says CONVERGED; the 10-point rule is reachable only by asking for it."""
    import json
    import os

    cloudy, O = gpu_cloudy, oracle
    d = cloudy._lib.PlanDesc()
    cloudy.lib().cloudy_plan_desc_init(C.byref(d))
    assert d.quad_mode == cloudy.QUAD_CONVERGED == 1 and d.quad_order == 0
    with open(os.path.join(os.path.dirname(__file__), "golden", "numerical_adaptive.json")) as f:
        gold = json.load(f)
    mkk = {0: cloudy.ConstantKernelFunction, 1: cloudy.LinearKernelFunction, 2: cloudy.HydrodynamicKernelFunction,
           3: cloudy.LongKernelFunction}
    mkd = {0: lambda: cloudy.ExponentialPrimitiveParticleDistribution(1.0, 1.0),
           1: lambda: cloudy.GammaPrimitiveParticleDistribution(1.0, 1.0, 1.0),
           3: lambda: cloudy.LognormalPrimitiveParticleDistribution(1.0, 1.0, 1.0)}
    rhs = cloudy.make_box_model_rhs(cloudy.NumericalCoalStyle())
    worst, worst_fixed, n_cases = 0.0, 0.0, 0
    for c in gold["cases"]:
        types = [int(dd[0]) for dd in c["pdists"]]
        npm = tuple(2 if t == 0 else 3 for t in types)
        od = [O.make_dist(int(t), n, th, k) for t, n, th, k in c["pdists"]]
        mom = np.array([O.moment(dd, float(q)) for dd, np_ in zip(od, npm) for q in range(np_)])[:, None].repeat(4, axis=1)
        par = cloudy.ODEParameters(tuple(mkd[t]() for t in types), None, npm, (1.0, 1.0), kernel_func=mkk[c["kf"][0]](*c["kf"][1]))
        assert not hasattr(par, "quad_mode") and not hasattr(par, "quad_order")
        m, dm = dev(cloudy, mom), cloudy.DeviceArray.zeros(*mom.shape)
        rhs(dm, m, par, 0.0)
        got = dm.to_numpy()[:, 0]
        Q, R, S = (np.abs(np.array(c[x])) for x in "QRS")
        scale = np.concatenate([[Q[mo, :, k].sum() + R[mo, :, k].sum() + S[mo, 0, k] + (S[mo, 1, k - 1] if k else 0.0)
                                 for mo in range(npm[k])] for k in range(len(types))])
        # (a Lognormal closure inverted from its moments carries the conditioning of sigma^2 = ln(M0 M2 / M1^2))
        err = float(np.max(np.abs(got - np.array(c["coal_ints"])) / scale))
        worst = max(worst, err)
        par.quad_mode = cloudy.QUAD_FIXED          # the explicit opt-in (BASELINE configs[3]: "10-pt Gauss quadrature")
        rhs(dm, m, par, 0.0)
        worst_fixed = max(worst_fixed, float(np.max(np.abs(dm.to_numpy()[:, 0] - np.array(c["coal_ints"])) / scale)))
        n_cases += 1
    from cloudy_jl_amd.box_model import _numerical_plan_for
    par = cloudy.ODEParameters((mkd[1](), mkd[1]()), None, (3, 3), (1.0, 1.0), kernel_func=cloudy.LinearKernelFunction(5e-3))
    plan = _numerical_plan_for(par)
    assert plan.quad_mode == cloudy.QUAD_CONVERGED and plan.quad_order == 8
    print(f"default drop-in: worst {worst:.1e} of scale over {n_cases} golden cases (10-point opt-in: {worst_fixed:.1e})")
    assert n_cases >= 32 and worst <= 1e-8 and worst_fixed > 1e-4


def test_converged_mode_on_random_multi_scale_mixtures_on_the_device(gpu_cloudy, oracle):
    """the mixtures a fixed composite rule got wrong (tests/test_numerical_oracle.py, same draw): random 2-3 mode Gamma /
    Exponential mixtures with scales 3.5 decades and number densities 3 decades apart, shapes 1e-3 ... 10, the four kernel
    functions, one batch per (N, kernel family) through cloudy_get_coal_ints.  Device against the same-rule oracle at the
    same tolerance: <= 1e-11 of scale (the adaptive decisions may flip on a rounding difference; an accepted panel is good to
    ~1e-14, so that is all a flip can move); device against the oracle at tolerance 1e-13: <= 1e-8."""
    cloudy, O = gpu_cloudy, oracle
    rng = np.random.default_rng(11)
    mk = {0: cloudy.ConstantKernelFunction, 1: cloudy.LinearKernelFunction, 2: cloudy.HydrodynamicKernelFunction,
          3: cloudy.LongKernelFunction}
    worst_same, worst_ref, n_mix = 0.0, 0.0, 0
    for N in (2, 3):
        for kind in range(4):
            prm = {0: (0.7,), 1: (5e-3,), 2: (0.3,), 3: (float(10 ** rng.uniform(-1, 1)), 9.0, 5.0)}[kind]
            kf, okf = mk[kind](*prm), O.kernel_func(kind, *prm)
            n = 48
            planes = np.zeros((3 * N, n))
            for p in range(n):
                for i in range(N):
                    k = float(rng.choice([rng.uniform(0.05, 1.0), rng.uniform(1, 10), rng.uniform(1, 10), 10 ** rng.uniform(-3, -1)]))
                    planes[3 * i:3 * i + 3, p] = (10 ** rng.uniform(-1, 2), 10 ** rng.uniform(-2, 1.5), k)
            got = cloudy.get_coal_ints(cloudy.NumericalCoalStyle(), ([1] * N, dev(cloudy, planes)), kf, quad_order=8,
                                       quad_mode=cloudy.QUAD_CONVERGED).to_numpy()
            for p in range(n):
                pd = [O.make_dist(O.GAMMA, *planes[3 * i:3 * i + 3, p]) for i in range(N)]
                same, sc = O.get_coal_ints_numerical_converged(pd, okf, 8, O.CONV_TOL, with_scale=True)
                ref = O.get_coal_ints_numerical_converged(pd, okf, 8, 1e-13)
                sc = np.maximum(sc, 1e-300)
                worst_same = max(worst_same, float(np.max(np.abs(got[:, p] - same) / sc)))
                worst_ref = max(worst_ref, float(np.max(np.abs(got[:, p] - ref) / sc)))
                n_mix += 1
    print(f"{n_mix} random multi-scale mixtures on the device: max |hip - oracle(same tolerance)| / scale = {worst_same:.1e}, "
          f"max |hip - oracle(tol 1e-13)| / scale = {worst_ref:.1e}")
    assert worst_same <= TOL_CONVERGED and worst_ref <= 1e-8
    # the same with Lognormal modes in either position (2-D rule of a Lognormal T_m, the Gamma-Lognormal P(Y' < X') grid)
    worst_same, worst_ref, n_mix = 0.0, 0.0, 0
    for types in ([3, 1], [1, 3], [3, 3], [1, 3, 1]):
        for kind in range(4):
            prm = {0: (0.7,), 1: (5e-3,), 2: (0.3,), 3: (float(10 ** rng.uniform(-1, 1)), 9.0, 5.0)}[kind]
            kf, okf = mk[kind](*prm), O.kernel_func(kind, *prm)
            n, N = 12, len(types)
            planes = np.zeros((3 * N, n))
            for p in range(n):
                for i, t in enumerate(types):
                    if t == 3:
                        planes[3 * i:3 * i + 3, p] = (10 ** rng.uniform(-1, 2), rng.uniform(-3, 2),
                                                      rng.choice([rng.uniform(0.2, 1.0), rng.uniform(0.03, 0.2)]))
                    else:
                        planes[3 * i:3 * i + 3, p] = (10 ** rng.uniform(-1, 2), 10 ** rng.uniform(-2, 1.5), rng.uniform(0.3, 10.0))
            got = cloudy.get_coal_ints(cloudy.NumericalCoalStyle(), (types, dev(cloudy, planes)), kf, quad_order=8,
                                       quad_mode=cloudy.QUAD_CONVERGED).to_numpy()
            for p in range(n):
                pd = [O.make_dist(O.LOGNORMAL if t == 3 else O.GAMMA, *planes[3 * i:3 * i + 3, p]) for i, t in enumerate(types)]
                same, sc = O.get_coal_ints_numerical_converged(pd, okf, 8, O.CONV_TOL, with_scale=True)
                ref = O.get_coal_ints_numerical_converged(pd, okf, 8, 1e-13)
                sc = np.maximum(sc, 1e-300)
                worst_same = max(worst_same, float(np.max(np.abs(got[:, p] - same) / sc)))
                worst_ref = max(worst_ref, float(np.max(np.abs(got[:, p] - ref) / sc)))
                n_mix += 1
    print(f"{n_mix} random mixtures with Lognormal modes on the device: max |hip - oracle(same tolerance)| / scale = "
          f"{worst_same:.1e}, max |hip - oracle(tol 1e-13)| / scale = {worst_ref:.1e}")
    assert worst_same <= TOL_CONVERGED and worst_ref <= 1e-8


def test_converged_mode_fused_ssprk33_and_full_size_properties(gpu_cloudy, oracle):
    cloudy = gpu_cloudy
    par, op, okf = converged_case(cloudy, oracle, [1, 1, 1], "hydro")
    n = 200
    mom = mixed_moments([1, 1, 1], n, seed=17)
    dt, n_steps = 1e-3, 2
    want = _ssprk33_host(lambda u: oracle.rhs_coal_numerical_converged_batch(op, okf, 8, u), mom, dt, n_steps)
    u = dev(cloudy, mom)
    cloudy.solve_ssprk33(par, u, dt, n_steps, coal_type=cloudy.NumericalCoalStyle())
    got = u.to_numpy()
    with np.errstate(all="ignore"):
        ok = np.isfinite(want).all(axis=0) & (np.abs(want[:2]) <= 10 * np.abs(mom[:2]) + 1e-300).all(axis=0)
    assert ok.sum() > 0.85 * n
    ref = np.abs(mom) + np.abs(want)
    assert (np.abs(got - want)[:, ok] / np.maximum(ref[:, ok], 1e-300)).max() < 1e-9
    # BASELINE configs[3] at its per-GPU size: mass conservation, signs, bilinearity in number
    n = 12_500_000
    mom = bench.synth_moments(3, n, seed=bench.SEED, degenerate_frac=0.0)
    rhs = cloudy.make_box_model_rhs(cloudy.NumericalCoalStyle())
    m, dm = dev(cloudy, mom), cloudy.DeviceArray.zeros(9, n)
    rhs(dm, m, par, 0.0)
    d = dm.to_numpy()
    assert np.all(np.isfinite(d))
    mass, mag = d[1] + d[4] + d[7], np.abs(d[1]) + np.abs(d[4]) + np.abs(d[7])
    assert np.max(np.abs(mass) / np.maximum(mag, 1e-300)) < 1e-9
    assert np.all(d[0] + d[3] + d[6] <= 0.0) and np.all(d[0] <= 0.0) and np.all(d[8] >= 0.0)
    # a slice of the batch evaluated alone equals the same parcels inside the full batch, bit for bit: every lane walks its
    # own panels (no wave-level decisions), so sharding a batch over GPUs cannot change a result
    sl = slice(7_000_003, 7_050_003)
    ms, dms = dev(cloudy, np.ascontiguousarray(mom[:, sl])), cloudy.DeviceArray.zeros(9, 50_000)
    rhs(dms, ms, par, 0.0)
    assert np.array_equal(dms.to_numpy(), d[:, sl])
    rhs(dm, dev(cloudy, 2.0 * mom), par, 0.0)
    d2 = dm.to_numpy()
    assert np.max(np.abs(d2 - 4.0 * d) / np.maximum(np.abs(4.0 * d), 1e-300)) < 1e-12


def test_numerical_plans_of_five_and_more_modes_through_the_fused_integrators(gpu_cloudy, oracle):
    """VERDICT r4 missing #3: get_coal_ints(::NumericalCoalStyle) of the reference is generic in the number of modes
    (Coalescence.jl:470-489; n_particles_*.jl take any Ndist).  Plans of five to eight modes exist only as kernels compiled
    for the plan: the default (converged) operator fused into SSPRK33 and Tsit5 steps against staged stepping with the
    device RHS and the same-rule oracle, the slice-of-batch property, and the limit of the ABI (nine modes)."""
    cloudy = gpu_cloudy
    dist_types = [1, 1, 0, 1, 1]
    par, op, okf = converged_case(cloudy, oracle, dist_types, "hydro")
    plan = cloudy.NumericalPlan(dist_types, par.kernel_func, NORMS, 8)
    assert plan.specialized
    n = 192
    mom = mixed_moments(dist_types, n, seed=29)
    d = run_numerical(cloudy, par, mom)
    want, scale = oracle.rhs_coal_numerical_converged_batch(op, okf, 8, mom, with_scale=True)
    assert_same_rule(d, want, scale, np.zeros_like(scale), "five modes", tol=TOL_CONVERGED)
    sub = run_numerical(cloudy, par, np.ascontiguousarray(mom[:, 37:101]))
    assert np.array_equal(sub, d[:, 37:101])
    # the diagnostics of such a plan are compiled for it as well: closure inversion, and get_coal_ints on (n, theta, k) planes
    prm = cloudy.update_dist_from_moments(plan, dev(cloudy, mom))
    assert np.array_equal(prm.to_numpy(), oracle.update_dist_batch(op, mom))
    ci = cloudy.DeviceArray.zeros(*mom.shape)
    cloudy._lib.check(cloudy.lib().cloudy_get_coal_ints(plan.handle, n, n, prm.ptr, ci.ptr, None))
    nrm = np.concatenate([[NORMS[0] * NORMS[1] ** q for q in range(2 if t == 0 else 3)] for t in dist_types])[:, None]
    assert np.allclose(ci.to_numpy() * nrm, d, rtol=1e-13, atol=1e-14 * np.abs(scale).max()), "coal_ints on parameter planes"
    with np.errstate(all="ignore"):
        live = (np.abs(d) > 0) & (mom > 0)
        dt = 1e-3 * float(np.min(np.where(live, mom / np.abs(d), np.inf)))
    staged = _ssprk33_host(lambda v: run_numerical(cloudy, par, v), mom, dt, 2)
    u = dev(cloudy, mom)
    cloudy.solve_ssprk33(par, u, dt, 2, coal_type=cloudy.NumericalCoalStyle())
    ref = np.abs(mom) + np.abs(staged)
    assert (np.abs(u.to_numpy() - staged) / np.maximum(ref, 1e-300)).max() < 1e-12
    u = dev(cloudy, mom)
    out = cloudy.DeviceArray.zeros(*mom.shape)
    cloudy._lib.check(cloudy.lib().cloudy_tsit5_steps(plan.handle, n, n, u.ptr, out.ptr, dt, 1, None))
    t5 = out.to_numpy()
    assert np.isfinite(t5).all() and (np.abs(t5 - staged) / np.maximum(ref, 1e-300)).max() < 1e-3   # two small steps of either scheme
    with pytest.raises(cloudy.CloudyError) as ei:
        cloudy.NumericalPlan([1] * 9, par.kernel_func, NORMS, 8)
    assert ei.value.code == cloudy._lib.EUNSUPPORTED


@pytest.mark.parametrize("kname", ["hydro", "long"])
def test_cost_hints_of_the_converged_kernel_change_no_bit(gpu_cloudy, kname):
    """Round 5: the kernel compiled for a converged-mode plan ranks the parcels of a workgroup by the number of panel evaluations
    each took in the plan's PREVIOUS call (one byte per parcel of plan-owned scratch; quad_kernels.hpp, coal_rhs_quad_body) so
    that waves hold parcels of equal cost.  Which lane computes a parcel must not change a bit of its result: the first call of
    a plan (no hints: natural order), the second (hints of the same batch), a call on OTHER parcels in between (foreign hints),
    a slice of the batch evaluated alone with its own leading dimension, and a fresh plan all give identical tendencies."""
    cloudy = gpu_cloudy
    L = cloudy.lib()
    n = 70_001          # not a multiple of the workgroup size: the last workgroup ranks lanes without a parcel last
    kf = {"hydro": cloudy.HydrodynamicKernelFunction(1e2 * np.pi), "long": cloudy.LongKernelFunction(5.236e-10, 9.44e9, 5.78)}[kname]
    kfn = cloudy.get_normalized_kernel_func(kf, bench.NORMS)
    mom = bench.synth_moments(3, n, seed=41)
    other = bench.synth_moments(3, n, seed=42)
    m, mo, dm = dev(cloudy, mom), dev(cloudy, other), cloudy.DeviceArray.zeros(9, n)

    def rhs(plan, src, count=n, ld=n, out=dm):
        cloudy._lib.check(L.cloudy_coal_rhs(plan.handle, count, ld, src.ptr, out.ptr, None))
        cloudy._lib.check(L.cloudy_stream_synchronize(None))
        return out.to_numpy().copy()

    plan = cloudy.NumericalPlan([1, 1, 1], kfn, bench.NORMS, 8, specialize=1, quad_mode=cloudy.QUAD_CONVERGED)
    first = rhs(plan, m)
    second = rhs(plan, m)                    # ranked by the costs of the first call
    rhs(plan, mo)                            # leaves the hints of OTHER parcels behind
    third = rhs(plan, m)
    assert np.array_equal(first, second, equal_nan=True) and np.array_equal(first, third, equal_nan=True)
    fresh = cloudy.NumericalPlan([1, 1, 1], kfn, bench.NORMS, 8, specialize=1, quad_mode=cloudy.QUAD_CONVERGED)   # (a new plan: no hints yet)
    assert np.array_equal(first, rhs(fresh, m), equal_nan=True)
    lo, cnt = 12_345, 30_011
    ms, ds = dev(cloudy, np.ascontiguousarray(mom[:, lo:lo + cnt])), cloudy.DeviceArray.zeros(9, cnt)
    alone = rhs(plan, ms, cnt, cnt, ds)      # the hints at these indices belong to other parcels of the batch
    assert np.array_equal(alone, first[:, lo:lo + cnt], equal_nan=True)
    assert np.isfinite(first).all(axis=0).mean() > 0.95
    # a LARGER batch than the plan has seen: the hint bytes are re-allocated (zeroed) on the stream of the call
    nb = 90_003
    big = bench.synth_moments(3, nb, seed=43)
    mb, db = dev(cloudy, big), cloudy.DeviceArray.zeros(9, nb)
    got_big = rhs(plan, mb, nb, nb, db)
    fresh2 = cloudy.NumericalPlan([1, 1, 1], kfn, bench.NORMS, 8, specialize=1, quad_mode=cloudy.QUAD_CONVERGED)
    assert np.array_equal(got_big, rhs(fresh2, mb, nb, nb, db), equal_nan=True)
    assert np.array_equal(first, rhs(plan, m), equal_nan=True)   # and back to the first batch, under the big batch's hints


def test_reference_driver_n_particles_lognorm_end_to_end(gpu_cloudy, oracle):
    """examples/n_particles_lognorm.py = test/examples/Numerical/n_particles_lognorm.jl line for line through the Python mirror:
    two Lognormal modes, LinearKernelFunction(5.0), NumericalCoalStyle, SSPRK33 with dt = 1 s to 50 s.  The trajectory is
    unpinned by the reference (it asserts nothing); here the fused device integrator (converged mode) is compared with the same
    150 stages driven on the host with the same-rule oracle RHS, and total mass must be conserved."""
    import importlib.util

    O = oracle
    spec = importlib.util.spec_from_file_location("n_particles_lognorm", os.path.join(ROOT, "examples", "n_particles_lognorm.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    u0 = ex.dist_moments[:, None].copy()
    got = ex.solve(u0)[:, 0]
    op = O.make_params([O.LOGNORMAL] * 2, np.zeros((1, 1)), (INF, INF), norms=ex.norms)
    okf = O.get_normalized_kernel_func(O.kernel_func(O.KF_LINEAR, ex.coalescence_coeff), ex.norms)
    f = lambda u: O.rhs_coal_numerical_converged_batch(op, okf, 8, np.ascontiguousarray(u))   # noqa: E731
    u = u0.copy()
    for _ in range(int(round(ex.T_end / ex.dt))):   # OrdinaryDiffEq's SSPRK33
        up = u
        u = up + ex.dt * f(up)
        u = (3.0 * up + u + ex.dt * f(u)) / 4.0
        u = (up + 2.0 * u + 2.0 * ex.dt * f(u)) / 3.0
    want = u[:, 0]
    assert np.all(np.isfinite(got)) and np.allclose(got, want, rtol=1e-9, atol=0.0), (got, want)
    assert abs((got[1] + got[4]) - (u0[1, 0] + u0[4, 0])) <= 1e-12 * (u0[1, 0] + u0[4, 0])      # mass
    assert got[0] + got[3] < u0[0, 0] + u0[3, 0] and got[3] > u0[3, 0]                             # coalescence: fewer, larger
