"""Pins the CPU oracle (oracle/cloudy_oracle.c) to every known-answer value the reference's own
unit tests hold for the coalescence moment-RHS path (tests/golden/reference_kats.json; each entry
cites test file:line in CliMA/Cloudy.jl v0.6.0).  CPU only."""
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EPS = np.finfo(np.float64).eps
TYPES = {"exponential": 0, "gamma": 1, "monodisperse": 2, "lognormal": 3}


def mk(O, spec):
    t = TYPES[spec[0]]
    return O.make_dist(t, *spec[1:])


def _thr(x):
    return np.inf if x == "inf" else x


def test_moment_source_helper_kats(oracle, kats):
    for e in kats["moment_source_helper"]:
        got = oracle.moment_source_helper(mk(oracle, e["dist"]), e["p1"], e["p2"], e["x_threshold"],
                                          e["n_bins_per_log_unit"])
        assert got == pytest.approx(e["expected"], rel=e["rtol"], abs=e.get("atol", 0.0)), e["cite"]
        if "restated" in e:
            assert got == pytest.approx(e["restated"], rel=1e-13), e["cite"]


def test_simpson_kat(oracle, kats):
    # integrate_SimpsonEvenFast(90, dx, j -> x[j]^2) on range(1, 10, 91) ~ 333 (atol 1e-6)
    import ctypes as C

    e = kats["simpson"]
    n = e["n_bins"]
    x = np.linspace(e["x0"], e["x1"], n + 1)
    dx = x[1] - x[0]
    CB = C.CFUNCTYPE(C.c_double, C.c_int, C.c_void_p)
    cb = CB(lambda j, ctx: float(x[j - 1] ** 2))
    f = oracle.lib().co_integrate_simpson_even_fast
    f.restype = C.c_double
    f.argtypes = [C.c_int, C.c_double, CB, C.c_void_p]
    assert abs(f(n, dx, cb, None) - e["expected"]) < e["atol"]
    assert math.isnan(f(2, dx, cb, None))  # reference: error("n_bins must be at least 3")


def test_update_dist_from_moments_kats(oracle, kats):
    for e in kats["update_dist_from_moments"]:
        t = TYPES[e["type"]]
        d = oracle.update_dist_from_moments(t, e["moments"], e.get("k_range"))
        if "expected_params" in e:
            ep = e["expected_params"]
            assert d.n == ep["n"] and d.theta == ep["theta"], e["cite"]
            if "k" in ep:
                assert d.k == ep["k"], e["cite"]
        if "expected_params_approx" in e:
            ep = e["expected_params_approx"]
            assert d.n == pytest.approx(ep["n"], rel=e["rtol"])
            assert d.theta == pytest.approx(ep["mu"], rel=e["rtol"])
            assert d.k == pytest.approx(ep["sigma"], rel=e["rtol"])
        if "expected_moments" in e:
            for q, em in enumerate(e["expected_moments"]):
                assert oracle.moment(d, float(q)) == pytest.approx(em, rel=e["rtol"], abs=1e-300), e["cite"]
        if "normed_density_at_0" in e:
            assert oracle.normed_density(d, 0.0) == e["normed_density_at_0"]
        if "normed_density_at_1" in e:
            assert oracle.normed_density(d, 1.0) == pytest.approx(e["normed_density_at_1"], rel=e["rtol"])
    for e in kats["wrong_arity"]:
        with pytest.raises(TypeError):
            oracle.update_dist_from_moments(TYPES[e["type"]], e["moments"])


def test_degenerate_moment_fallbacks(oracle):
    # ParticleDistributions.jl:461-475: M0 <= eps or M1 <= eps -> (0, 1, 1); k clamp [eps, 10]
    d = oracle.update_dist_from_moments(1, [EPS, 1.0, 1.0])
    assert (d.n, d.theta, d.k) == (0.0, 1.0, 1.0)
    d = oracle.update_dist_from_moments(1, [1.0, 1.0, 1.0])  # zero variance -> k = +Inf -> 10
    assert d.k == 10.0 and d.theta == 0.1
    d = oracle.update_dist_from_moments(1, [1.0, 1.0, 0.5])  # negative variance -> k < 0 -> eps
    assert d.k == EPS
    d = oracle.update_dist_from_moments(1, [1.0, 2.0, 1e9])
    assert d.k == pytest.approx(4.0 / (1e9 - 4.0))


def test_moment_kats(oracle, kats):
    for e in kats["moments"]:
        d = mk(oracle, e["dist"])
        for q, ex in zip(e["q"], e["expected"]):
            got = oracle.moment(d, q)
            if e.get("exact"):
                assert got == ex, (e["cite"], q)
            else:
                assert got == pytest.approx(ex, rel=e["rtol"]), (e["cite"], q)
    assert oracle.moment(mk(oracle, ["gamma", 1.0, 1.0, 2.0]), 2 / 3) == pytest.approx(
        math.gamma(2 + 2 / 3) / math.gamma(2), rel=1e-15)
    assert list(oracle.get_moments(mk(oracle, ["gamma", 1.0, 1.0, 2.0]))) == [1.0, 2.0, 6.0]
    assert list(oracle.get_moments(mk(oracle, ["exponential", 1.0, 2.0]))) == [1.0, 2.0]


def test_constructor_errors(oracle):
    # ParticleDistributions.jl:72-75,101-104,126-129,153-156
    for spec in (["monodisperse", -1.0, 2.0], ["monodisperse", 1.0, -2.0], ["exponential", -1.0, 2.0],
                 ["exponential", 1.0, -2.0], ["gamma", -1.0, 2.0, 3.0], ["gamma", 1.0, -2.0, 3.0],
                 ["gamma", 1.0, 2.0, -3.0], ["lognormal", -1.0, 2.0, 3.0], ["lognormal", 1.0, 2.0, -3.0]):
        with pytest.raises(ValueError):
            mk(oracle, spec)
    assert [oracle.nparams(t) for t in range(4)] == [2, 3, 2, 3]


def test_density_kats(oracle, kats):
    for e in kats["density"]:
        d = mk(oracle, e["dist"])
        for x, ex in zip(e["x"], e["expected"]):
            assert oracle.density(d, x) == pytest.approx(ex, rel=e["rtol"], abs=0.0), e["cite"]
    with pytest.raises(ValueError):
        oracle.density(mk(oracle, ["gamma", 1.0, 1.0, 2.0]), -3.1)
    assert math.isnan(oracle.density(mk(oracle, ["lognormal", 1.0, 1.0, 2.0]), 0.0))


def test_compute_threshold_kats(oracle, kats):
    for e in kats["compute_thresholds"]:
        pd = [mk(oracle, s) for s in e["pdists"]]
        if "expect_gt" in e:
            for d, lo in zip(pd, e["expect_gt"]):
                assert oracle.compute_threshold(d, e["percentile"]) > lo
        elif "expected" in e:
            for d, ex in zip(pd, e["expected"]):
                assert abs(oracle.compute_threshold(d, e["percentile"]) - ex) < e["atol"]
        elif "thresholds_default" in e:
            t = oracle.compute_thresholds(pd)
            assert t[0] == pytest.approx(e["thresholds_default"][0], rel=e["rtol"])
            assert t[0] == pytest.approx(e["restated"], rel=1e-13)
            assert t[1] > 1e6
        else:
            t = oracle.compute_thresholds(pd, e["percentiles"])
            assert t[0] == pytest.approx(e["thresholds_first"], rel=e["rtol"])
            assert t[0] == pytest.approx(e["restated"], rel=1e-13)
            assert np.isinf(t[1])


def test_helper_function_kats(oracle, kats):
    e = kats["helper_functions"]
    npm = e["NProgMoms"]
    for i, m, ex in e["moment_ind"]:
        assert oracle.get_dist_moment_ind(npm, i, m) == ex
    for i, m in e["moment_ind_throws"]:
        with pytest.raises(ValueError):
            oracle.get_dist_moment_ind(npm, i, m)
    for i, a, b in e["ind_range"]:
        assert oracle.get_dist_moments_ind_range(npm, i) == range(a, b + 1)
    for i in e["ind_range_throws"]:
        with pytest.raises(IndexError):
            oracle.get_dist_moments_ind_range(npm, i)
    nf = oracle.get_moments_normalizing_factors(npm, e["norms"])
    assert np.allclose(nf, e["norm_factors"], rtol=0, atol=e["atol"])
    with pytest.raises(ValueError):
        oracle.get_moments_normalizing_factors(npm, (0.0, 1.0))


def test_kernel_tensor_kats(oracle, kats):
    e = kats["normalized_kernel_tensor"]
    got = oracle.get_normalized_kernel_tensor(e["c"], e["norms"])
    assert np.allclose(got, e["expected"], rtol=0, atol=e["atol"])
    for c in kats["check_symmetry"]["ok"]:
        oracle.check_symmetry(c)
    for c in kats["check_symmetry"]["throws"]:
        with pytest.raises(ValueError):
            oracle.check_symmetry(c)
    with pytest.raises(ValueError):
        oracle.coalescence_data([[1.0, -0.2], [0.2, 2.0]], (3,), (np.inf,))


def test_kernel_function_kats(oracle, kats):
    L = oracle.lib()
    fn = {"constant": L.co_constant_kernel, "linear": L.co_linear_kernel, "hydrodynamic": L.co_hydrodynamic_kernel,
          "long": L.co_long_kernel}
    for e in kats["kernel_functions"]:
        f = fn[e["kind"]]
        got = f(*e["params"], e["x"], e["y"])
        if e["kind"] == "hydrodynamic":
            x, y = e["x"], e["y"]
            r1, r2 = (3 / 4 / math.pi * x) ** (1 / 3), (3 / 4 / math.pi * y) ** (1 / 3)
            ex = e["params"][0] * (r1 + r2) ** 2 * abs(math.pi * r1**2 - math.pi * r2**2)
            assert got == pytest.approx(ex, rel=4 * EPS)
        else:
            assert got == e["expected"], e["cite"]
        if e.get("symmetric"):
            assert f(*e["params"], e["y"], e["x"]) == got


def test_sm1916_constant_kernel(oracle, kats):
    # test_Sources_correctness.jl:41-85: 5 Euler steps, constant kernel, Exponential, thr = Inf
    e = kats["sm1916"]
    cd = oracle.coalescence_data(e["kernel_c"], (2,), (np.inf,))
    mom = list(e["init_moments"])
    for _ in range(e["n_steps"]):
        d = oracle.update_dist_from_moments(0, mom)
        dm = oracle.get_coal_ints([d], cd)
        mom = [e["dt"] * dm[i] + mom[i] for i in range(2)]
    t_end = e["dt"] * e["n_steps"]
    for i in range(e["n_steps"] + 1):
        t = e["dt"] * i  # the reference loop compares the end state against every t (rtol 1e-3 absorbs it)
        ana = 1.0 / (1.0 / e["a"] + e["b"] / 2.0 * t)
        assert mom[0] == pytest.approx(ana, rel=e["rtol"])
        assert mom[1] == pytest.approx(e["expected_M1"], rel=e["rtol"])
    assert mom[0] == pytest.approx(1.0 / (1.0 / e["a"] + e["b"] / 2.0 * t_end), rel=1e-7)
    assert mom[1] == 2.0  # mass exactly conserved


def test_gamma_exp_coal_ints_vs_in_test_triple_loop(oracle, kats):
    """test_Sources_correctness.jl:87-169: get_coal_ints vs the test's own independent loop, rtol 10*eps."""
    e = kats["gamma_exp_coal_ints"]
    dist = [mk(oracle, s) for s in e["pdists"]]
    c = np.array(e["kernel_c"])
    NProgMoms = e["NProgMoms"]
    thr = [_thr(t) for t in e["thresholds"]]
    order = e["order"]
    cd = oracle.coalescence_data(c, NProgMoms, thr)
    coal_ints = oracle.get_coal_ints(dist, cd)

    n_mom = max(NProgMoms) + order
    mom = np.zeros((2, n_mom))
    for i in range(2):
        for j in range(n_mom):
            mom[i, j] = oracle.moment(dist[i], float(j))
    int_w = np.zeros((n_mom, n_mom))
    mtm = np.zeros((n_mom, n_mom))
    for i in range(n_mom):
        for j in range(i, n_mom):
            mtm[i, j] = mom[0, i] * mom[0, j]
            tmp = 0.0 if mtm[i, j] < EPS else oracle.moment_source_helper(dist[0], float(i), float(j), thr[0])
            int_w[i, j] = min(mtm[i, j], tmp)
            mtm[j, i] = mtm[i, j]
            int_w[j, i] = int_w[i, j]
    coal_int = np.zeros(5)
    for i in (1, 2):
        j = 2 if i == 1 else 1
        for k in range(NProgMoms[i - 1]):
            temp = 0.0
            for a in range(order + 1):
                for b in range(order + 1):
                    coef = c[a, b]
                    temp -= coef * mom[i - 1, a + k] * mom[i - 1, b]
                    temp -= coef * mom[i - 1, a + k] * mom[j - 1, b]
                    for cc in range(k + 1):
                        cb = coef * math.comb(k, cc)
                        if i == 1:
                            temp += 0.5 * cb * int_w[a + cc, b + k - cc]
                        else:
                            temp += 0.5 * cb * (mtm[a + cc, b + k - cc] - int_w[a + cc, b + k - cc])
                            temp += 0.5 * cb * mom[i - 1, a + cc] * mom[i - 1, b + k - cc]
                            temp += cb * mom[j - 1, a + cc] * mom[i - 1, b + k - cc]
            coal_int[oracle.get_dist_moment_ind(NProgMoms, i, k + 1) - 1] = temp
    assert np.allclose(coal_ints, coal_int, rtol=e["rtol_vs_in_test_triple_loop"], atol=0)
    assert np.allclose(coal_ints, e["restated"], rtol=1e-12, atol=0)
    # mass conservation between the two modes
    assert coal_ints[1] + coal_ints[4] == pytest.approx(0.0, abs=1e-15)


def test_weighting_fn_kats(oracle, kats):
    for e in kats["weighting_fn"]:
        pd = [mk(oracle, s) for s in e["pdists"]]
        assert oracle.weighting_fn(e["x"], e["k"], pd) == e["expected"], e["cite"]
    with pytest.raises(AssertionError):
        oracle.weighting_fn(10.0, 2, [mk(oracle, ["gamma", 10.0, 10.0, 3.0])])


def test_sedimentation_kat(oracle, kats):
    e = kats["sedimentation"]
    got = oracle.get_sedimentation_flux([mk(oracle, s) for s in e["pdists"]], e["vel"])
    ex = [-1.0 + math.gamma(1.0 + 1.0 / 6), -1.0 + math.gamma(2.0 + 1.0 / 6)]
    assert np.allclose(got, ex, rtol=e["rtol"], atol=0)
    assert np.allclose(got, e["expected"], rtol=e["rtol"], atol=0)


def test_special_functions_vs_scipy(oracle):
    """SpecialFunctions.jl is un-vendored: the restated P(a,x) / its inverse are checked against an
    independent implementation (scipy/cephes) over the range the path visits."""
    sp = pytest.importorskip("scipy.special")
    rng = np.random.default_rng(20260723)
    a = rng.uniform(1e-3, 20.0, 20000)
    x = 10.0 ** rng.uniform(-6, 2.7, 20000)
    mine = np.array([oracle.gamma_inc_p(ai, xi) for ai, xi in zip(a, x)])
    ref = sp.gammainc(a, x)
    sel = ref > 1e-100
    assert np.max(np.abs(mine[sel] - ref[sel]) / ref[sel]) < 5e-14
    p = rng.uniform(1e-6, 1 - 1e-6, 4000)
    a2 = rng.uniform(0.02, 12.0, 4000)
    inv = np.array([oracle.gamma_inc_inv(ai, pi) for ai, pi in zip(a2, p)])
    refi = sp.gammaincinv(a2, p)
    sel = refi > 1e-100
    assert np.max(np.abs(inv[sel] - refi[sel]) / refi[sel]) < 1e-12
    g = np.array([oracle.gamma(v) for v in a])
    assert np.max(np.abs(g - sp.gamma(a)) / sp.gamma(a)) < 1e-14


def test_closed_forms_all_inf(oracle):
    """single mode, thresholds Inf.  constant: dM0=-A/2 M0^2, dM1=0, dM2=A M1^2;
    Golovin: dM0=-b M0 M1, dM1=0, dM2=2b M1 M2 (in normalised units, norms=(1,1))."""
    rng = np.random.default_rng(1)
    for _ in range(200):
        n, k, th = 10 ** rng.uniform(-2, 3), rng.uniform(0.5, 8), 10 ** rng.uniform(-2, 1)
        d = oracle.make_dist(1, n, th, k)
        M0, M1, M2 = oracle.get_moments(d)
        A = 0.7
        out = oracle.get_coal_ints([d], oracle.coalescence_data([[A]], (3,), (np.inf,)))
        assert np.allclose(out, [-A / 2 * M0 * M0, 0.0, A * M1 * M1], rtol=1e-12, atol=1e-14 * A * M0 * M1)
        b = 0.3
        out = oracle.get_coal_ints([d], oracle.coalescence_data([[0, b], [b, 0]], (3,), (np.inf,)))
        assert np.allclose(out, [-b * M0 * M1, 0.0, 2 * b * M1 * M2], rtol=1e-12, atol=1e-13 * b * M0 * M2)


def test_rhs_coal_batch_matches_single(oracle):
    inf = np.inf
    p = oracle.make_params([1, 1], np.array([[0, 5.0], [5.0, 0]]), (5e-10, inf), norms=(1e6, 1e-9))
    rng = np.random.default_rng(3)
    n = 64
    mom = np.zeros((6, n))
    for mode, (nlo, nhi, xlo, xhi) in enumerate([(1e6, 1e9, 1e-11, 1e-9), (1.0, 1e5, 1e-9, 1e-7)]):
        nn = 10 ** rng.uniform(np.log10(nlo), np.log10(nhi), n)
        k = rng.uniform(0.5, 8, n)
        xb = 10 ** rng.uniform(np.log10(xlo), np.log10(xhi), n)
        th = xb / k
        mom[3 * mode + 0] = nn
        mom[3 * mode + 1] = nn * k * th
        mom[3 * mode + 2] = nn * k * (k + 1) * th * th
    d, scale = oracle.rhs_coal_batch(p, mom, with_scale=True, n_threads=2)
    for i in (0, 7, 63):
        assert np.array_equal(d[:, i], oracle.rhs_coal(p, mom[:, i]))
    # mass conservation across the two modes (physical units).  The net mass tendency of a mode is a
    # small difference of large Q/R/S terms, so the residual is measured against `scale`,
    # the sum of |terms| the oracle reports per output.
    assert np.all(np.abs(d[1] + d[4]) <= 1e-12 * scale[1])


def test_condensation_kats(oracle, kats):
    """test_Sources_correctness.jl:274-308: get_cond_evap closed forms (moments via the oracle's own moment())."""
    C = (4 * math.pi / 3) ** (2 / 3) / 1000.0 ** (1 / 3)
    for e in kats["condensation"]:
        pd = [mk(oracle, s) for s in e["pdists"]]
        got = oracle.get_cond_evap(pd, e["s"], e["xi"])
        want = []
        for d in pd:
            want.append(0.0)
            for j in range(1, oracle.nparams(d.type)):
                want.append(3 * j * e["xi"] * e["s"] * oracle.moment(d, j - 2 / 3) * C)
        assert np.allclose(got, want, rtol=1.5e-8, atol=0)
    # independent value of the fractional moment: Exp(1,1): M_{1/3} = Gamma(4/3)
    got = oracle.get_cond_evap([mk(oracle, ["exponential", 1.0, 1.0])], 0.01, 1e-6)
    assert got[1] == pytest.approx(3 * 1e-6 * 0.01 * math.gamma(4 / 3) * C, rel=1e-14)


def test_get_standard_N_q_kats(oracle, kats):
    """test_ParticleDistributions_correctness.jl:234-247: liquid + rain number / mass add up to the totals and the
    cloud part grows with the cutoff."""
    e = kats["get_standard_N_q"]
    pd = [mk(oracle, s) for s in e["pdists"]]
    q1, q2 = (oracle.get_standard_N_q(pd, c) for c in e["cutoffs"])
    for q in (q1, q2):
        assert q[0] + q[1] == pytest.approx(e["N_total"], rel=e["rtol"])
        assert q[2] + q[3] == pytest.approx(e["M_total"], rel=e["rtol"])
    assert q1[0] > q2[0] and q1[2] > q2[2]
    # closed form of one term: Exp(10, 1): N below 1.0 = 10 (1 - e^-1)
    assert oracle.partial_moment(pd[0], 0.0, 1.0) == pytest.approx(10.0 * (1 - math.exp(-1.0)), rel=1e-14)


def test_lognormal_partial_moment_and_threshold_rule_vs_adaptive_quadrature(oracle):
    """The two Lognormal integrals the reference evaluates with quadgk (ParticleDistributions.jl:255-269, :614-625):
    the oracle's closed form / fixed 48-node rule -- the same the HIP kernels evaluate -- against adaptive quadrature of
    the reference integrands (tests/golden/lognormal_adaptive.json, generated by oracle/lognormal_adaptive.py with
    scipy in the build container).  Errors are stated relative to the unthresholded value (M_q, M_p1 M_p2): that is what
    enters F = min(M_p1 M_p2, .) and the cloud / rain split."""
    import json

    with open(os.path.join(ROOT, "tests", "golden", "lognormal_adaptive.json")) as f:
        gold = json.load(f)
    worst_pm = worst_msh = 0.0
    for c in gold["cases"]:
        d = oracle.make_dist(oracle.LOGNORMAL, c["n"], c["mu"], c["sigma"])
        for q, want in c["partial_moment"].items():
            got = oracle.partial_moment(d, float(q), c["xt"])
            worst_pm = max(worst_pm, abs(got - want) / oracle.moment(d, float(q)))
        for key, want in c["moment_source_helper"].items():
            p1, p2 = (float(v) for v in key.split(","))
            got = oracle.moment_source_helper(d, p1, p2, c["xt"])
            worst_msh = max(worst_msh, abs(got - want) / (oracle.moment(d, p1) * oracle.moment(d, p2)))
    print(f"lognormal partial_moment closed form vs adaptive: {worst_pm:.2e} of M_q; "
          f"moment_source_helper 48-node rule vs nested adaptive: {worst_msh:.2e} of M_p1 M_p2")
    assert worst_pm <= 1e-12 and worst_msh <= 1e-11


def test_get_standard_N_q_with_a_lognormal_mode(oracle):
    """the reference's own performance-test configuration, get_standard_N_q((mono, lognormal, gamma))
    (performance_tests.jl:94-99): liquid + rain recover the totals, the split moves with the cutoff"""
    pd = [oracle.make_dist(oracle.MONODISPERSE, 1.0, 0.5), oracle.make_dist(oracle.LOGNORMAL, 1.0, 1.0, 2.0),
          oracle.make_dist(oracle.GAMMA, 1.0, 1.0, 2.0)]
    a, b = oracle.get_standard_N_q(pd, 1.0), oracle.get_standard_N_q(pd, 0.2)
    n_tot = sum(oracle.moment(d, 0.0) for d in pd)
    m_tot = sum(oracle.moment(d, 1.0) for d in pd)
    for r in (a, b):
        assert r[0] + r[1] == pytest.approx(n_tot, rel=1e-14) and r[2] + r[3] == pytest.approx(m_tot, rel=1e-14)
    assert a[0] > b[0] and a[2] > b[2] and all(np.isfinite(a)) and all(np.isfinite(b))


def test_bench_lognorm_example_batch_is_the_reference_configuration(oracle):
    """bench.lognorm_example_moments: boxes of test/examples/Numerical/n_particles_lognorm.jl:17-24 -- two Lognormal modes,
    sigma = ln 2 exactly, numbers / mass scales within a factor 2 of (1e7, 1e5) per m^3 and (1e-10, 1e-9) kg -- recovered by
    update_dist_from_moments (ParticleDistributions.jl:483-505) in normalised units."""
    import bench

    O = oracle
    mom = bench.lognorm_example_moments(2000, seed=5)
    p = O.make_params([O.LOGNORMAL] * 2, np.zeros((1, 1)), (np.inf,) * 2, norms=bench.NORMS)
    ntk = O.update_dist_batch(p, mom)
    assert np.allclose(ntk[2], np.log(2.0), rtol=1e-9) and np.allclose(ntk[5], np.log(2.0), rtol=1e-9)
    assert (ntk[0] >= 5.0 - 1e-9).all() and (ntk[0] <= 20.0 + 1e-9).all() and (ntk[3] >= 0.05 - 1e-12).all() and (ntk[3] <= 0.2 + 1e-12).all()
    assert (np.abs(ntk[1] - np.log(0.1)) <= np.log(2.0) + 1e-9).all() and (np.abs(ntk[4]) <= np.log(2.0) + 1e-9).all()


def test_check_moment_consistency_kats_and_closure_stats(oracle, kats, cloudy):
    """check_moment_consistency (ParticleDistributions.jl:437-449): the reference's seven known answers
    (test_ParticleDistributions_correctness.jl:221-234) through the oracle and through the host-side Python mirror; and
    co_closure_stats -- the batch form of the reference's silent clamps and of this check, the checker of
    cloudy_closure_stats -- against a plain numpy restatement on the bench batch (whose ~1 % degenerate parcels make every
    counter non-zero)."""
    import bench

    O = oracle
    for c in kats["check_moment_consistency"]:
        assert (O.check_moment_consistency(c["m"]) != 0) == c["throws"], c
        if c["throws"]:
            with pytest.raises(ValueError):
                cloudy.check_moment_consistency(c["m"])
        else:
            assert cloudy.check_moment_consistency(c["m"]) is None
    assert O.check_moment_consistency([0.1, -1.0]) == 1 and O.check_moment_consistency([1.0, 3.0, 2.0]) == 2
    n = 50_000
    mom = bench.synth_moments(2, n, seed=5)
    p = bench.oracle_params("cfg3b")
    got = O.closure_stats(p, mom)
    par = O.update_dist_batch(p, mom)
    eps = np.finfo(np.float64).eps
    norms = np.array([bench.NORMS[0] * bench.NORMS[1] ** q for q in range(3)])
    with np.errstate(all="ignore"):
        for m in range(2):
            mz = mom[3 * m:3 * m + 3] / norms[:, None]
            fb = ~((mz[0] > eps) & (mz[1] > eps))
            assert np.array_equal(fb, (par[3 * m] == 0.0) & (par[3 * m + 1] == 1.0) & (par[3 * m + 2] == 1.0))
            k = par[3 * m + 2]
            cm = (mz[2] / mz[0] + (-2.0 * (mz[1] / mz[0])) * (mz[1] / mz[0])) + (mz[1] / mz[0]) ** 2 * (mz[0] / mz[0])
            want = [fb.sum(), (~fb & (k == eps)).sum(), (~fb & (k == 10.0)).sum(), ((mz < 0).any(axis=0) | (cm < 0)).sum()]
            assert list(got[m]) == want, (m, got[m], want)
            assert min(want) > 0   # the batch exercises every counter
