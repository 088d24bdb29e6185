"""ctypes binding of the CPU oracle (oracle/cloudy_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package (cloudy.jl_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcloudy_oracle.so")

MAX_MODES, MAX_P, MAX_VEL = 8, 8, 8
EXPONENTIAL, GAMMA, MONODISPERSE, LOGNORMAL = 0, 1, 2, 3
FIXED_THRESHOLD, MOVING_THRESHOLD = 0, 1


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("cloudy_oracle.c", "cloudy_oracle_quad.c", "cloudy_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "libcloudy_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Dist(C.Structure):
    _fields_ = [("type", C.c_int), ("n", C.c_double), ("theta", C.c_double), ("k", C.c_double)]

    def __repr__(self):
        return f"Dist(type={self.type}, n={self.n!r}, theta={self.theta!r}, k={self.k!r})"


class CoalData(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("P", C.c_int), ("N_mom_max", C.c_int),
        ("N_2d_ints", C.c_int * MAX_MODES),
        ("dist_thresholds", C.c_double * MAX_MODES),
        ("c", C.c_double * (MAX_MODES * MAX_MODES * MAX_P * MAX_P)),
    ]


class Params(C.Structure):
    _fields_ = [
        ("N", C.c_int),
        ("dist_type", C.c_int * MAX_MODES),
        ("NProgMoms", C.c_int * MAX_MODES),
        ("norms", C.c_double * 2),
        ("k_range", C.c_double * 2),
        ("threshold_style", C.c_int),
        ("coal_data", CoalData),
        ("n_vel", C.c_int),
        ("vel", C.c_double * (MAX_VEL * 2)),
    ]


KF_CONSTANT, KF_LINEAR, KF_HYDRODYNAMIC, KF_LONG = 0, 1, 2, 3


class KernelFunc(C.Structure):
    """co_kernel_func: CoalescenceKernelFunction (KernelFunctions.jl:39-86) as (kind, parameters)."""
    _fields_ = [("kind", C.c_int), ("p", C.c_double * 3)]


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.co_gamma.restype = C.c_double
        L.co_gamma.argtypes = [C.c_double]
        L.co_gamma_inc_p.restype = C.c_double
        L.co_gamma_inc_p.argtypes = [C.c_double, C.c_double]
        L.co_gamma_inc_inv.restype = C.c_double
        L.co_gamma_inc_inv.argtypes = [C.c_double] * 3
        L.co_get_dist_moment_ind.argtypes = [_ip, C.c_int, C.c_int, C.c_int]
        L.co_get_dist_moments_ind_range.argtypes = [_ip, C.c_int, C.c_int, _ip, _ip]
        L.co_get_moments_normalizing_factors.argtypes = [_ip, C.c_int, _dp, _dp]
        L.co_check_symmetry.argtypes = [_dp, C.c_int]
        L.co_get_normalized_kernel_tensor.argtypes = [_dp, C.c_int, _dp, _dp]
        for f in ("co_constant_kernel", "co_linear_kernel", "co_hydrodynamic_kernel"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double] * 3
        L.co_long_kernel.restype = C.c_double
        L.co_long_kernel.argtypes = [C.c_double] * 5
        L.co_nparams.argtypes = [C.c_int]
        L.co_dist_valid.argtypes = [C.POINTER(Dist)]
        L.co_moment.restype = C.c_double
        L.co_moment.argtypes = [C.POINTER(Dist), C.c_double]
        L.co_get_moments.argtypes = [C.POINTER(Dist), _dp]
        L.co_partial_moment.restype = C.c_double
        L.co_partial_moment.argtypes = [C.POINTER(Dist), C.c_double, C.c_double]
        L.co_get_standard_N_q.argtypes = [C.POINTER(Dist), C.c_int, C.c_double, _dp]
        L.co_density.restype = C.c_double
        L.co_density.argtypes = [C.POINTER(Dist), C.c_double]
        L.co_normed_density.restype = C.c_double
        L.co_normed_density.argtypes = [C.POINTER(Dist), C.c_double]
        L.co_update_dist_from_moments.argtypes = [C.c_int, _dp, C.c_int, _dp, C.POINTER(Dist)]
        L.co_moment_source_helper.restype = C.c_double
        L.co_moment_source_helper.argtypes = [C.POINTER(Dist), C.c_double, C.c_double, C.c_double, C.c_int]
        L.co_moment_source_helper_lognormal.restype = C.c_double
        L.co_moment_source_helper_lognormal.argtypes = [C.POINTER(Dist), C.c_double, C.c_double, C.c_double, C.c_int]
        L.co_compute_threshold.restype = C.c_double
        L.co_compute_threshold.argtypes = [C.POINTER(Dist), C.c_double, C.c_double]
        L.co_compute_thresholds.argtypes = [C.POINTER(Dist), C.c_int, _dp, _dp]
        L.co_coalescence_data_init.argtypes = [C.POINTER(CoalData), C.c_int, C.c_int, _dp, _ip, _dp, _dp, C.c_int]
        L.co_get_coal_ints.argtypes = [C.POINTER(Dist), C.POINTER(CoalData), C.c_int, _dp, _dp]
        L.co_get_moments_matrix.argtypes = [C.POINTER(Dist), C.c_int, C.c_int, C.c_int, _dp]
        L.co_get_finite_2d_integrals.argtypes = [C.POINTER(Dist), C.c_int, C.c_int, _dp, _dp, _ip, _dp]
        L.co_weighting_fn.restype = C.c_double
        L.co_weighting_fn.argtypes = [C.c_double, C.c_int, C.POINTER(Dist), C.c_int]
        L.co_get_sedimentation_flux.argtypes = [C.POINTER(Dist), C.c_int, _dp, C.c_int, _dp]
        L.co_get_cond_evap.argtypes = [C.POINTER(Dist), C.c_int, C.c_double, C.c_double, C.c_double, _dp]
        L.co_rhs_condensation_batch.argtypes = [C.POINTER(Params), C.c_double, _dp, C.c_double, C.c_long, C.c_long, _dp, _dp]
        L.co_rhs_coal.argtypes = [C.POINTER(Params), _dp, _dp, _dp]
        L.co_rhs_coal_batch.argtypes = [C.POINTER(Params), C.c_long, C.c_long, _dp, _dp, _dp, C.c_int]
        L.co_rainshaft_cell_batch.argtypes = [C.POINTER(Params), C.c_long, C.c_long, _dp, _dp, _dp, C.c_int]
        L.co_update_dist_batch.argtypes = [C.POINTER(Params), C.c_long, C.c_long, _dp, _dp]
        # cloudy_oracle_quad.c: NumericalCoalStyle with a fixed Gauss rule
        L.co_kernel_func_eval.restype = C.c_double
        L.co_kernel_func_eval.argtypes = [C.POINTER(KernelFunc), C.c_double, C.c_double]
        L.co_get_normalized_kernel_func.argtypes = [C.POINTER(KernelFunc), _dp, C.POINTER(KernelFunc)]
        L.co_gauss_gamma_rule.argtypes = [C.c_int, C.c_double, _dp, _dp]
        L.co_gauss_hermite_rule.argtypes = [C.c_int, _dp, _dp]
        L.co_dist_rule.argtypes = [C.POINTER(Dist), C.c_int, _dp, _dp]
        L.co_get_coal_ints_numerical_fixed.argtypes = [C.POINTER(Dist), C.c_int, C.POINTER(KernelFunc), C.c_int, _dp, _dp, _dp]
        L.co_rhs_coal_numerical_batch.argtypes = [C.POINTER(Params), C.POINTER(KernelFunc), C.c_int, C.c_long, C.c_long,
                                                  _dp, _dp, _dp, _dp, C.c_int]
        L.co_inc_beta.restype = C.c_double
        L.co_inc_beta.argtypes = [C.c_double, C.c_double, C.c_double]
        L.co_gauss_legendre_rule.argtypes = [C.c_int, _dp, _dp]
        L.co_get_coal_ints_numerical_converged.argtypes = [C.POINTER(Dist), C.c_int, C.POINTER(KernelFunc), C.c_int, C.c_double,
                                                           _dp, _dp]
        L.co_conv_node_count.argtypes = [C.c_int]
        L.co_conv_node_count.restype = C.c_long
        L.co_rhs_coal_numerical_converged_batch.argtypes = [C.POINTER(Params), C.POINTER(KernelFunc), C.c_int, C.c_double,
                                                            C.c_long, C.c_long, _dp, _dp, _dp, C.c_int]
        L.co_get_coal_ints_numerical_adaptive.argtypes = [C.POINTER(Dist), C.c_int, C.POINTER(KernelFunc), C.c_double,
                                                          C.c_double, _dp, _dp, _dp, _dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _darr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def _iarr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))


# ---- special functions ------------------------------------------------------------------
def gamma(x):
    return lib().co_gamma(float(x))


def gamma_inc_p(a, x):
    return lib().co_gamma_inc_p(float(a), float(x))


def gamma_inc_inv(a, p, q=None):
    return lib().co_gamma_inc_inv(float(a), float(p), float(1.0 - p if q is None else q))


# ---- helper_functions.jl ----------------------------------------------------------------
def get_dist_moment_ind(NProgMoms, i, m):
    a = _iarr(NProgMoms)
    r = lib().co_get_dist_moment_ind(a.ctypes.data_as(_ip), len(a), int(i), int(m))
    if r < 0:
        raise ValueError("moment index must be positive integer and equal or smaller than the dist number "
                         "of prognostic moments!!!")
    return r


def get_dist_moments_ind_range(NProgMoms, i):
    a = _iarr(NProgMoms)
    f, l = C.c_int(), C.c_int()
    if lib().co_get_dist_moments_ind_range(a.ctypes.data_as(_ip), len(a), int(i), C.byref(f), C.byref(l)) < 0:
        raise IndexError("distribution index out of range")
    return range(f.value, l.value + 1)


def get_moments_normalizing_factors(NProgMoms, norms):
    a = _iarr(NProgMoms)
    nn = _darr(norms)
    out = np.zeros(int(a.sum()))
    r = lib().co_get_moments_normalizing_factors(a.ctypes.data_as(_ip), len(a), _d(nn), _d(out))
    if r < 0:
        raise ValueError("norms must be positive!")
    return out


# ---- KernelTensors.jl ---------------------------------------------------------------------
def check_symmetry(c):
    c = _darr(c)
    if c.ndim != 2 or c.shape[0] != c.shape[1]:
        raise ValueError("array needs to be quadratic in order to be symmetric.")
    if lib().co_check_symmetry(_d(c), c.shape[0]) != 0:
        raise ValueError("array not symmetric.")


def get_normalized_kernel_tensor(c, norms):
    c = _darr(c)
    out = np.zeros_like(c)
    lib().co_get_normalized_kernel_tensor(_d(c), c.shape[0], _d(_darr(norms)), _d(out))
    return out


# ---- ParticleDistributions.jl -------------------------------------------------------------
def make_dist(dist_type, n, theta, k=1.0):
    d = Dist(int(dist_type), float(n), float(theta), float(k))
    if not lib().co_dist_valid(C.byref(d)):
        raise ValueError("invalid distribution parameters")
    return d


def nparams(dist_type):
    return lib().co_nparams(int(dist_type))


def moment(d, q):
    return lib().co_moment(C.byref(d), float(q))


def get_moments(d):
    out = np.zeros(nparams(d.type))
    lib().co_get_moments(C.byref(d), _d(out))
    return out


def partial_moment(d, q, x_threshold):
    return lib().co_partial_moment(C.byref(d), float(q), float(x_threshold))


def get_standard_N_q(pdists, size_cutoff=1e-6):
    """(N_liq, N_rai, M_liq, M_rai), ParticleDistributions.jl:634-687."""
    arr = (Dist * len(pdists))(*pdists)
    out = np.zeros(4)
    lib().co_get_standard_N_q(arr, len(pdists), float(size_cutoff), _d(out))
    return out


def density(d, x):
    if x < 0:
        raise ValueError("Density can only be evaluated at nonnegative values.")
    return lib().co_density(C.byref(d), float(x))


def normed_density(d, x):
    if x < 0:
        raise ValueError("Density can only be evaluated at nonnegative values.")
    return lib().co_normed_density(C.byref(d), float(x))


def update_dist_from_moments(dist_type, moments, k_range=None):
    m = _darr(moments)
    out = Dist()
    kr = _darr(k_range) if k_range is not None else None
    r = lib().co_update_dist_from_moments(int(dist_type), _d(m), len(m), _d(kr) if kr is not None else None,
                                          C.byref(out))
    if r != 0:
        raise TypeError("no method matching update_dist_from_moments for this number of moments")
    return out


def moment_source_helper(d, p1, p2, x_threshold, n_bins_per_log_unit=15):
    return lib().co_moment_source_helper(C.byref(d), float(p1), float(p2), float(x_threshold),
                                         int(n_bins_per_log_unit))


def compute_threshold(d, percentile=0.97, minx=1e-18):
    return lib().co_compute_threshold(C.byref(d), float(percentile), float(minx))


def compute_thresholds(pdists, percentiles=None):
    arr = (Dist * len(pdists))(*pdists)
    out = np.zeros(len(pdists))
    pc = _darr(percentiles) if percentiles is not None else None
    lib().co_compute_thresholds(arr, len(pdists), _d(pc) if pc is not None else None, _d(out))
    return out


# ---- Coalescence.jl -------------------------------------------------------------------------
def coalescence_data(kernel_c, NProgMoms, dist_thresholds, norms=(1.0, 1.0), threshold_style=FIXED_THRESHOLD):
    """kernel_c: [P,P] (one tensor for all pairs, Coalescence.jl:89-104) or [N,N,P,P]."""
    npm = _iarr(NProgMoms)
    N = len(npm)
    kc = _darr(kernel_c)
    if kc.ndim == 2:
        kc = np.ascontiguousarray(np.broadcast_to(kc, (N, N) + kc.shape))
    P = kc.shape[-1]
    cd = CoalData()
    r = lib().co_coalescence_data_init(C.byref(cd), N, P, _d(kc), npm.ctypes.data_as(_ip),
                                       _d(_darr(dist_thresholds)), _d(_darr(norms)), int(threshold_style))
    if r == -2:
        raise ValueError("array not symmetric.")
    if r != 0:
        raise ValueError("invalid CoalescenceData arguments")
    return cd


def get_coal_ints(pdists, coal_data, threshold_style=FIXED_THRESHOLD, with_scale=False):
    arr = (Dist * len(pdists))(*pdists)
    nm = sum(nparams(d.type) for d in pdists)
    out, sc = np.zeros(nm), np.zeros(nm)
    lib().co_get_coal_ints(arr, C.byref(coal_data), int(threshold_style), _d(out), _d(sc))
    return (out, sc) if with_scale else out


def get_moments_matrix(pdists, M, N_mom_max):
    arr = (Dist * len(pdists))(*pdists)
    out = np.zeros((len(pdists), M))
    lib().co_get_moments_matrix(arr, len(pdists), M, N_mom_max, _d(out))
    return out


def get_finite_2d_integrals(pdists, thresholds, moments, N_2d_ints):
    arr = (Dist * len(pdists))(*pdists)
    N, M = moments.shape
    F = np.zeros((N, M, M))
    n2 = _iarr(N_2d_ints)
    lib().co_get_finite_2d_integrals(arr, N, M, _d(_darr(thresholds)), _d(_darr(moments)),
                                     n2.ctypes.data_as(_ip), _d(F))
    return F


def weighting_fn(x, k, pdists):
    arr = (Dist * len(pdists))(*pdists)
    r = lib().co_weighting_fn(float(x), int(k), arr, len(pdists))
    if r != r:
        raise AssertionError("k out of range")
    return r


def get_sedimentation_flux(pdists, vel):
    arr = (Dist * len(pdists))(*pdists)
    v = _darr(vel).reshape(-1, 2)
    nm = sum(nparams(d.type) for d in pdists)
    out = np.zeros(nm)
    lib().co_get_sedimentation_flux(arr, len(pdists), _d(v), v.shape[0], _d(out))
    return out


def get_cond_evap(pdists, s, xi, rho_l=1000.0):
    arr = (Dist * len(pdists))(*pdists)
    nm = sum(nparams(d.type) for d in pdists)
    out = np.zeros(nm)
    lib().co_get_cond_evap(arr, len(pdists), float(s), float(xi), float(rho_l), _d(out))
    return out


def rhs_condensation_batch(p, xi, s, mom):
    """rhs_condensation! (box_model_helpers.jl:55-67) for a batch; s scalar or per-parcel array."""
    m = _darr(mom)
    nm, n = m.shape
    d = np.empty_like(m)
    sa = _darr(s) if np.ndim(s) else None
    lib().co_rhs_condensation_batch(C.byref(p), float(xi), _d(sa) if sa is not None else None,
                                    0.0 if sa is not None else float(s), n, n, _d(m), _d(d))
    return d


# ---- box_model_helpers.jl / rainshaft_helpers.jl --------------------------------------------
def make_params(dist_types, kernel_c, dist_thresholds, norms=(1.0, 1.0), threshold_style=FIXED_THRESHOLD,
                k_range=(np.finfo(np.float64).eps, 10.0), vel=()):
    p = Params()
    N = len(dist_types)
    p.N = N
    npm = [nparams(t) for t in dist_types]
    for i in range(N):
        p.dist_type[i] = int(dist_types[i])
        p.NProgMoms[i] = npm[i]
    p.norms[0], p.norms[1] = float(norms[0]), float(norms[1])
    p.k_range[0], p.k_range[1] = float(k_range[0]), float(k_range[1])
    p.threshold_style = int(threshold_style)
    p.coal_data = coalescence_data(kernel_c, npm, dist_thresholds, norms, threshold_style)
    v = _darr(vel).reshape(-1, 2)
    p.n_vel = v.shape[0]
    for i in range(v.shape[0]):
        p.vel[2 * i], p.vel[2 * i + 1] = v[i, 0], v[i, 1]
    return p


def nmom_of(p):
    return sum(p.NProgMoms[i] for i in range(p.N))


def rhs_coal(p, mom, with_scale=False):
    m = _darr(mom)
    d, s = np.zeros_like(m), np.zeros_like(m)
    if lib().co_rhs_coal(C.byref(p), _d(m), _d(d), _d(s)) < 0:
        raise ValueError("rhs_coal failed")
    return (d, s) if with_scale else d


def rhs_coal_batch(p, mom, with_scale=False, n_threads=0, out=None):
    """mom: [nmom, n_parcels] moment-major (C-contiguous) -> dmom of the same shape (`out`: preallocated result)."""
    m = _darr(mom)
    nm, n = m.shape
    assert nm == nmom_of(p)
    if out is not None:
        assert out.shape == m.shape and out.dtype == np.float64 and out.flags.c_contiguous
    d = out if out is not None else np.empty_like(m)
    s = np.empty_like(m) if with_scale else None
    if lib().co_rhs_coal_batch(C.byref(p), n, n, _d(m), _d(d), _d(s) if s is not None else None, int(n_threads)) < 0:
        raise ValueError("rhs_coal_batch failed")
    return (d, s) if with_scale else d


def rainshaft_cell_batch(p, mom, n_threads=0):
    m = _darr(mom)
    nm, n = m.shape
    cs, sf = np.empty_like(m), np.empty_like(m)
    lib().co_rainshaft_cell_batch(C.byref(p), n, n, _d(m), _d(cs), _d(sf), int(n_threads))
    return cs, sf


def update_dist_batch(p, mom):
    m = _darr(mom)
    nm, n = m.shape
    out = np.empty((3 * p.N, n))
    if lib().co_update_dist_batch(C.byref(p), n, n, _d(m), _d(out)) < 0:
        raise ValueError("update_dist_batch failed")
    return out


def check_moment_consistency(m):
    """ParticleDistributions.jl:437-449: 0 = consistent (the reference returns nothing), 1 / 2 = which of its two checks throws"""
    a = _darr(np.asarray(m, dtype=np.float64))
    f = lib().co_check_moment_consistency
    f.argtypes, f.restype = [C.POINTER(C.c_double), C.c_int], C.c_int
    return int(f(_d(a), int(a.size)))


def closure_stats(p, mom):
    """(N, 4) counts per mode: fallback (0, 1, 1), shape at the lower clamp, at the upper clamp, inconsistent moments"""
    m = _darr(mom)
    nm, n = m.shape
    out = (C.c_ulonglong * (4 * p.N))()
    f = lib().co_closure_stats
    f.argtypes, f.restype = [C.c_void_p, C.c_long, C.c_long, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)], C.c_int
    if f(C.cast(C.byref(p), C.c_void_p), n, n, _d(m), out) < 0:
        raise ValueError("closure_stats failed")
    return np.array(out[:], dtype=np.int64).reshape(p.N, 4)


def max_threads():
    return lib().co_max_threads()


# ---- Coalescence.jl NumericalCoalStyle with a fixed Gauss rule (cloudy_oracle_quad.c) --------------------------
def kernel_func(kind, *params):
    kf = KernelFunc()
    kf.kind = int(kind)
    for i, v in enumerate(params):
        kf.p[i] = float(v)
    return kf


def kernel_func_eval(kf, x, y):
    return lib().co_kernel_func_eval(C.byref(kf), float(x), float(y))


def get_normalized_kernel_func(kf, norms):
    out = KernelFunc()
    if lib().co_get_normalized_kernel_func(C.byref(kf), _d(_darr(norms)), C.byref(out)) < 0:
        raise ValueError("unknown kernel function")
    return out


def gauss_gamma_rule(nq, k):
    """(u, W): nq-point rule for the normalised weight u^(k-1) e^-u / Gamma(k)."""
    u, w = np.zeros(nq), np.zeros(nq)
    if lib().co_gauss_gamma_rule(int(nq), float(k), _d(u), _d(w)) < 0:
        raise ValueError("bad rule arguments")
    return u, w


def gauss_hermite_rule(nq):
    t, w = np.zeros(nq), np.zeros(nq)
    if lib().co_gauss_hermite_rule(int(nq), _d(t), _d(w)) < 0:
        raise ValueError("bad rule arguments")
    return t, w


def get_coal_ints_numerical_fixed(pdists, kf, nq=10, with_scale=False):
    arr = (Dist * len(pdists))(*pdists)
    nmom = sum(nparams(d.type) for d in pdists)
    out, sc = np.zeros(nmom), np.zeros(nmom)
    if lib().co_get_coal_ints_numerical_fixed(arr, len(pdists), C.byref(kf), int(nq), _d(out), _d(sc), None) < 0:
        raise ValueError("get_coal_ints_numerical_fixed failed (Monodisperse has no normed density)")
    return (out, sc) if with_scale else out


CONV_TOL = 1e-7   # acceptance tolerance of the converged mode's adaptive rules (kConvTol in csrc/quad_conv.hpp), every kernel function


def get_coal_ints_numerical_converged(pdists, kf, q=8, tol=CONV_TOL, with_scale=False, with_nodes=False):
    """get_coal_ints(::NumericalCoalStyle, ...) in converged mode (cloudy_oracle_quad.c): closed forms for Q and R, one
    adaptive Gauss-Kronrod (7, 15) rule per mode for the weighting_fn split (acceptance tolerance tol; q = points per panel
    of the inner rule of a Lognormal mode).  with_nodes: also the number of integrand evaluations of the adaptive rules."""
    arr = (Dist * len(pdists))(*pdists)
    nmom = sum(nparams(d.type) for d in pdists)
    out, sc = np.zeros(nmom), np.zeros(nmom)
    lib().co_conv_node_count(1)
    r = lib().co_get_coal_ints_numerical_converged(arr, len(pdists), C.byref(kf), int(q), float(tol), _d(out), _d(sc))
    nodes = lib().co_conv_node_count(1)
    if r < 0:
        raise ValueError("get_coal_ints_numerical_converged: " + ("Lognormal / Monodisperse modes are not served"
                                                                  if r == -2 else "bad arguments"))
    if with_nodes:
        return (out, sc, nodes) if with_scale else (out, nodes)
    return (out, sc) if with_scale else out


def conv_set_ln_inner(sigmas=3.0, max_panels=256):
    """inner rule of a Lognormal mode's T_m in converged mode: panel width in sigma and the cap on the panel count (tests use a
    finer rule as the reference of the default one; call without arguments to restore the default)"""
    f = lib().co_conv_set_ln_inner
    f.argtypes, f.restype = [C.c_double, C.c_int], None
    f(float(sigmas), int(max_panels))


def get_coal_ints_numerical_adaptive(pdists, kf, eps_outer=1e-10, eps_inner=1e-12):
    """get_coal_ints(::NumericalCoalStyle, ...) by nested adaptive Gauss-Kronrod quadrature (cloudy_oracle_adaptive.c):
    (coal_ints, Q[orders, N, N], R[orders, N, N], S[orders, 2, N])."""
    arr = (Dist * len(pdists))(*pdists)
    N = len(pdists)
    orders = max(nparams(d.type) for d in pdists)
    nmom = sum(nparams(d.type) for d in pdists)
    out, Q, R, S = np.zeros(nmom), np.zeros((orders, N, N)), np.zeros((orders, N, N)), np.zeros((orders, 2, N))
    if lib().co_get_coal_ints_numerical_adaptive(arr, N, C.byref(kf), float(eps_outer), float(eps_inner), _d(out), _d(Q),
                                                 _d(R), _d(S)) < 0:
        raise ValueError("get_coal_ints_numerical_adaptive failed")
    return out, Q, R, S


def rhs_coal_numerical_batch(p, kf_normalized, nq, mom, with_scale=False, n_threads=0, out=None, with_noise=False):
    """rhs_coal!(NumericalCoalStyle(), ...) for a moment-major batch; p: make_params(...) (its tensors are unused).
    with_noise: also the absolute rounding error of the reference's (1 - weighting_fn) form (cloudy_oracle_quad.c)."""
    m = _darr(mom)
    nm, n = m.shape
    assert nm == nmom_of(p)
    d = out if out is not None else np.empty_like(m)
    s = np.empty_like(m) if (with_scale or with_noise) else None
    z = np.empty_like(m) if with_noise else None
    if lib().co_rhs_coal_numerical_batch(C.byref(p), C.byref(kf_normalized), int(nq), n, n, _d(m), _d(d),
                                         _d(s) if s is not None else None, _d(z) if z is not None else None,
                                         int(n_threads)) < 0:
        raise ValueError("rhs_coal_numerical_batch failed")
    if with_noise:
        return d, s, z
    return (d, s) if with_scale else d


def rhs_coal_numerical_converged_batch(p, kf_normalized, q, mom, tol=CONV_TOL, with_scale=False, n_threads=0, out=None):
    """rhs_coal!(NumericalCoalStyle(), ...) in converged mode for a moment-major batch (same-rule restatement of
    csrc/quad_conv.hpp); p: make_params(...) (its tensors are unused)."""
    m = _darr(mom)
    nm, n = m.shape
    assert nm == nmom_of(p)
    d = out if out is not None else np.empty_like(m)
    s = np.empty_like(m) if with_scale else None
    if lib().co_rhs_coal_numerical_converged_batch(C.byref(p), C.byref(kf_normalized), int(q), float(tol), n, n, _d(m),
                                                   _d(d), _d(s) if s is not None else None, int(n_threads)) < 0:
        raise ValueError("rhs_coal_numerical_converged_batch failed (Lognormal modes are not served in converged mode)")
    return (d, s) if with_scale else d
