"""CoalescenceData and the batched get_coal_ints operator (src/Sources/Coalescence.jl:45-185)."""
import ctypes as C

import numpy as np

from . import _lib
from .device import DeviceArray, as_device
from .EquationTypes import (AnalyticalCoalStyle, FixedThreshold, MovingThreshold, NumericalCoalStyle, ThresholdStyle)
from .KernelFunctions import (ConstantKernelFunction, HydrodynamicKernelFunction, LinearKernelFunction,
                              LongKernelFunction)
from .KernelTensors import CoalescenceTensor, check_symmetry

EPS = float(np.finfo(np.float64).eps)


class Plan:
    """Owner of a cloudy_plan handle (include/cloudy_hip.h): one immutable constant block per configuration."""

    @staticmethod
    def make_desc(dist_types, kernel_c, dist_thresholds, norms, threshold_style, k_range=(EPS, 10.0),
                  n_bins_per_log_unit=15, vel=(), kernel_is_normalized=False, device=-1, dtype=0, specialize=0):
        """The cloudy_plan_desc of a configuration and the array its kernel_c pointer refers to (keep it alive)."""
        L = _lib.lib()
        d = _lib.PlanDesc()
        L.cloudy_plan_desc_init(C.byref(d))
        N = len(dist_types)
        if N > _lib.MAX_MODES:
            raise _lib.CloudyError(_lib.EUNSUPPORTED, f"more than {_lib.MAX_MODES} modes")
        d.n_modes = N
        for i, t in enumerate(dist_types):
            d.dist_type[i] = int(t)
        kc = np.ascontiguousarray(np.asarray(kernel_c, dtype=np.float64))
        d.tensor_p = kc.shape[-1]
        d.kernel_layout = 1 if kc.ndim == 4 else 0
        d.kernel_is_normalized = int(bool(kernel_is_normalized))
        d.kernel_c = kc.ctypes.data_as(C.POINTER(C.c_double))
        for i, t in enumerate(dist_thresholds):
            d.dist_thresholds[i] = float(t)
        d.threshold_style = 1 if threshold_style == 1 or isinstance(threshold_style, MovingThreshold) else 0
        d.norms[0], d.norms[1] = float(norms[0]), float(norms[1])
        d.k_range[0], d.k_range[1] = float(k_range[0]), float(k_range[1])
        d.n_bins_per_log_unit = int(n_bins_per_log_unit)
        v = np.asarray(vel, dtype=np.float64).reshape(-1, 2)
        if v.shape[0] > _lib.MAX_VEL:
            raise _lib.CloudyError(_lib.EUNSUPPORTED, f"more than {_lib.MAX_VEL} velocity terms")
        d.n_vel = v.shape[0]
        for i in range(v.shape[0]):
            d.vel[2 * i], d.vel[2 * i + 1] = v[i, 0], v[i, 1]
        d.device = int(device)
        d.dtype = int(dtype)  # CLOUDY_F64 = 0; CLOUDY_F32 = 1: float planes in HBM, fp64 arithmetic in registers
        # plan-time compiled kernels: 0 = when available, 1 = required, -1 = off
        d.specialize = int(specialize)
        return d, kc

    def __init__(self, dist_types, kernel_c, dist_thresholds, norms, threshold_style, k_range=(EPS, 10.0),
                 n_bins_per_log_unit=15, vel=(), kernel_is_normalized=False, device=-1, dtype=0, specialize=0):
        L = _lib.lib()
        d, self._kc = Plan.make_desc(dist_types, kernel_c, dist_thresholds, norms, threshold_style, k_range,
                                     n_bins_per_log_unit, vel, kernel_is_normalized, device, dtype, specialize)
        N = len(dist_types)
        self.dtype = int(dtype)
        h = C.c_void_p()
        _lib.check(L.cloudy_plan_create(C.byref(d), C.byref(h)))
        self.handle = h
        self.N = self.n_modes = N
        self.P = self.tensor_p = int(d.tensor_p)
        # which kernel family serves cloudy_coal_rhs (cloudy_hip.hip, build_host_plan): the streaming all-Inf kernel,
        # or the regime-sorted kernel with a Simpson pass (a finite FixedThreshold below the last mode / MovingThreshold)
        thr = [float(t) for t in dist_thresholds][:N]
        self.all_inf = (N == 1) or (d.threshold_style == 0 and all(t == float("inf") for t in thr[:N - 1]))
        self.nmom = L.cloudy_plan_nmom(h)
        self.specialized = bool(L.cloudy_plan_specialized(h))

    def jit_log(self):
        return _lib.lib().cloudy_plan_jit_log(self.handle).decode("utf-8", "replace")

    def get(self):
        """(N_mom_max, N_2d_ints, thresholds, normalised kernels [N,N,P,P], mom_norms) as the library holds them."""
        L = _lib.lib()
        nmm = C.c_int32()
        n2d = (C.c_int32 * self.N)()
        thr = np.zeros(self.N)
        kc = np.zeros((self.N, self.N, self.P, self.P))
        mn = np.zeros(self.nmom)
        dp = C.POINTER(C.c_double)
        _lib.check(L.cloudy_plan_get(self.handle, C.byref(nmm), n2d, thr.ctypes.data_as(dp), kc.ctypes.data_as(dp),
                                     mn.ctypes.data_as(dp)))
        return nmm.value, list(n2d), thr, kc, mn

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().cloudy_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kernel_func_code(kf):
    """(CLOUDY_KFUNC_* id, parameters) of a CoalescenceKernelFunction (KernelFunctions.jl:39-86)."""
    if isinstance(kf, ConstantKernelFunction):
        return 0, (kf.coll_coal_rate,)
    if isinstance(kf, LinearKernelFunction):
        return 1, (kf.coll_coal_rate,)
    if isinstance(kf, HydrodynamicKernelFunction):
        return 2, (kf.coal_eff,)
    if isinstance(kf, LongKernelFunction):
        return 3, (kf.x_threshold, kf.coal_rate_below_threshold, kf.coal_rate_above_threshold)
    raise TypeError("kernel_func must be a CoalescenceKernelFunction")


class NumericalPlan(Plan):
    """Plan of a NumericalCoalStyle operator: get_coal_ints(::NumericalCoalStyle, pdists, kernel_func)
    (Coalescence.jl:470-489).  quad_mode = QUAD_CONVERGED (the default): closed forms + one adaptive Gauss-Kronrod rule
    per mode (csrc/quad_conv.hpp), within 1e-8 of the reference's nested quadgk(rtol = 1e-8); QUAD_FIXED: one
    `quad_order`-point Gauss rule per distribution (csrc/quad.hpp; BASELINE configs[3] "10-pt Gauss quadrature").
    quad_order = 0: the mode's default (8 / 10).
    `kernel_func` is the normalised kernel function the reference passes in p.kernel_func
    (get_normalized_kernel_func(kernel, norms), Numerical/n_particles_gamma.jl:35)."""

    @staticmethod
    def make_desc(dist_types, kernel_func, norms, quad_order=0, k_range=(EPS, 10.0), device=-1, dtype=0, specialize=0,
                  kernel_func_is_normalized=True, quad_mode=1):
        L = _lib.lib()
        d = _lib.PlanDesc()
        L.cloudy_plan_desc_init(C.byref(d))
        N = len(dist_types)
        if N > _lib.MAX_MODES:
            raise _lib.CloudyError(_lib.EUNSUPPORTED, f"more than {_lib.MAX_MODES} modes")
        d.n_modes = N
        for i, t in enumerate(dist_types):
            d.dist_type[i] = int(t)
        d.norms[0], d.norms[1] = float(norms[0]), float(norms[1])
        d.k_range[0], d.k_range[1] = float(k_range[0]), float(k_range[1])
        d.device, d.dtype, d.specialize = int(device), int(dtype), int(specialize)
        d.coal_style = 1
        d.kernel_func, params = kernel_func_code(kernel_func)
        for i, v in enumerate(params):
            d.kernel_func_params[i] = float(v)
        d.kernel_func_is_normalized = int(bool(kernel_func_is_normalized))
        d.quad_order = int(quad_order)
        d.quad_mode = int(quad_mode)
        return d

    def __init__(self, dist_types, kernel_func, norms, quad_order=0, k_range=(EPS, 10.0), device=-1, dtype=0,
                 specialize=0, kernel_func_is_normalized=True, quad_mode=1):
        L = _lib.lib()
        d = NumericalPlan.make_desc(dist_types, kernel_func, norms, quad_order, k_range, device, dtype, specialize,
                                    kernel_func_is_normalized, quad_mode)
        N = len(dist_types)
        h = C.c_void_p()
        _lib.check(L.cloudy_plan_create(C.byref(d), C.byref(h)))
        self.handle = h
        self.dtype = int(dtype)
        self.N = self.n_modes = N
        self.P = self.tensor_p = 1
        self.quad_mode = int(quad_mode)
        self.quad_order = int(quad_order) or (8 if self.quad_mode == 1 else 10)
        self.all_inf = True
        self.numerical = True
        self.nmom = L.cloudy_plan_nmom(h)
        self.specialized = bool(L.cloudy_plan_specialized(h))


_numerical_plans = {}


QUAD_FIXED, QUAD_CONVERGED = 0, 1   # cloudy_plan_desc.quad_mode
F64, F32, F32_FAST, F64_RELAXED = 0, 1, 2, 3   # cloudy_plan_desc.dtype (F64_RELAXED: fp64 planes, incomplete-gamma series cut at 1e-11)


def numerical_plan(dist_types, kernel_func, norms, quad_order=0, k_range=(EPS, 10.0), dtype=0, specialize=0,
                   quad_mode=QUAD_CONVERGED):
    """Cached NumericalPlan (one per distinct configuration).  quad_mode: QUAD_CONVERGED (default) = the integrals split
    along the kernel function's non-smooth sets (closed forms + one adaptive Gauss-Kronrod rule per mode,
    csrc/quad_conv.hpp; quad_order = points per panel of the inner rule of a Lognormal mode, default 8) -- the mode that
    meets the reference's quadgk(rtol = 1e-8) answer; QUAD_FIXED = one quad_order-point Gauss rule per distribution
    (default 10), the explicit opt-in BASELINE configs[3] words."""
    key = (tuple(int(t) for t in dist_types), kernel_func, tuple(norms), int(quad_order), tuple(k_range), int(dtype),
           int(specialize), int(quad_mode))
    if key not in _numerical_plans:
        _numerical_plans[key] = NumericalPlan(key[0], kernel_func, norms, quad_order, k_range, dtype=dtype,
                                              specialize=specialize, quad_mode=quad_mode)
    return _numerical_plans[key]


class CoalescenceData:
    """CoalescenceData(kernel, NProgMoms, dist_thresholds, norms=(1, 1), ts=FixedThreshold())
    (Coalescence.jl:55-104).  `kernel` is one CoalescenceTensor or an N x N nested tuple of them."""

    def __init__(self, kernel, NProgMoms, dist_thresholds, norms=(1.0, 1.0), ts=None):
        ts = ts if ts is not None else FixedThreshold()
        if not isinstance(ts, ThresholdStyle):
            raise TypeError("ts must be a ThresholdStyle")
        N = len(NProgMoms)
        if len(dist_thresholds) != N:
            raise ValueError("dist_thresholds must have one entry per distribution")
        if isinstance(kernel, CoalescenceTensor):
            kc = kernel.c
            self.P = kernel.P
        else:
            if len(kernel) != N or any(len(row) != N for row in kernel):
                raise ValueError("kernel must be an N x N tuple of CoalescenceTensor")
            P = kernel[0][0].P
            kc = np.zeros((N, N, P, P))
            for j in range(N):
                for k in range(N):
                    check_symmetry(kernel[j][k].c)
                    kc[j, k] = kernel[j][k].c
            self.P = P
        self.kernel_c = np.array(kc, dtype=np.float64)
        self.NProgMoms = tuple(int(x) for x in NProgMoms)
        self.dist_thresholds_in = tuple(float(t) for t in dist_thresholds)
        self.norms = (float(norms[0]), float(norms[1]))
        self.ts = ts
        self.N = N
        # derived fields, Coalescence.jl:69-84
        self.N_mom_max = max(self.NProgMoms) + (self.P - 1)
        self.N_2d_ints = tuple(
            (self.P - 1) + (max(self.NProgMoms[i], self.NProgMoms[i + 1]) if i < N - 1 else self.NProgMoms[i])
            for i in range(N))
        if isinstance(ts, FixedThreshold):
            self.dist_thresholds = tuple(t / self.norms[1] for t in self.dist_thresholds_in)
        else:
            self.dist_thresholds = self.dist_thresholds_in
        self._plans = {}

    def plan(self, dist_types, k_range=(EPS, 10.0), vel=(), dtype=0, specialize=0):
        """The device plan for this data and the closure types of `pdists` (built once, cached).  `specialize`:
        plan-time compiled kernels for all-Inf thresholds (0 = when available, 1 = required, -1 = off)."""
        dist_types = tuple(int(t) for t in dist_types)
        for t, npm in zip(dist_types, self.NProgMoms):
            if {0: 2, 1: 3, 2: 2, 3: 3}.get(t) != npm:
                raise ValueError("NProgMoms does not match nparams of the distributions")
        key = (dist_types, tuple(k_range), tuple(map(tuple, np.asarray(vel, dtype=float).reshape(-1, 2))), int(dtype),
               int(specialize))
        if key not in self._plans:
            self._plans[key] = Plan(dist_types, self.kernel_c, self.dist_thresholds_in, self.norms, self.ts,
                                    k_range=k_range, vel=vel, dtype=dtype, specialize=specialize)
        return self._plans[key]


def get_coal_ints(cs, pdists, coal_data, ts=None, out=None, stream=None, k_range=(EPS, 10.0), quad_order=0,
                  quad_mode=QUAD_CONVERGED):
    """get_coal_ints(::AnalyticalCoalStyle, pdists, coal_data[, ::MovingThreshold])  (Coalescence.jl:115-185) and
    get_coal_ints(::NumericalCoalStyle, pdists, kernel_func) (:470-489; third argument = the normalised kernel
    function; quad_mode / quad_order as numerical_plan), batched: `pdists` = (dist_types, params) with params a
    (3N, n) device array of (n, theta, k) in normalised units.  Returns the (nmom, n) device array of normalised
    tendencies."""
    dist_types, params = pdists
    if isinstance(cs, NumericalCoalStyle):
        plan = numerical_plan(dist_types, coal_data, (1.0, 1.0), quad_order, k_range, quad_mode=quad_mode)
    elif not isinstance(cs, AnalyticalCoalStyle):
        raise ValueError("Invalid coal style!")
    else:
        if isinstance(ts, MovingThreshold) != isinstance(coal_data.ts, MovingThreshold):
            raise ValueError("threshold style of the call does not match the CoalescenceData")
        plan = coal_data.plan(dist_types, k_range=k_range)
    ptr, planes, n, ld = as_device(params)
    if planes != 3 * plan.N:
        raise ValueError("params must have 3N planes")
    o = out if out is not None else DeviceArray(plan.nmom, n)
    optr, _, _, old = as_device(o)
    if old != ld:
        raise ValueError("out must have the same leading dimension as params")
    _lib.check(_lib.lib().cloudy_get_coal_ints(plan.handle, n, ld, ptr, optr, stream))
    return o


def get_finite_2d_integrals(plan, params, out=None, stream=None):
    """get_finite_2d_integrals (Coalescence.jl:200-244), batched: (N*M*M, n) device array, planes (i, p1, p2)."""
    ptr, planes, n, ld = as_device(params)
    M = plan.P + 2
    o = out if out is not None else DeviceArray(plan.N * M * M, n)
    _lib.check(_lib.lib().cloudy_finite_2d_integrals(plan.handle, n, ld, ptr, as_device(o)[0], stream))
    return o
