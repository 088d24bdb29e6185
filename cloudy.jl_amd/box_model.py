"""The operator boundary: make_box_model_rhs (test/examples/utils/box_model_helpers.jl:22-53), batched.

    rhs = make_box_model_rhs(AnalyticalCoalStyle())          # same factory signature
    rhs(dm, m, par, t)                                        # same 4-argument in-place ODE function

`m`, `dm` are (nmom, n_parcels) device arrays (DeviceArray or CUDA torch tensors), i.e. Julia's m[parcel, moment].
`par` carries the fields rhs_coal! reads: pdists (types only), coal_data, NProgMoms, norms.
"""
from types import SimpleNamespace

import numpy as np

from . import _lib
from .device import as_device, dtype_code
from .EquationTypes import AnalyticalCoalStyle, CoalescenceStyle, FixedThreshold, MovingThreshold, NumericalCoalStyle
from .ParticleDistributions import nparams

EPS = float(np.finfo(np.float64).eps)


def ODEParameters(pdists, coal_data, NProgMoms, norms, **extra):
    """The ODE_parameters NamedTuple of the examples (e.g. box_gamma_mixture.jl:29-35; NumericalCoalStyle drivers pass
    coal_data=None and kernel_func=<normalised kernel function>, Numerical/n_particles_gamma.jl:37-38)."""
    return SimpleNamespace(pdists=tuple(pdists), coal_data=coal_data, NProgMoms=tuple(NProgMoms), norms=tuple(norms),
                           **extra)


def _numerical_plan_for(par, dtype=0):
    from .Coalescence import numerical_plan

    if tuple(nparams(d) for d in par.pdists) != tuple(par.NProgMoms):
        raise ValueError("NProgMoms must equal nparams of p.pdists")
    return numerical_plan([d.type_id for d in par.pdists], par.kernel_func, par.norms,
                          quad_order=getattr(par, "quad_order", 0), k_range=getattr(par, "k_range", (EPS, 10.0)),
                          dtype=dtype, quad_mode=getattr(par, "quad_mode", 1))  # default: converged mode


def _plan_for(par, dtype=0):
    if tuple(nparams(d) for d in par.pdists) != tuple(par.NProgMoms):
        raise ValueError("NProgMoms must equal nparams of p.pdists")
    if tuple(par.norms) != tuple(par.coal_data.norms):
        raise ValueError("p.norms differs from the norms the CoalescenceData was built with")
    if dtype == 1 and getattr(par, "fast_f32", False):
        dtype = 2  # CLOUDY_F32_FAST: single-precision arithmetic in the per-node Simpson / incomplete-gamma pass
    return par.coal_data.plan([d.type_id for d in par.pdists], k_range=getattr(par, "k_range", (EPS, 10.0)),
                              vel=getattr(par, "vel", ()), dtype=dtype)


def rhs_coal(coal_type, dmom, mom, p, threshold_style, stream=None):
    """rhs_coal! (box_model_helpers.jl:29-53): normalise, invert closures, get_coal_ints, de-normalise --
    one fused HIP kernel per call."""
    if not isinstance(coal_type, (AnalyticalCoalStyle, NumericalCoalStyle)):
        raise ValueError("Invalid coal style!")
    if dtype_code(mom) != dtype_code(dmom):
        raise TypeError("mom and dmom must have the same element type")
    if isinstance(coal_type, NumericalCoalStyle):
        # get_coal_ints(coal_type, p.pdists, p.kernel_func), box_model_helpers.jl:47-48
        plan = _numerical_plan_for(p, dtype_code(mom))
    else:
        if isinstance(threshold_style, MovingThreshold) != isinstance(p.coal_data.ts, MovingThreshold):
            raise ValueError("threshold style of the RHS does not match the CoalescenceData")
        plan = _plan_for(p, dtype_code(mom))  # float32 arrays select a CLOUDY_F32 plan (fp32 planes, fp64 arithmetic)
    mptr, planes, n, ld = as_device(mom)
    dptr, dplanes, dn, dld = as_device(dmom)
    if planes != plan.nmom or dplanes != plan.nmom or dn != n or dld != ld:
        raise ValueError(f"mom and dmom must both be ({plan.nmom}, n) with equal leading dimension")
    _lib.check(_lib.lib().cloudy_coal_rhs(plan.handle, n, ld, mptr, dptr, stream))
    return dmom


def make_box_model_rhs(coal_type, threshold_style=None):
    """make_box_model_rhs(coal_type::CoalescenceStyle, threshold_style::ThresholdStyle = FixedThreshold())."""
    if not isinstance(coal_type, CoalescenceStyle):
        raise TypeError("coal_type must be a CoalescenceStyle")
    ts = threshold_style if threshold_style is not None else FixedThreshold()

    def rhs(dm, m, par, t):
        return rhs_coal(coal_type, dm, m, par, ts)

    return rhs


def solve_ssprk33(par, u, dt, n_steps, out=None, stream=None, coal_type=None):
    """solve(ODEProblem(rhs, u, tspan, par), SSPRK33(), dt = dt) for n_steps fixed steps, on the device
    (cloudy_ssprk33_steps): the final state only (the examples' `sol.u[end]`).  `u` is advanced in place unless
    `out` is given.  `coal_type`: the style the RHS was made with (make_box_model_rhs(coal_type)); default
    AnalyticalCoalStyle, NumericalCoalStyle() integrates get_coal_ints(::NumericalCoalStyle, p.pdists, p.kernel_func)."""
    if coal_type is not None and not isinstance(coal_type, (AnalyticalCoalStyle, NumericalCoalStyle)):
        raise ValueError("Invalid coal style!")
    plan = _numerical_plan_for(par, dtype_code(u)) if isinstance(coal_type, NumericalCoalStyle) else _plan_for(par, dtype_code(u))
    uptr, planes, n, ld = as_device(u)
    o = out if out is not None else u
    optr, oplanes, on, old = as_device(o)
    if planes != plan.nmom or oplanes != plan.nmom or on != n or old != ld:
        raise ValueError(f"u and out must both be ({plan.nmom}, n) with equal leading dimension")
    _lib.check(_lib.lib().cloudy_ssprk33_steps(plan.handle, n, ld, uptr, optr, float(dt), int(n_steps), stream))
    return o


def solve_tsit5(par, u, dt, n_steps, out=None, stream=None, coal_type=None):
    """solve(ODEProblem(rhs, u, tspan, par), Tsit5(), dt = dt, adaptive = false) for n_steps fixed steps, on the device
    (cloudy_tsit5_steps): BASELINE configs[0] names Tsit5; no reference driver uses it (they call SSPRK33).  Every plan
    solve_ssprk33 serves (thresholds Inf / fixed / moving, NumericalCoalStyle through `coal_type`), fp64 or float planes.
    `u` is advanced in place unless `out` is given."""
    if coal_type is not None and not isinstance(coal_type, (AnalyticalCoalStyle, NumericalCoalStyle)):
        raise ValueError("Invalid coal style!")
    plan = _numerical_plan_for(par, dtype_code(u)) if isinstance(coal_type, NumericalCoalStyle) else _plan_for(par, dtype_code(u))
    uptr, planes, n, ld = as_device(u)
    o = out if out is not None else u
    optr, oplanes, on, old = as_device(o)
    if planes != plan.nmom or oplanes != plan.nmom or on != n or old != ld:
        raise ValueError(f"u and out must both be ({plan.nmom}, n) with equal leading dimension")
    _lib.check(_lib.lib().cloudy_tsit5_steps(plan.handle, n, ld, uptr, optr, float(dt), int(n_steps), stream))
    return o
