"""Batched sedimentation flux (src/Sources/Sedimentation.jl:22-37) and the rainshaft per-cell sources
(test/examples/utils/rainshaft_helpers.jl:52-78, without the inter-cell flux divergence)."""
import numpy as np

from . import _lib
from .device import DeviceArray, as_device, plane_dtype


def get_sedimentation_flux(plan, mom, out=None, stream=None):
    """mom (nmom, n) device, physical units -> flux (nmom, n) device, physical units."""
    ptr, planes, n, ld = as_device(mom)
    o = out if out is not None else DeviceArray(plan.nmom, n, plane_dtype(plan))
    _lib.check(_lib.lib().cloudy_sedimentation_flux(plan.handle, n, ld, ptr, as_device(o)[0], stream))
    return o


def rainshaft_sources(plan, mom, coal_source=None, sedi_flux=None, stream=None):
    ptr, planes, n, ld = as_device(mom)
    dt = plane_dtype(plan)
    cs = coal_source if coal_source is not None else DeviceArray(plan.nmom, n, dt)
    sf = sedi_flux if sedi_flux is not None else DeviceArray(plan.nmom, n, dt)
    _lib.check(_lib.lib().cloudy_rainshaft_sources(plan.handle, n, ld, ptr, as_device(cs)[0], as_device(sf)[0], stream))
    return cs, sf


def rhs_condensation(plan, dmom, mom, xi, s, stream=None):
    """rhs_condensation!(dmom, mom, p, s) (box_model_helpers.jl:55-67) -> get_cond_evap (Condensation.jl:22-37),
    batched.  `s`: scalar supersaturation or an (1, n) fp64 device array (one value per parcel); xi = p.xi."""
    ptr, planes, n, ld = as_device(mom)
    optr = as_device(dmom)[0]
    if hasattr(s, "data_ptr") or isinstance(s, DeviceArray):
        sptr, sval = as_device(s)[0], 0.0
    else:
        sptr, sval = None, float(s)
    _lib.check(_lib.lib().cloudy_cond_evap(plan.handle, n, ld, ptr, sptr, sval, float(xi), optr, stream))
    return dmom


def make_rainshaft_rhs(coal_type=None):
    """make_rainshaft_rhs(coal_type) (rainshaft_helpers.jl:45-89): returns rhs(m, p, t) -> d(m)/dt for a column
    (or a batch of columns stacked along the cell axis, `p.nz` cells each) -- coalescence source plus the upwind
    divergence of the sedimentation flux.  `m` is an (nmom, nz * n_columns) device array; `p` carries coal_data,
    pdists, NProgMoms, norms, vel and dz."""
    from .box_model import _plan_for
    from .device import dtype_code

    def rhs(m, p, t, out=None, work=None):
        plan = _plan_for(p, dtype_code(m))
        ptr, planes, n, ld = as_device(m)
        nz = int(getattr(p, "nz", n))
        if n % nz:
            raise ValueError("number of cells must be a multiple of p.nz")
        dt = plane_dtype(plan)
        o = out if out is not None else DeviceArray(plan.nmom, n, dt)
        w = work if work is not None else DeviceArray(plan.nmom, n, dt)
        _lib.check(_lib.lib().cloudy_rainshaft_rhs(plan.handle, nz, n // nz, ld, ptr, float(p.dz), as_device(w)[0],
                                                  as_device(o)[0], None))
        return o

    return rhs


def solve_rainshaft_ssprk33(par, u, n_steps, out=None, stream=None):
    """`solve(ODEProblem(make_rainshaft_rhs(...), m, tspan, p), SSPRK33(), dt = p.dt)` of the rainshaft drivers
    (rainshaft_gamma_mixture.jl:59-60), final state only, for a batch of independent columns of `par.nz` cells (fused in one launch up to 1024 cells -- workgroups of 256, 512 or
    1024 threads holding whole columns, picked by `nz` in the plan-time compiled kernel; taller columns stage by stage inside
    the library):
    one launch, state in registers, flux exchange through LDS (cloudy_rainshaft_ssprk33_steps).  `u` is advanced in
    place unless `out` is given."""
    from .box_model import _plan_for
    from .device import dtype_code

    plan = _plan_for(par, dtype_code(u))
    ptr, planes, n, ld = as_device(u)
    nz = int(getattr(par, "nz", n))
    if n % nz:
        raise ValueError("number of cells must be a multiple of par.nz")
    optr = as_device(out)[0] if out is not None else ptr
    _lib.check(_lib.lib().cloudy_rainshaft_ssprk33_steps(plan.handle, nz, n // nz, ld, ptr, optr, float(par.dz),
                                                         float(par.dt), int(n_steps), stream))
    return out if out is not None else u
