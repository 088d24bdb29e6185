"""Closure families and their batched inversion (src/ParticleDistributions/ParticleDistributions.jl).

A distribution object is the host-side descriptor the reference passes as `p.pdists[i]`: in rhs_coal! only its
TYPE matters (its parameters are overwritten by update_dist_from_moments, box_model_helpers.jl:32-38).  Batched
parameter arrays live on the GPU as 3N planes (n, theta, k) per mode.
"""
from dataclasses import dataclass

import numpy as np

from . import _lib
from .device import DeviceArray, as_device, plane_dtype

EXPONENTIAL, GAMMA, MONODISPERSE, LOGNORMAL = 0, 1, 2, 3
NPARAMS = {EXPONENTIAL: 2, GAMMA: 3, MONODISPERSE: 2, LOGNORMAL: 3}


class AbstractParticleDistribution:
    pass


class PrimitiveParticleDistribution(AbstractParticleDistribution):
    pass


@dataclass(frozen=True)
class ExponentialPrimitiveParticleDistribution(PrimitiveParticleDistribution):
    n: float
    θ: float
    type_id = EXPONENTIAL

    def __post_init__(self):  # ParticleDistributions.jl:72-75
        if self.n < 0 or self.θ <= 0:
            raise ValueError("n needs to be nonnegative. θ needs to be positive.")


@dataclass(frozen=True)
class GammaPrimitiveParticleDistribution(PrimitiveParticleDistribution):
    n: float
    θ: float
    k: float
    type_id = GAMMA

    def __post_init__(self):  # ParticleDistributions.jl:101-104
        if self.n < 0 or self.θ <= 0 or self.k <= 0:
            raise ValueError("n needs to be nonnegative. θ and k need to be positive.")


@dataclass(frozen=True)
class MonodispersePrimitiveParticleDistribution(PrimitiveParticleDistribution):
    n: float
    θ: float
    type_id = MONODISPERSE

    def __post_init__(self):  # ParticleDistributions.jl:126-129
        if self.n < 0 or self.θ <= 0:
            raise ValueError("n needs to be nonnegative. θ needs to be positive.")


@dataclass(frozen=True)
class LognormalPrimitiveParticleDistribution(PrimitiveParticleDistribution):
    n: float
    μ: float
    σ: float
    type_id = LOGNORMAL

    def __post_init__(self):  # ParticleDistributions.jl:153-156
        if self.n < 0 or self.σ <= 0:
            raise ValueError("n needs to be nonnegative. σ needs to be positive.")


def nparams(dist):
    """ParticleDistributions.jl:425-427."""
    return NPARAMS[dist.type_id]


def get_moments(pdist):
    """First nparams moments, closed form (ParticleDistributions.jl:293-315); used to build initial conditions."""
    import math

    if isinstance(pdist, GammaPrimitiveParticleDistribution):
        return [pdist.n, pdist.n * pdist.k * pdist.θ, pdist.n * pdist.k * (pdist.k + 1) * pdist.θ**2]
    if isinstance(pdist, LognormalPrimitiveParticleDistribution):
        return [pdist.n, pdist.n * math.exp(pdist.μ + pdist.σ**2 / 2), pdist.n * math.exp(2.0 * pdist.μ + 2.0 * pdist.σ**2)]
    return [pdist.n, pdist.n * pdist.θ]


def pack_params(pdists_batch):
    """[(n, theta, k) arrays per mode] -> (3N, n_parcels) host array in the device plane order."""
    rows = []
    for d in pdists_batch:
        n = np.asarray(d[0], dtype=np.float64)
        th = np.asarray(d[1], dtype=np.float64)
        k = np.asarray(d[2], dtype=np.float64) if len(d) > 2 else np.ones_like(n)
        rows += [n, th, np.broadcast_to(k, n.shape)]
    return np.ascontiguousarray(np.stack([np.atleast_1d(r) for r in rows]))


def update_dist_from_moments(plan, mom, params_out=None, stream=None):
    """Batched update_dist_from_moments (ParticleDistributions.jl:456-476, 512-523) on the GPU.

    mom: (nmom, n) device array, physical units -> (3N, n) device array of (n, theta, k), normalised units."""
    ptr, planes, n, ld = as_device(mom)
    if planes != plan.nmom:
        raise TypeError(f"no method matching update_dist_from_moments: expected {plan.nmom} moments, got {planes}")
    out = params_out if params_out is not None else DeviceArray(3 * plan.N, n)
    optr, oplanes, on, old = as_device(out)
    if old != ld or oplanes != 3 * plan.N:
        raise ValueError("params_out must be (3N, n) with the same leading dimension as mom")
    _lib.check(_lib.lib().cloudy_update_dist_from_moments(plan.handle, n, ld, ptr, optr, stream))
    return out


def check_moment_consistency(m):
    """check_moment_consistency(m) (ParticleDistributions.jl:437-449) for ONE tuple of moments, on the host: raises ValueError where
    the reference raises (a negative moment; a negative even-ordered central moment), returns None otherwise (IEEE division as in
    Julia: (0, 1, 2) gives Inf - Inf = NaN, which is not < 0).  The batched form is closure_stats."""
    import math

    import numpy as np

    m = np.asarray(m, dtype=np.float64)
    if (m < 0.0).any():
        raise ValueError("all moments need to be nonnegative.")
    with np.errstate(all="ignore"):
        for order in range(2, len(m), 2):
            cm = np.float64(0.0)
            for i in range(order + 1):
                cm = cm + (math.comb(order, i) * (-1) ** i) * (m[1] / m[0]) ** i * (m[order - i] / m[0])
            if cm < 0.0:
                raise ValueError("order central moment needs to be nonnegative.")
    return None


CLOSURE_STATS_FIELDS = ("fallback", "k_min", "k_max", "inconsistent")


def closure_stats(plan, mom, stream=None):
    """Batched validity diagnostic (cloudy_closure_stats): per mode, how many parcels of `mom` (nmom, n; device; physical units)
    took the fallback distribution (0, 1, 1), have their shape at the lower / upper clamp, or fail check_moment_consistency.
    -> (N, 4) integer array, columns CLOSURE_STATS_FIELDS.  The reference clamps silently and checks one tuple at a time
    (ParticleDistributions.jl:437-449, 456-541)."""
    import ctypes as C

    import numpy as np

    ptr, planes, n, ld = as_device(mom)
    if planes != plan.nmom:
        raise TypeError(f"closure_stats: expected {plan.nmom} moments, got {planes}")
    buf = (C.c_uint64 * (4 * plan.N))()
    _lib.check(_lib.lib().cloudy_closure_stats(plan.handle, n, ld, ptr, buf, stream))
    return np.array(buf[:], dtype=np.int64).reshape(plan.N, 4)


def compute_thresholds(plan, params, out=None, stream=None):
    """Per-parcel thresholds used by the S terms (ParticleDistributions.jl:734-761); (N, n) device array."""
    ptr, planes, n, ld = as_device(params)
    o = out if out is not None else DeviceArray(plan.N, n)
    optr = as_device(o)[0]
    _lib.check(_lib.lib().cloudy_compute_thresholds(plan.handle, n, ld, ptr, optr, stream))
    return o


def get_standard_N_q(plan, mom, size_cutoff=1e-6, out=None, stream=None):
    """get_standard_N_q(pdists, size_cutoff) (ParticleDistributions.jl:634-687), batched: mom (nmom, n) device,
    physical units -> (4, n) device array of (N_liq, N_rai, M_liq, M_rai)."""
    ptr, planes, n, ld = as_device(mom)
    o = out if out is not None else DeviceArray(4, n, plane_dtype(plan))
    if as_device(o)[3] != ld:
        raise ValueError("out must have the same leading dimension as mom")
    _lib.check(_lib.lib().cloudy_standard_N_q(plan.handle, n, ld, ptr, float(size_cutoff), as_device(o)[0], stream))
    return o
