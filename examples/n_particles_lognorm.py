#!/usr/bin/env python3
"""test/examples/Numerical/n_particles_lognorm.jl of the reference, line for line, through the Python mirror of the Cloudy API
(cloudy.jl_amd) -- two Lognormal modes, LinearKernelFunction(5.0), NumericalCoalStyle (the converged mode of the MI355X
operator), SSPRK33 with dt = 1 s to T_end = 50 s -- for ONE box (the example) and for a batch of boxes around it.

    python examples/n_particles_lognorm.py [n_boxes]

Needs a GPU (the package has no CPU path).  The Julia original prints `sol.u`; this prints the final moments."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

cl = load_package()

T_end, coalescence_coeff, dt = 50.0, 5.0, 1.0          # n_particles_lognorm.jl:12-14
particle_number, mass_scale = [1e7, 1e5], [1e-10, 1e-9]  # :17-19
pdists = tuple(cl.LognormalPrimitiveParticleDistribution(n, math.log(m), math.log(2.0)) for n, m in zip(particle_number, mass_scale))
dist_moments = np.concatenate([cl.get_moments(d) for d in pdists])   # :25
kernel = cl.LinearKernelFunction(coalescence_coeff)                 # :29
NProgMoms = tuple(cl.nparams(d) for d in pdists)
norms = (1e6, 1e-9)                                                 # :33
kernel_n = cl.get_normalized_kernel_func(kernel, norms)
par = cl.ODEParameters(pdists, None, NProgMoms, norms, kernel_func=kernel_n)   # the ODE_parameters NamedTuple (:36-37)


def solve(u0):
    """solve(ODEProblem(make_box_model_rhs(NumericalCoalStyle()), u0, (0, T_end), par), SSPRK33(), dt = dt), final state"""
    u = cl.DeviceArray.from_numpy(np.ascontiguousarray(u0))
    cl.solve_ssprk33(par, u, dt, int(round(T_end / dt)), coal_type=cl.NumericalCoalStyle())
    return u.to_numpy()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    u0 = np.repeat(dist_moments[:, None], n, axis=1)
    if n > 1:   # boxes around the example: numbers and masses within 10 %
        rng = np.random.default_rng(1)
        u0 = u0 * rng.uniform(0.9, 1.1, (1, n))
    uT = solve(u0)
    print("moments at t = 0    :", u0[:, 0])
    print(f"moments at t = {T_end:g} s:", uT[:, 0])
    mass0, massT = u0[1] + u0[4], uT[1] + uT[4]
    print(f"total mass conserved to {np.max(np.abs(massT - mass0) / mass0):.1e}; total number {u0[0, 0] + u0[3, 0]:.4g} -> {uT[0, 0] + uT[3, 0]:.4g}")
