#!/usr/bin/env python3
"""The fused SSPRK33 integrator of a converged-mode NumericalCoalStyle plan (cloudy_ssprk33_steps: 3 evaluations per step, state in
registers) against three cloudy_coal_rhs launches of the same plan: python tools/time_conv_fused.py <kernel> <dists> [n_parcels]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from __graft_entry__ import load_package

pkg = load_package()
L = pkg.lib()
kname, dists = sys.argv[1], [int(x) for x in sys.argv[2].split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4_000_000
kf = {"constant": pkg.ConstantKernelFunction(1e-4), "linear": pkg.LinearKernelFunction(5.0),
      "hydrodynamic": pkg.HydrodynamicKernelFunction(1e2 * np.pi), "long": pkg.LongKernelFunction(5.236e-10, 9.44e9, 5.78)}[kname]
N = len(dists)
mom = bench.synth_moments(N, n, bench.SEED)
m, dm = pkg.DeviceArray.from_numpy(mom), pkg.DeviceArray.zeros(3 * N, n)
plan = pkg.NumericalPlan(dists, pkg.get_normalized_kernel_func(kf, bench.NORMS), bench.NORMS, 8, specialize=1, quad_mode=pkg.QUAD_CONVERGED)
ms_rhs = bench._sustained_ms(pkg, lambda: pkg._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None)), min_reps=3, max_reps=10)
ms_int = bench._sustained_ms(pkg, lambda: pkg._lib.check(L.cloudy_ssprk33_steps(plan.handle, n, n, m.ptr, dm.ptr, C.c_double(1e-3), 1, None)), min_reps=3, max_reps=10)
print(f"converged {kname} {dists}, {n} parcels: cloudy_coal_rhs {ms_rhs:.3f} ms; cloudy_ssprk33_steps (1 step = 3 evaluations) {ms_int:.3f} ms "
      f"= {ms_int / (3 * ms_rhs):.2f} x three launches", flush=True)
