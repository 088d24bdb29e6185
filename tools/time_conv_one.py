#!/usr/bin/env python3
"""One converged-mode NumericalCoalStyle plan on the cfg4q batch, timed by HIP events (for rocprofv3 --pmc passes of ONE kernel):
python tools/time_conv_one.py <constant|linear|hydrodynamic|long> <dist,dist,...> [n_parcels] [reps]   (1 Gamma, 0 Exponential, 3 Lognormal)"""
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
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
kf = {"constant": pkg.ConstantKernelFunction(1e-4), "linear": pkg.LinearKernelFunction(5.0),
      "hydrodynamic": pkg.HydrodynamicKernelFunction(1e2 * np.pi), "long": pkg.LongKernelFunction(5.236e-10, 9.44e9, 5.78)}[kname]
N = len(dists)
mom = bench.synth_moments(N, n, bench.SEED)
m, dm = pkg.DeviceArray.from_numpy(mom), pkg.DeviceArray.zeros(3 * N, n)
plan = pkg.NumericalPlan(dists, pkg.get_normalized_kernel_func(kf, bench.NORMS), bench.NORMS, 8, specialize=1, quad_mode=pkg.QUAD_CONVERGED)
ms = bench._sustained_ms(pkg, lambda: pkg._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None)), min_reps=reps, max_reps=max(reps, 20))
print(f"converged {kname} {dists}: {ms:.3f} ms per {n} parcels = {n / ms * 1e3:.3e} parcel-RHS/s", flush=True)
