#!/usr/bin/env python3
"""The converged-mode kernel of the cfg4q batch (3 Gamma modes) under each kernel-function family, and with a Lognormal mode:
python tools/time_conv_kernels.py [n_parcels]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from __graft_entry__ import load_package

pkg = load_package(); L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
mom = bench.synth_moments(3, n, bench.SEED)
m, dm = pkg.DeviceArray.from_numpy(mom), pkg.DeviceArray.zeros(9, n)
kfs = {"constant": pkg.ConstantKernelFunction(1e-4), "linear": pkg.LinearKernelFunction(5.0),
       "hydrodynamic": pkg.HydrodynamicKernelFunction(1e2 * np.pi), "long": pkg.LongKernelFunction(5.236e-10, 9.44e9, 5.78)}
for dist, tag in (([1, 1, 1], "3 Gamma"), ([1, 3, 1], "Gamma, Lognormal, Gamma")):
    for name, kf in kfs.items():
        if 3 in dist and name in ("constant", "linear"):
            continue
        nn = n if 3 not in dist else n // 40
        plan = pkg.NumericalPlan(dist, pkg.get_normalized_kernel_func(kf, bench.NORMS), bench.NORMS, 8, specialize=1, quad_mode=pkg.QUAD_CONVERGED)
        ms = bench._sustained_ms(pkg, lambda: pkg._lib.check(L.cloudy_coal_rhs(plan.handle, nn, n, m.ptr, dm.ptr, None)), min_reps=3, max_reps=50)
        print(f"converged, {tag}, {name}: {ms:.3f} ms per {nn} parcels = {nn / ms * 1e3:.3e} parcel-RHS/s", flush=True)
