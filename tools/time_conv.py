#!/usr/bin/env python3
"""Times the CONVERGED-mode NumericalCoalStyle kernel of the bench's cfg4q batch under occupancy overrides
(CLOUDY_HIP_JIT_QUAD_WAVES); run on the GPU box: python tools/time_conv.py [n_parcels] [waves ...]."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
mom = bench.synth_moments(3, n, bench.SEED)
m, dm = pkg.DeviceArray.from_numpy(mom), pkg.DeviceArray.zeros(9, n)
kfn = pkg.get_normalized_kernel_func(pkg.HydrodynamicKernelFunction(1e2 * np.pi), bench.NORMS)
for waves in (sys.argv[2:] or ["", "2", "3", "4"]):
    os.environ.pop("CLOUDY_HIP_JIT_QUAD_WAVES", None)
    if waves:
        os.environ["CLOUDY_HIP_JIT_QUAD_WAVES"] = waves
    plan = pkg.NumericalPlan([1, 1, 1], kfn, bench.NORMS, 8, specialize=1, quad_mode=pkg.QUAD_CONVERGED)
    for _ in range(2):
        pkg._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None))
    ms = C.c_float()
    pkg._lib.check(L.cloudy_time_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None, 3, C.byref(ms)))
    print(f"converged N=3 hydro waves={waves or 'auto'}: {ms.value:.3f} ms per {n} parcels = {n / ms.value * 1e3:.3e} parcel-RHS/s",
          flush=True)
