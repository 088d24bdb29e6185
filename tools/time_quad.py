#!/usr/bin/env python3
"""Times the NumericalCoalStyle (fixed Gauss rule) kernel of the bench's cfg4q workload under occupancy overrides
(CLOUDY_HIP_JIT_QUAD_WAVES) and rule orders; run on the GPU box: python tools/time_quad.py [n_parcels]."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from __graft_entry__ import load_package


def main():
    pkg = load_package()
    L = pkg.lib()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    quick = len(sys.argv) > 2 and sys.argv[2] == "quick"   # first configuration, current occupancy setting only
    cfgs = ((3, pkg.HydrodynamicKernelFunction(1e2 * np.pi), 10), (3, pkg.LongKernelFunction(5.236e-10, 9.44e9, 5.78), 10),
                      (2, pkg.HydrodynamicKernelFunction(1e2 * np.pi), 10), (1, pkg.HydrodynamicKernelFunction(1e2 * np.pi), 10),
                      (3, pkg.HydrodynamicKernelFunction(1e2 * np.pi), 6), (3, pkg.HydrodynamicKernelFunction(1e2 * np.pi), 16))
    for N, kf, nq in cfgs[:1] if quick else cfgs:
        mom = bench.synth_moments(N, n, bench.SEED)
        m, dm = pkg.DeviceArray.from_numpy(mom), pkg.DeviceArray.zeros(3 * N, n)
        kfn = pkg.get_normalized_kernel_func(kf, bench.NORMS)
        for waves in ((os.environ.get("CLOUDY_HIP_JIT_QUAD_WAVES", ""),) if quick else ("", "2", "3", "4")):
            os.environ.pop("CLOUDY_HIP_JIT_QUAD_WAVES", None)
            os.environ.pop("CLOUDY_HIP_JIT_QUAD_LICM", None)
            if waves.rstrip("L0"):
                os.environ["CLOUDY_HIP_JIT_QUAD_WAVES"] = waves.rstrip("L0")
            if waves.endswith("L0"):
                os.environ["CLOUDY_HIP_JIT_QUAD_LICM"] = "0"
            try:
                plan = pkg.NumericalPlan([1] * N, kfn, bench.NORMS, nq, specialize=1, quad_mode=pkg.QUAD_FIXED)
            except pkg.CloudyError as e:
                print(f"N={N} {type(kf).__name__} nq={nq} waves={waves or 'auto'}: {e}")
                continue
            for _ in range(2):
                pkg._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None))
            ms = C.c_float()
            pkg._lib.check(L.cloudy_time_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None, 3, C.byref(ms)))
            print(f"N={N} {type(kf).__name__} nq={nq} waves={waves or 'auto'}: {ms.value:.3f} ms per {n} parcels = "
                  f"{n / ms.value * 1e3:.3e} parcel-RHS/s", flush=True)
        if quick:
            continue
        aot = pkg.NumericalPlan([1] * N, kfn, bench.NORMS, nq, specialize=-1, quad_mode=pkg.QUAD_FIXED)
        pkg._lib.check(L.cloudy_coal_rhs(aot.handle, n, n, m.ptr, dm.ptr, None))
        ms = C.c_float()
        pkg._lib.check(L.cloudy_time_coal_rhs(aot.handle, n, n, m.ptr, dm.ptr, None, 2, C.byref(ms)))
        print(f"N={N} {type(kf).__name__} nq={nq} ahead-of-time: {ms.value:.3f} ms", flush=True)
        del m, dm


if __name__ == "__main__":
    main()
