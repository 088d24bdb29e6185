#!/usr/bin/env python3
"""Converged mode on the DEVICE (the default operator, second call = ranked by cost hints) against the same-rule CPU oracle run at
tolerance 1e-13, on random multi-scale Gamma mixtures -- shapes 1e-3 ... 10, scales 3.5 decades and numbers 3 decades apart -- for
N = 2, 3 and the four kernel functions:  python tools/conv_wild_device.py [mixtures_per_family] > profiles/r05_converged_wild_device.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from __graft_entry__ import load_package
from oracle import cloudy_oracle as O

pkg = load_package()
L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(2026)
NORMS = (1.0, 1.0)
kfs = {0: ("constant", (0.7,)), 1: ("linear", (5e-3,)), 2: ("hydrodynamic", (0.3,)), 3: ("long", None)}
mk = {0: pkg.ConstantKernelFunction, 1: pkg.LinearKernelFunction, 2: pkg.HydrodynamicKernelFunction, 3: pkg.LongKernelFunction}
worst_all = 0.0
for N in (2, 3):
    for kind in range(4):
        prm = kfs[kind][1] or (float(10 ** rng.uniform(-1, 1)), 9.0, 5.0)
        kf, okf = mk[kind](*prm), O.kernel_func(kind, *prm)
        mom = np.zeros((3 * N, n))
        for i in range(N):
            c = rng.integers(0, 4, n)
            k = np.select([c == 0, c == 1, c == 2, c == 3], [rng.uniform(0.05, 1.0, n), rng.uniform(1, 10, n), rng.uniform(1, 10, n), 10 ** rng.uniform(-3, -1, n)])
            nn, th = 10 ** rng.uniform(-1, 2, n), 10 ** rng.uniform(-2, 1.5, n)
            mom[3 * i], mom[3 * i + 1], mom[3 * i + 2] = nn, nn * k * th, nn * k * (k + 1) * th * th
        plan = pkg.NumericalPlan([1] * N, pkg.get_normalized_kernel_func(kf, NORMS), NORMS, 8, specialize=1, quad_mode=pkg.QUAD_CONVERGED)
        m, dm = pkg.DeviceArray.from_numpy(mom), pkg.DeviceArray.zeros(3 * N, n)
        for _ in range(2):   # the second call is ranked by the first call's cost hints
            pkg._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None))
        pkg._lib.check(L.cloudy_stream_synchronize(None))
        got = dm.to_numpy()
        op = O.make_params([O.GAMMA] * N, np.zeros((1, 1)), (np.inf,) * N, norms=NORMS)
        okn = O.get_normalized_kernel_func(okf, NORMS)
        ref, sc = O.rhs_coal_numerical_converged_batch(op, okn, 8, mom, tol=1e-13, with_scale=True)
        same = O.rhs_coal_numerical_converged_batch(op, okn, 8, mom)
        ok = np.isfinite(ref) & (sc > 0)
        e_ref = np.abs(got - ref)[ok] / sc[ok]
        e_same = np.abs(got - same)[ok] / sc[ok]
        worst_all = max(worst_all, float(e_ref.max()))
        print(f"N={N} {kfs[kind][0]:12s} {prm}: {n} mixtures; max |hip - oracle(1e-13)| / scale {e_ref.max():.2e} (99.9 % {np.percentile(e_ref, 99.9):.1e}); "
              f"max |hip - oracle(same tolerance)| / scale {e_same.max():.2e}", flush=True)
print(f"worst over all families: {worst_all:.2e} of scale (guaranteed bound of the mode: 1e-8)")
