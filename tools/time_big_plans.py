#!/usr/bin/env python3
"""Kernel time of cloudy_coal_rhs for plans beyond the ahead-of-time families (plan-time compiled kernels only):
python tools/time_big_plans.py [n_parcels]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bench
import fuzz_parity as F
from __graft_entry__ import load_package

pkg = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
INF = float("inf")
rng = np.random.default_rng(1)
cases = [("n6p2 no thresholds", [1] * 6, 2, (INF,) * 6, False), ("n8p2 no thresholds", [1] * 8, 2, (INF,) * 8, False),
         ("n2p8 no thresholds", [1, 1], 8, (INF, INF), False), ("n8p8 no thresholds", [1] * 8, 8, (INF,) * 8, False),
         ("n5p3 four thresholds", [1] * 5, 3, (1e-10, 1e-9, 1e-8, 1e-7, INF), False),
         ("n8p2 six thresholds", [1] * 8, 2, (1e-11, 1e-10, INF, 1e-8, 1e-7, 1e-6, 1e-5, INF), False),
         ("n6p2 moving", [1] * 6, 2, (0.9, 0.95, 0.99, 0.9, 0.99, 1.0), True)]
for name, dist, P, thr, moving in cases:
    N = len(dist)
    kc = np.zeros((N, N, P, P))
    for j in range(N):
        for k in range(j, N):
            c = rng.uniform(0.1, 1.0, (P, P)) * (rng.random((P, P)) < 0.5)
            c = np.triu(c) + np.triu(c, 1).T
            kc[j, k] = kc[k, j] = c * 1e-3 * (1e9 ** np.add.outer(np.arange(P), np.arange(P)))
    kernels = tuple(tuple(pkg.CoalescenceTensor(kc[j, k]) for k in range(N)) for j in range(N))
    cd = pkg.CoalescenceData(kernels, (3,) * N, thr, bench.NORMS, pkg.MovingThreshold() if moving else pkg.FixedThreshold())
    plan = cd.plan(dist)
    nn = n if not (moving or any(np.isfinite(thr))) else n // 4
    mom = F.moments_for(dist, nn, 5) if N > 4 else bench.synth_moments(N, nn, 5)
    m, dm = pkg.DeviceArray.from_numpy(mom), pkg.DeviceArray.zeros(plan.nmom, nn)
    ms = bench._event_ms(pkg, plan, m, dm, 5)
    gb = 2 * 8 * plan.nmom * nn / ms / 1e6
    print(f"{name:24s} {nn} parcels: {ms:.3f} ms = {nn / ms * 1e3:.3e} parcel-RHS/s, {gb:.0f} GB/s of moment traffic", flush=True)
