#!/usr/bin/env python3
"""Divergence probe of the converged-mode kernel: batches made of one parcel replicated, of another, and of the two
interleaved lane by lane -- T(A), T(B), T(AB).  T(AB) ~ max(T(A), T(B)): lanes share the instruction stream of the adaptive
walk; T(AB) ~ T(A) + T(B): they are serialised.  Run on the GPU box: python tools/conv_divergence_probe.py"""
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
n = 2_000_000
base = bench.synth_moments(3, 4096, bench.SEED)
kfn = pkg.get_normalized_kernel_func(pkg.HydrodynamicKernelFunction(1e2 * np.pi), bench.NORMS)
plan = pkg.NumericalPlan([1, 1, 1], kfn, bench.NORMS, 8, specialize=1, quad_mode=pkg.QUAD_CONVERGED)


def timed(mom):
    m, dm = pkg.DeviceArray.from_numpy(np.ascontiguousarray(mom)), pkg.DeviceArray.zeros(9, mom.shape[1])
    for _ in range(2):
        pkg._lib.check(L.cloudy_coal_rhs(plan.handle, mom.shape[1], mom.shape[1], m.ptr, dm.ptr, None))
    ms = C.c_float()
    pkg._lib.check(L.cloudy_time_coal_rhs(plan.handle, mom.shape[1], mom.shape[1], m.ptr, dm.ptr, None, 3, C.byref(ms)))
    return ms.value


for ia, ib in ((0, 1), (2, 3), (4, 5), (10, 700)):
    A = np.repeat(base[:, ia:ia + 1], n, axis=1)
    B = np.repeat(base[:, ib:ib + 1], n, axis=1)
    AB = A.copy()
    AB[:, 1::2] = B[:, 1::2]
    ta, tb, tab = timed(A), timed(B), timed(AB)
    print(f"parcels {ia}, {ib}: T(A) {ta:.2f} ms  T(B) {tb:.2f} ms  T(AB) {tab:.2f} ms   max {max(ta, tb):.2f}  sum {ta + tb:.2f}", flush=True)
full = bench.synth_moments(3, n, bench.SEED)
print(f"the mixed batch: {timed(full):.2f} ms;  sorted by the first mode's mean size: "
      f"{timed(full[:, np.argsort(full[1] / full[0])]):.2f} ms", flush=True)
