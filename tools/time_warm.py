#!/usr/bin/env python3
"""Short kernels timed cold (the bench's old variant protocol: one call, then 5 timed) and warm (>= 150 ms of back-to-back
launches first, then 40 timed): after idling the GPU needs tens of ms to reach its sustained clock (tools/launch_ramp_timing.py),
so a 1-ms kernel timed over 5 launches right after a host-side copy is priced at a lower clock than it sustains.
usage (GPU box): python tools/time_warm.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
L = pkg.lib()


def timed(fn, reps):
    with bench._EventTimer(pkg) as tm:
        for _ in range(reps):
            fn()
    return tm.ms / reps


def both(name, fn):
    pkg._lib.check(L.cloudy_stream_synchronize(None))
    time.sleep(0.3)                      # the GPU idles, as it does while the bench prepares a variant on the host
    fn()
    pkg._lib.check(L.cloudy_stream_synchronize(None))
    cold = timed(fn, 5)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        for _ in range(10):
            fn()
        pkg._lib.check(L.cloudy_stream_synchronize(None))
    warm = timed(fn, 40)
    print(f"{name:44s} cold (1 + 5 launches) {cold:8.4f} ms   warm (150 ms + 40 launches) {warm:8.4f} ms", flush=True)


n = 10_000_000
wl = bench.make_workload("cfg3a", n, seed=bench.SEED)
plan = wl["coal_data"].plan(wl["dist_types"])
u = pkg.DeviceArray.from_numpy(wl["mom"])
m, dm = pkg.DeviceArray.from_numpy(wl["mom"]), pkg.DeviceArray.zeros(*wl["mom"].shape)
both("cfg3a cloudy_coal_rhs", lambda: pkg._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None)))
both("cfg3a fused SSPRK33, 4 steps (12 evals)",
     lambda: pkg._lib.check(L.cloudy_ssprk33_steps(plan.handle, n, n, u.ptr, u.ptr, C.c_double(1e-3), 4, None)))
wl = bench.make_workload("cfg3b", n, seed=bench.SEED)
plan = wl["coal_data"].plan(wl["dist_types"])
m = pkg.DeviceArray.from_numpy(wl["mom"])
both("cfg3b cloudy_coal_rhs", lambda: pkg._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None)))
n4 = 2_500_000
wl = bench.make_workload("moving4", n4, seed=bench.SEED)
plan = wl["coal_data"].plan(wl["dist_types"])
m4, d4 = pkg.DeviceArray.from_numpy(wl["mom"]), pkg.DeviceArray.zeros(*wl["mom"].shape)
both("moving4 cloudy_coal_rhs", lambda: pkg._lib.check(L.cloudy_coal_rhs(plan.handle, n4, n4, m4.ptr, d4.ptr, None)))
