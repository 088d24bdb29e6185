#!/usr/bin/env python3
"""Sustained time of the fused all-Inf integrators (cfg3a SSPRK33 x 4 steps, cfg2 Tsit5 x 4 steps on 1e7 parcels) under the
occupancy override CLOUDY_HIP_JIT_INT_WAVES: python tools/time_allinf_integrators.py"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from __graft_entry__ import load_package
pkg = load_package(); L = pkg.lib()
n = 10_000_000
out = []
for name, fn, steps in (("cfg3a", "cloudy_ssprk33_steps", 4), ("cfg2", "cloudy_tsit5_steps", 4)):
    wl = bench.make_workload(name, n)
    plan = wl["coal_data"].plan(wl["dist_types"])
    u, o = pkg.DeviceArray.from_numpy(wl["mom"]), pkg.DeviceArray.zeros(plan.nmom, n)
    f = getattr(L, fn)
    ms = bench._sustained_ms(pkg, lambda: pkg._lib.check(f(plan.handle, n, n, u.ptr, o.ptr, C.c_double(1e-3), steps, None)))
    out.append(f"{name} {fn} {ms:.4f} ms")
print("INT_WAVES", os.environ.get("CLOUDY_HIP_JIT_INT_WAVES", "-"), "|", " | ".join(out), flush=True)
