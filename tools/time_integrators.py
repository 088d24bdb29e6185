#!/usr/bin/env python3
"""Timing of the fused integrators (cloudy_ssprk33_steps with thresholds, cloudy_rainshaft_ssprk33_steps) for one
libcloudy_hip.so build (CLOUDY_HIP_LIB selects it).  usage: python tools/time_integrators.py [--cells N]"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=4_000_000)
ap.add_argument("--nz", type=int, default=20)
a = ap.parse_args()
pkg = load_package()
L = pkg.lib()
n = (a.cells // a.nz) * a.nz
res = {}


def timed(fn, reps=3):
    fn()
    pkg._lib.check(L.cloudy_stream_synchronize(None))
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        pkg._lib.check(L.cloudy_stream_synchronize(None))
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for name in ("cfg3a", "cfg3b", "cfg4"):
    wl = bench.make_workload(name, n if name != "cfg4" else n // 4, seed=7)
    n_w = wl["mom"].shape[1]
    vel = ((50.0, 1.0 / 6),)
    plan = wl["coal_data"].plan(wl["dist_types"], vel=vel)
    u = pkg.DeviceArray.from_numpy(wl["mom"])
    out = pkg.DeviceArray.zeros(*wl["mom"].shape)
    steps = 2
    ms = timed(lambda: pkg._lib.check(L.cloudy_ssprk33_steps(plan.handle, n_w, n_w, u.ptr, out.ptr, 1e-3, steps, None)))
    res[f"box_ssprk33_{name}_ms_per_rhs_eval_1e6"] = ms / (3 * steps) / (n_w / 1e6)
    if name == "cfg4":
        continue
    ms = timed(lambda: pkg._lib.check(L.cloudy_rainshaft_ssprk33_steps(plan.handle, a.nz, n // a.nz, n, u.ptr, out.ptr,
                                                                      150.0, 1e-3, steps, None)))
    res[f"rainshaft_ssprk33_{name}_ms_per_rhs_eval_1e6"] = ms / (3 * steps) / (n / 1e6)
    # the unfused pieces for reference: coalescence rhs + sedimentation flux launches
    cs, sf = pkg.DeviceArray.zeros(*wl["mom"].shape), pkg.DeviceArray.zeros(*wl["mom"].shape)
    ms = timed(lambda: pkg._lib.check(L.cloudy_rainshaft_rhs(plan.handle, a.nz, n // a.nz, n, u.ptr, 150.0, sf.ptr,
                                                             cs.ptr, None)))
    res[f"rainshaft_rhs_{name}_ms_per_1e6"] = ms / (n / 1e6)
print(json.dumps(res))
