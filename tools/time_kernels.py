#!/usr/bin/env python3
"""Kernel time of cloudy_coal_rhs for named bench workloads (HIP events on the launch stream), one line per workload.
usage: python tools/time_kernels.py [--reps R] cfg3b cfg4 moving4 ...      (CLOUDY_HIP_LIB / CLOUDY_HIP_JIT respected)"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from __graft_entry__ import load_package


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--parcels", type=int, default=0)
    ap.add_argument("--degenerate", type=float, default=0.01, help="fraction of degenerate parcels in the synthetic batch")
    ap.add_argument("--error", action="store_true", help="also the error against the oracle on 20000 parcels")
    ap.add_argument("workloads", nargs="+")
    a = ap.parse_args()
    pkg = load_package()
    L = pkg.lib()
    out = []
    if a.degenerate != 0.01:
        import functools

        bench.synth_moments = functools.partial(bench.synth_moments, degenerate_frac=a.degenerate)
    for name in a.workloads:
        n = a.parcels or bench.workload_spec(name)["default_parcels"]
        wl = bench.make_workload(name, n)
        plan = wl["coal_data"].plan(wl["dist_types"])
        m, dm = pkg.DeviceArray.from_numpy(wl["mom"]), pkg.DeviceArray.zeros(plan.nmom, n)
        ms_v = bench._event_ms(pkg, plan, m, dm, a.reps)   # sustained clock: >= 150 ms of launches first
        err = ""
        if a.error:   # max |hip - oracle| / scale on the first 20000 parcels
            import numpy as np
            from oracle import cloudy_oracle as O
            ns = min(n, 20000)
            want, scale = O.rhs_coal_batch(bench.oracle_params(name), np.ascontiguousarray(wl["mom"][:, :ns]), with_scale=True)
            got = dm.columns_to_numpy(ns)
            ok = np.isfinite(want) & (scale > 0)
            err = f", err {float((np.abs(got - want)[ok] / scale[ok]).max()):.1e}"
        out.append(f"{name} {ms_v:.3f} ms ({n / ms_v * 1e3:.3e}/s, jit={int(plan.specialized)}{err})")
        del m, dm
    print(os.environ.get("CLOUDY_HIP_LIB", "default").split("/")[-1], "|", " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
