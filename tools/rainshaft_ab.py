#!/usr/bin/env python3
"""A/B of the fused column integrator: time per call on the bench batch (500000 columns x 20 cells, cfg3b physics) and
SHA-256 of the complete output for several column heights, for the library CLOUDY_HIP_LIB names (default: the in-tree
one).  Two builds agree bit for bit iff their digests agree.
usage: [CLOUDY_HIP_LIB=tools/variants/libcloudy_hip_base.so] python tools/rainshaft_ab.py"""
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from __graft_entry__ import load_package

pkg = load_package()
L = pkg.lib()


def run(plan, nz, ncol, u, out, steps):
    pkg._lib.check(L.cloudy_rainshaft_ssprk33_steps(plan.handle, nz, ncol, nz * ncol, u.ptr, out.ptr, C.c_double(150.0),
                                                    C.c_double(1e-3), steps, None))


def sweep():
    """time per call for several column heights at ~5e6 cells (which workgroup size serves which height best)"""
    tag = os.environ.get("CLOUDY_HIP_RS_BLOCK", "default")
    for nz in (5, 7, 16, 20, 33, 64, 100, 128, 200, 256):
        ncol = 5000000 // nz
        wl = bench.make_workload("cfg3b", nz * ncol, seed=7)
        plan = wl["coal_data"].plan(wl["dist_types"], vel=((50.0, 1.0 / 6),))
        u = pkg.DeviceArray.from_numpy(wl["mom"])
        out = pkg.DeviceArray.zeros(*wl["mom"].shape)
        for _ in range(8):
            run(plan, nz, ncol, u, out, 2)
        pkg._lib.check(L.cloudy_stream_synchronize(None))
        t0 = time.perf_counter()
        for _ in range(10):
            run(plan, nz, ncol, u, out, 2)
        pkg._lib.check(L.cloudy_stream_synchronize(None))
        print(f"block {tag} nz={nz}: {(time.perf_counter() - t0) / 10 * 1e3 / (nz * ncol) * 1e7:.3f} ms per 1e7 cells", flush=True)


def main():
    if os.environ.get("RS_AB_SWEEP"):
        return sweep()
    tag = os.path.basename(os.environ.get("CLOUDY_HIP_LIB", "in-tree"))
    for name, nz, ncol in (("cfg3b", 20, 500000), ("cfg3b", 7, 30011), ("cfg3b", 256, 513), ("cfg3b", 300, 257), ("cfg4", 20, 40000), ("cfg4", 20, 2000),
                           ("cfg3a", 20, 50000)):
        n = nz * ncol
        wl = bench.make_workload(name, n, seed=7)
        plan = wl["coal_data"].plan(wl["dist_types"], vel=((50.0, 1.0 / 6),))
        u = pkg.DeviceArray.from_numpy(wl["mom"])
        out = pkg.DeviceArray.zeros(*wl["mom"].shape)
        run(plan, nz, ncol, u, out, 2)
        pkg._lib.check(L.cloudy_stream_synchronize(None))
        res = np.ascontiguousarray(out.to_numpy())
        res[np.isnan(res)] = np.nan   # (one NaN pattern: the sign of a NaN is not a result)
        h = hashlib.sha256(res.tobytes()).hexdigest()[:16]
        if os.environ.get("RS_AB_DUMP") and n <= 50000:
            np.save(os.path.join(os.environ["RS_AB_DUMP"], f"rsab_{tag}_{name}_{nz}.npy"), res)
        ms = float("nan")
        if ncol >= 500000:
            for _ in range(12):
                run(plan, nz, ncol, u, out, 2)
            pkg._lib.check(L.cloudy_stream_synchronize(None))
            t0 = time.perf_counter()
            for _ in range(10):
                run(plan, nz, ncol, u, out, 2)
            pkg._lib.check(L.cloudy_stream_synchronize(None))
            ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"{tag} {name} nz={nz} ncol={ncol}: sha {h}  ms/call {ms:.3f}", flush=True)


if __name__ == "__main__":
    main()
