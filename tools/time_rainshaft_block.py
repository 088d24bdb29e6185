import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from __graft_entry__ import load_package
pkg = load_package(); L = pkg.lib()
nz, ncol = 20, 500000
n = nz * ncol
wl = bench.make_workload("cfg3b", n, seed=7)
plan = wl["coal_data"].plan(wl["dist_types"], vel=((50.0, 1.0 / 6),))
u = pkg.DeviceArray.from_numpy(wl["mom"]); out = pkg.DeviceArray.zeros(*wl["mom"].shape)
import time
def run():
    pkg._lib.check(L.cloudy_rainshaft_ssprk33_steps(plan.handle, nz, ncol, n, u.ptr, out.ptr, C.c_double(150.0), C.c_double(1e-3), 2, None))
for _ in range(12): run()
pkg._lib.check(L.cloudy_stream_synchronize(None))
t0 = time.perf_counter()
for _ in range(10): run()
pkg._lib.check(L.cloudy_stream_synchronize(None))
print("RS_BLOCK", os.environ.get("CLOUDY_HIP_RS_BLOCK", "256"), "ms per call", (time.perf_counter() - t0) / 10 * 1e3, "checksum", float(np.nansum(out.columns_to_numpy(1000, 0))))
