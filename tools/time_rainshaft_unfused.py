"""the column RHS (cloudy_rainshaft_rhs) on the batch of tools/time_rainshaft_block.py: what one evaluation costs outside the fused
integrator -- one launch (round 5), or with CLOUDY_HIP_RS_FUSED_RHS=0 the cell kernel + the divergence launch (PMC: tools/pmc_one.sh)"""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from __graft_entry__ import load_package
pkg = load_package(); L = pkg.lib()
nz, ncol = 20, 500000
n = nz * ncol
wl = bench.make_workload("cfg3b", n, seed=7)
plan = wl["coal_data"].plan(wl["dist_types"], vel=((50.0, 1.0 / 6),))
u = pkg.DeviceArray.from_numpy(wl["mom"]); out = pkg.DeviceArray.zeros(*wl["mom"].shape); work = pkg.DeviceArray.zeros(*wl["mom"].shape)
def run():
    pkg._lib.check(L.cloudy_rainshaft_rhs(plan.handle, nz, ncol, n, u.ptr, C.c_double(150.0), work.ptr, out.ptr, None))
for _ in range(12): run()
pkg._lib.check(L.cloudy_stream_synchronize(None))
t0 = time.perf_counter()
for _ in range(10): run()
pkg._lib.check(L.cloudy_stream_synchronize(None))
print("cloudy_rainshaft_rhs, CLOUDY_HIP_RS_FUSED_RHS =", os.environ.get("CLOUDY_HIP_RS_FUSED_RHS", "1"), ": ms per evaluation of 1e7 cells", (time.perf_counter() - t0) / 10 * 1e3)
