"""Edge shapes of the column kernels on the GPU box (round 5): one-cell columns, one column, columns of exactly 1024 cells, a leading
dimension larger than the batch, three modes -- cloudy_rainshaft_rhs in one launch against its two-launch path (bit for bit, padding
untouched) and the column integrator in place against out of place."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from __graft_entry__ import load_package
pkg = load_package(); L = pkg.lib()
for name, nz, ncol, pad in (("cfg3b", 1, 777, 0), ("cfg3b", 2, 300, 3), ("cfg3b", 1024, 2, 0), ("cfg3b", 513, 3, 5), ("cfg3a", 255, 5, 1), ("cfg3b", 20, 1, 0), ("cfg4", 64, 17, 2)):
    n = nz * ncol; ld = n + pad
    wl = bench.make_workload(name, n, seed=41)
    plan = wl["coal_data"].plan(wl["dist_types"], vel=((50.0, 1.0 / 6),))
    buf = np.zeros((wl["mom"].shape[0], ld)); buf[:, :n] = wl["mom"]
    u = pkg.DeviceArray.from_numpy(buf)
    res = {}
    for fused in ("0", "1"):
        os.environ["CLOUDY_HIP_RS_FUSED_RHS"] = fused
        rhs = pkg.DeviceArray.zeros(buf.shape[0], ld); flux = pkg.DeviceArray.zeros(buf.shape[0], ld)
        pkg._lib.check(L.cloudy_rainshaft_rhs(plan.handle, nz, ncol, ld, u.ptr, C.c_double(150.0), flux.ptr, rhs.ptr, None))
        res[fused] = (rhs.to_numpy(), flux.to_numpy())
    ok = np.array_equal(res["0"][0], res["1"][0], equal_nan=True) and np.array_equal(res["0"][1], res["1"][1], equal_nan=True)
    # the integrator on the same shapes: one step in place vs out of place
    o1 = pkg.DeviceArray.zeros(buf.shape[0], ld)
    pkg._lib.check(L.cloudy_rainshaft_ssprk33_steps(plan.handle, nz, ncol, ld, u.ptr, o1.ptr, C.c_double(150.0), C.c_double(1e-3), 2, None))
    u2 = pkg.DeviceArray.from_numpy(buf)
    pkg._lib.check(L.cloudy_rainshaft_ssprk33_steps(plan.handle, nz, ncol, ld, u2.ptr, u2.ptr, C.c_double(150.0), C.c_double(1e-3), 2, None))
    ok2 = np.array_equal(o1.to_numpy()[:, :n], u2.to_numpy()[:, :n], equal_nan=True) and np.all(o1.to_numpy()[:, n:] == 0)
    print(name, nz, ncol, pad, "rhs one == two launches:", ok, " pad untouched:", bool(np.all(res["1"][0][:, n:] == 0)), " integrator in/out of place equal:", ok2, flush=True)
