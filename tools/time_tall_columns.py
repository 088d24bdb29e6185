"""columns taller than 1024 cells (stepped stage by stage inside the library): the eager loop against the captured step replayed as a
hipGraph (CLOUDY_HIP_GRAPH), by number of steps and batch size"""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from __graft_entry__ import load_package
pkg = load_package(); L = pkg.lib()
for nz, ncol in ((1500, 3), (1500, 64), (4000, 256)):
    n = nz * ncol
    wl = bench.make_workload("cfg3b", n, seed=31)
    plan = wl["coal_data"].plan(wl["dist_types"], vel=((50.0, 1.0 / 6),))
    u = pkg.DeviceArray.from_numpy(wl["mom"]); out = pkg.DeviceArray.zeros(*wl["mom"].shape)
    for n_steps in (12, 100, 1000):
        row = []
        for g in ("0", "1"):
            os.environ["CLOUDY_HIP_GRAPH"] = g
            best = 1e30
            for rep in range(3):
                t0 = time.perf_counter()
                pkg._lib.check(L.cloudy_rainshaft_ssprk33_steps(plan.handle, nz, ncol, n, u.ptr, out.ptr, C.c_double(150.0), C.c_double(1e-5), n_steps, None))
                pkg._lib.check(L.cloudy_stream_synchronize(None))
                best = min(best, (time.perf_counter() - t0) * 1e3)
            row.append(best)
        print(f"{ncol} columns x {nz} cells, {n_steps} steps: eager {row[0]:.2f} ms, graph {row[1]:.2f} ms", flush=True)
