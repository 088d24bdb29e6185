"""Launch-ramp timing script (run by hand on a GPU box; not a test -- pytest collects tests/ only, see pytest.ini)."""
import sys, time, ctypes as C


def main():
    sys.path.insert(0, '.')
    import __graft_entry__ as ge, bench
    pkg = ge.load_package(); L = pkg.lib()
    wl = bench.make_workload("cfg3a", 10_000_000)
    plan = wl["coal_data"].plan(wl["dist_types"])
    m = pkg.DeviceArray.from_numpy(wl["mom"]); dm = pkg.DeviceArray.zeros(6, 10_000_000)
    n = 10_000_000
    for _ in range(10): L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None)
    L.cloudy_stream_synchronize(None)
    for K in (10, 50, 200, 1000):
        t0 = time.perf_counter()
        for _ in range(K): L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None)
        t1 = time.perf_counter()
        L.cloudy_stream_synchronize(None)
        t2 = time.perf_counter()
        ms = C.c_float(); L.cloudy_time_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None, K, C.byref(ms))
        print(f"K={K}: enqueue {1e3*(t1-t0):.2f} ms, total wall {1e3*(t2-t0):.2f} ms = {1e3*(t2-t0)/K:.4f} ms/step; events {ms.value:.4f} ms/step")



if __name__ == "__main__":
    main()
