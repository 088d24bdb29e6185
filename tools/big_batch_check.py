#!/usr/bin/env python3
"""64-bit indexing check on one GPU: 8e8 parcels x 6 planes = 4.8e9 elements per array (> 2^32), 38 GB in + 38 GB out.
The batch repeats a 2^20-parcel tile, so every tile of the output must equal the first one."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from __graft_entry__ import load_package

pkg = load_package()
L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 800_000_000
tile = 1 << 20
wl = bench.make_workload("cfg3a", tile, seed=3)
plan = wl["coal_data"].plan(wl["dist_types"])
EXIT_NOMEM = 77  # the caller (tests/test_gpu_parity.py) skips on exactly this status
try:
    m = pkg.DeviceArray(6, n)
    dm = pkg.DeviceArray(6, n)
except pkg._lib.CloudyError as e:
    if e.code == pkg._lib.ENOMEM:
        print(f"cloudy_malloc: CLOUDY_ENOMEM for 2 x {6 * n * 8 / 1e9:.0f} GB", file=sys.stderr)
        sys.exit(EXIT_NOMEM)
    raise
host = np.ascontiguousarray(wl["mom"])
for q in range(6):                                   # replicate the tile along every plane
    for off in range(0, n, tile):
        cnt = min(tile, n - off)
        pkg._lib.check(L.cloudy_memcpy_h2d(C.c_void_p(m.ptr + 8 * (q * n + off)), host[q].ctypes.data_as(C.c_void_p),
                                           8 * cnt, None))
pkg._lib.check(L.cloudy_stream_synchronize(None))
t0 = time.perf_counter()
pkg._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None))
pkg._lib.check(L.cloudy_stream_synchronize(None))
dt = time.perf_counter() - t0
ref = np.empty((6, tile))
chk = np.empty(tile)
for q in range(6):
    pkg._lib.check(L.cloudy_memcpy_d2h(ref[q].ctypes.data_as(C.c_void_p), C.c_void_p(dm.ptr + 8 * q * n), 8 * tile, None))
bad = 0
for q in range(6):
    for off in (tile * 7, (n // tile // 2) * tile, ((n - 1) // tile) * tile):
        cnt = min(tile, n - off)
        pkg._lib.check(L.cloudy_memcpy_d2h(chk.ctypes.data_as(C.c_void_p), C.c_void_p(dm.ptr + 8 * (q * n + off)), 8 * cnt, None))
        pkg._lib.check(L.cloudy_stream_synchronize(None))
        bad += int(not np.array_equal(chk[:cnt], ref[q][:cnt], equal_nan=True))
print(f"n = {n}: {6 * n / 2**32:.2f} x 2^32 elements per array, one launch {dt * 1e3:.1f} ms = {n / dt:.3e} parcel-RHS/s, "
      f"{96 * n / dt / 1e12:.2f} TB/s, mismatching tiles: {bad}")
sys.exit(1 if bad else 0)
