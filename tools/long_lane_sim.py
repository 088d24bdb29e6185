#!/usr/bin/env python3
"""CPU simulation of the active-lane fraction of the converged-mode kernels under different parcel-ranking schemes (no GPU).
The same-rule oracle (oracle/cloudy_oracle_quad.c) counts the panel evaluations of every parcel per phase (0: homogeneous kernels;
1 / 2: the Long kernel's two loops) -- the numbers the device writes into its hint bytes.  A wave runs a loop for as long as its
lane with the most evaluations; a workgroup ranks its parcels by a key.
usage: python tools/long_lane_sim.py <long|hydrodynamic> [n_parcels] [c1 c2]      (c1, c2: instructions per trip of the two loops)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import cloudy_oracle as O  # noqa: E402


def counts(kname, n, seed=bench.SEED):
    N = 3
    mom = bench.synth_moments(N, n, seed)
    p = O.make_params([O.GAMMA] * N, np.zeros((1, 1)), (np.inf,) * N, norms=bench.NORMS)
    kf = {"long": O.kernel_func(O.KF_LONG, 5.236e-10, 9.44e9, 5.78), "hydrodynamic": O.kernel_func(O.KF_HYDRODYNAMIC, 1e2 * np.pi)}[kname]
    kfn = O.get_normalized_kernel_func(kf, bench.NORMS)
    L = O.lib()
    L.co_conv_phase_evals.argtypes = [C.POINTER(C.c_long), C.c_int]
    L.co_conv_phase_evals.restype = None
    ev = np.zeros((n, 3), dtype=np.int64)
    buf = (C.c_long * 3)()
    L.co_conv_phase_evals(buf, 1)
    for i in range(n):
        O.rhs_coal_numerical_converged_batch(p, kfn, 8, mom[:, i:i + 1].copy(), n_threads=1)
        L.co_conv_phase_evals(buf, 1)
        ev[i] = buf[:]
    return ev


def util(cost_loops, order_per_loop, wg):
    """cost_loops: list of (per-parcel trips, instructions per trip); order_per_loop: list of rank keys (None: memory order), one
    per loop (the same key for every loop = one ranking per call).  -> useful lane-instructions / issued lane-instructions"""
    n = len(cost_loops[0][0]) // wg * wg
    useful = issued = 0.0
    for (trips, c), key in zip(cost_loops, order_per_loop):
        t = trips[:n].reshape(-1, wg)
        if key is not None:
            idx = np.argsort(key[:n].reshape(-1, wg), axis=1, kind="stable")
            t = np.take_along_axis(t, idx, axis=1)
        w = t.reshape(t.shape[0], wg // 64, 64)
        useful += c * t.sum()
        issued += c * 64 * w.max(axis=2).sum()
    return useful / issued


def main():
    kname = sys.argv[1] if len(sys.argv) > 1 else "long"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64 * 1024
    c1, c2 = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (2440.0, 4700.0)
    cache = f"/tmp/long_lane_sim_{kname}_{n}.npy"
    ev = np.load(cache) if os.path.exists(cache) else counts(kname, n)
    np.save(cache, ev)
    if kname != "long":
        e = ev[:, 0]
        print(f"{kname}: {e.mean():.1f} evaluations per parcel (max {e.max()})")
        for wg in (256, 512, 1024):
            print(f"  workgroup {wg}: no ranking {util([(e, 1.0)], [None], wg):.3f}   ranked by the evaluations themselves "
                  f"{util([(e, 1.0)], [np.minimum(e, 255)], wg):.3f}")
        return
    p1, p2 = ev[:, 1], ev[:, 2]
    print(f"long: phase 1 {p1.mean():.1f} evaluations per parcel (max {p1.max()}), phase 2 {p2.mean():.1f} (max {p2.max()}; "
          f"{(p2 > 0).mean():.2f} of the parcels have a hole); instruction shares {c1 * p1.mean():.0f} / {c2 * p2.mean():.0f}")
    loops = [(p1, c1), (p2, c2)]
    key44 = (np.minimum(p2, 15) << 4) | np.minimum(p1 >> 2, 15)
    key16 = (np.minimum(p2, 255) << 8) | np.minimum(p1, 255)
    tot = np.minimum((c1 * p1 + c2 * p2) / c1, 255).astype(np.int64)
    for wg in (256, 512):
        print(f"  workgroup {wg}:")
        print(f"    no ranking                                   {util(loops, [None, None], wg):.3f}")
        print(f"    one rank, key = 4 bits p2 | 4 bits p1/4 (r5) {util(loops, [key44, key44], wg):.3f}")
        print(f"    one rank, key = 8 bits p2 | 8 bits p1        {util(loops, [key16, key16], wg):.3f}")
        print(f"    one rank, key = weighted total               {util(loops, [tot, tot], wg):.3f}")
        print(f"    a rank per phase (p1, then p2)               {util(loops, [p1, p2], wg):.3f}")
        print(f"      phase 1 alone {util(loops[:1], [p1], wg):.3f} (r5 key: {util(loops[:1], [key44], wg):.3f}), "
              f"phase 2 alone {util(loops[1:], [p2], wg):.3f} (r5 key: {util(loops[1:], [key44], wg):.3f})")


if __name__ == "__main__":
    main()
