#!/usr/bin/env python3
"""ONE timing driver for the GPU box (round 6: replaces fourteen one-off tools/time_*.py; what they measured is recorded in
profiles/ and in the commit log).  HIP events on the launch stream after >= 150 ms of back-to-back launches (bench._sustained_ms);
every experiment switch of the library is an environment variable, so an A/B run is two calls of the same line:

  python tools/timeit.py kernels cfg3b cfg4 moving4 [--reps R] [--parcels P] [--error] [--dtype 0|1|2|3]
        cloudy_coal_rhs of named bench workloads (tensor plans); --error: max |hip - oracle| / scale on 20000 parcels
  python tools/timeit.py conv <constant|linear|hydrodynamic|long> <dist,dist,...> [n] [--fused] [--fixed NQ] [--lognorm-example]
        a NumericalCoalStyle plan on the synthetic batch (1 Gamma, 0 Exponential, 3 Lognormal); --fused: cloudy_ssprk33_steps beside
        three launches; --fixed NQ: the fixed Gauss rule instead of converged mode
  python tools/timeit.py columns [--nz NZ] [--cells N] [--workload cfg3b] [--steps S]
        cloudy_rainshaft_ssprk33_steps and cloudy_rainshaft_rhs (ms per 1e7 cells)
  python tools/timeit.py integrators [--parcels P]
        cloudy_ssprk33_steps / cloudy_tsit5_steps of the tensor plans cfg3a, cfg3b, cfg2
  python tools/timeit.py host [--parcels P]
        cloudy_coal_rhs_host on cfg3a: the PCIe-inclusive rate (host arrays staged through the device)

switches read by the library: CLOUDY_HIP_LIB, CLOUDY_HIP_JIT, CLOUDY_HIP_JIT_DEFS, CLOUDY_HIP_CONV_HINTS, CLOUDY_HIP_CONV_BLOCK,
CLOUDY_HIP_CONV_ROUNDS, CLOUDY_HIP_LONG_SPLIT, CLOUDY_HIP_JIT_QUAD_WAVES, CLOUDY_HIP_RS_BLOCK, CLOUDY_HIP_RS_FUSED_RHS, ..."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def kernel_functions(pkg):
    return {"constant": pkg.ConstantKernelFunction(1e-4), "linear": pkg.LinearKernelFunction(5.0),
            "hydrodynamic": pkg.HydrodynamicKernelFunction(1e2 * np.pi), "long": pkg.LongKernelFunction(5.236e-10, 9.44e9, 5.78)}


def cmd_kernels(a, pkg, L):
    out = []
    for name in a.names:
        n = a.parcels or bench.workload_spec(name)["default_parcels"]
        wl = bench.make_workload(name, n)
        plan = wl["coal_data"].plan(wl["dist_types"], dtype=a.dtype)
        dt = pkg.plane_dtype(plan) if hasattr(pkg, "plane_dtype") else np.float64
        m, dm = pkg.DeviceArray.from_numpy(wl["mom"].astype(dt)), pkg.DeviceArray.zeros(plan.nmom, n, dt)
        ms = bench._event_ms(pkg, plan, m, dm, a.reps)
        err = ""
        if a.error:
            from oracle import cloudy_oracle as O

            ns = min(n, 20000)
            want, scale = O.rhs_coal_batch(bench.oracle_params(name), np.ascontiguousarray(wl["mom"][:, :ns]), with_scale=True)
            got = dm.columns_to_numpy(ns)
            ok = np.isfinite(want) & (scale > 0)
            err = f", err {float((np.abs(got - want)[ok] / scale[ok]).max()):.1e}, nan {int(np.isnan(got[ok]).sum())}"
        out.append(f"{name} {ms:.3f} ms ({n / ms * 1e3:.3e}/s, jit={int(plan.specialized)}{err})")
        del m, dm
    print(os.environ.get("CLOUDY_HIP_JIT_DEFS", "") or "default", "|", " | ".join(out), flush=True)


def cmd_conv(a, pkg, L):
    dists = [int(x) for x in a.dists.split(",")]
    N = len(dists)
    n = a.n
    if a.lognorm_example:
        mom = bench.lognorm_example_moments(n)
    elif N <= 4:
        mom = bench.synth_moments(N, n, bench.SEED)
    else:   # N size classes between 1e-12 and 1e-4 kg, number densities falling with size (as tests/test_gpu_parity.py::many_mode_moments)
        rng = np.random.Generator(np.random.Philox(key=bench.SEED))
        edges = np.logspace(-12, -4, N + 1)
        mom = np.ascontiguousarray(np.concatenate([bench._gamma_mode(rng, n, 1e6 * 10.0 ** (-1.5 * i * 8 / N), 1e9 * 10.0 ** (-1.5 * i * 8 / N),
                                                                      0.5 if i == 0 else 1.0, 7.0, edges[i], edges[i + 1]) for i in range(N)]))
    m, dm = pkg.DeviceArray.from_numpy(mom), pkg.DeviceArray.zeros(3 * N, n)
    kfn = pkg.get_normalized_kernel_func(kernel_functions(pkg)[a.kernel], bench.NORMS)
    plan = pkg.NumericalPlan(dists, kfn, bench.NORMS, a.fixed or 8, specialize=1,
                             quad_mode=pkg.QUAD_FIXED if a.fixed else pkg.QUAD_CONVERGED)
    ms = bench._sustained_ms(pkg, lambda: pkg._lib.check(L.cloudy_coal_rhs(plan.handle, n, n, m.ptr, dm.ptr, None)), min_reps=a.reps,
                             max_reps=max(a.reps, 20))
    line = f"{'fixed ' + str(a.fixed) if a.fixed else 'converged'} {a.kernel} {dists}: {ms:.3f} ms per {n} parcels = {n / ms * 1e3:.3e} parcel-RHS/s"
    if a.fused:
        mi = bench._sustained_ms(pkg, lambda: pkg._lib.check(
            L.cloudy_ssprk33_steps(plan.handle, n, n, m.ptr, dm.ptr, C.c_double(1e-3), 1, None)), min_reps=3, max_reps=10)
        line += f"; cloudy_ssprk33_steps (3 evaluations) {mi:.3f} ms = {mi / (3 * ms):.2f} x three launches"
    print(line, flush=True)


def cmd_columns(a, pkg, L):
    nz, ncol = a.nz, max(a.cells // a.nz, 1)
    n = nz * ncol
    wl = bench.make_workload(a.workload, n, seed=7)
    plan = wl["coal_data"].plan(wl["dist_types"], vel=((50.0, 1.0 / 6),))
    u, out = pkg.DeviceArray.from_numpy(wl["mom"]), pkg.DeviceArray.zeros(*wl["mom"].shape)
    flux = pkg.DeviceArray.zeros(*wl["mom"].shape)
    ms_i = bench._sustained_ms(pkg, lambda: pkg._lib.check(
        L.cloudy_rainshaft_ssprk33_steps(plan.handle, nz, ncol, n, u.ptr, out.ptr, 150.0, 1e-3, a.steps, None)), min_reps=3)
    ms_r = bench._sustained_ms(pkg, lambda: pkg._lib.check(
        L.cloudy_rainshaft_rhs(plan.handle, nz, ncol, n, u.ptr, 150.0, flux.ptr, out.ptr, None)), min_reps=3)
    print(f"{a.workload} columns nz={nz} x {ncol}: integrator {ms_i:.3f} ms per call of {a.steps} step(s) = {ms_i / (3 * a.steps) * 1e7 / n:.3f} ms "
          f"per evaluation of 1e7 cells; cloudy_rainshaft_rhs {ms_r * 1e7 / n:.3f} ms per 1e7 cells  (CLOUDY_HIP_RS_BLOCK="
          f"{os.environ.get('CLOUDY_HIP_RS_BLOCK', 'auto')})", flush=True)


def cmd_host(a, pkg, L):
    """the PCIe-inclusive rate: cloudy_coal_rhs_host stages host arrays through the device (never the bench's `value`)"""
    import time

    n = a.parcels or 10_000_000
    wl = bench.make_workload("cfg3a", n)
    plan = wl["coal_data"].plan(wl["dist_types"])
    mom, out = np.ascontiguousarray(wl["mom"]), np.zeros_like(wl["mom"])
    pin, pout = mom.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        pkg._lib.check(L.cloudy_coal_rhs_host(plan.handle, n, n, pin, pout))
        best = min(best, time.perf_counter() - t0)
    gb = 2 * plan.nmom * 8 * n / 1e9
    print(f"cloudy_coal_rhs_host, cfg3a, {n} parcels (pageable host arrays, {gb:.2f} GB over PCIe): {best * 1e3:.1f} ms = {n / best:.3e} parcel-RHS/s "
          f"= {gb / best:.1f} GB/s of host traffic", flush=True)


def cmd_integrators(a, pkg, L):
    for name in ("cfg3a", "cfg3b", "cfg2"):
        n = a.parcels or 4_000_000
        wl = bench.make_workload(name, n, seed=7)
        plan = wl["coal_data"].plan(wl["dist_types"])
        u, out = pkg.DeviceArray.from_numpy(wl["mom"]), pkg.DeviceArray.zeros(*wl["mom"].shape)
        ms_s = bench._sustained_ms(pkg, lambda: pkg._lib.check(L.cloudy_ssprk33_steps(plan.handle, n, n, u.ptr, out.ptr, C.c_double(1e-3), 4, None)))
        ms_t = bench._sustained_ms(pkg, lambda: pkg._lib.check(L.cloudy_tsit5_steps(plan.handle, n, n, u.ptr, out.ptr, C.c_double(1e-3), 4, None)))
        print(f"{name}, {n} parcels: SSPRK33 {ms_s / 12:.4f} ms per evaluation, Tsit5 {ms_t / 25:.4f} ms per evaluation", flush=True)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    k = sub.add_parser("kernels")
    k.add_argument("names", nargs="+")
    k.add_argument("--reps", type=int, default=5)
    k.add_argument("--parcels", type=int, default=0)
    k.add_argument("--error", action="store_true")
    k.add_argument("--dtype", type=int, default=0)
    c = sub.add_parser("conv")
    c.add_argument("kernel")
    c.add_argument("dists")
    c.add_argument("n", nargs="?", type=int, default=4_000_000)
    c.add_argument("--reps", type=int, default=3)
    c.add_argument("--fused", action="store_true")
    c.add_argument("--fixed", type=int, default=0)
    c.add_argument("--lognorm-example", action="store_true")
    r = sub.add_parser("columns")
    r.add_argument("--nz", type=int, default=20)
    r.add_argument("--cells", type=int, default=10_000_000)
    r.add_argument("--workload", default="cfg3b")
    r.add_argument("--steps", type=int, default=2)
    i = sub.add_parser("integrators")
    i.add_argument("--parcels", type=int, default=0)
    hh = sub.add_parser("host")
    hh.add_argument("--parcels", type=int, default=0)
    a = ap.parse_args()
    pkg = load_package()
    {"kernels": cmd_kernels, "conv": cmd_conv, "columns": cmd_columns, "integrators": cmd_integrators, "host": cmd_host}[a.cmd](a, pkg, pkg.lib())


if __name__ == "__main__":
    main()
