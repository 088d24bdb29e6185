#!/usr/bin/env python3
"""Register / scratch / LDS use of the plan-specialised (hiprtc) kernels of named bench workloads, without a GPU:
cloudy_jit_selfcheck compiles the plan's translation unit for gfx950, CLOUDY_HIP_JIT_DUMP keeps the code object, and
the figures are read from its metadata notes.
usage: python tools/jit_resources.py [--keep DIR] cfg3b cfg4 moving4 cfg4q ..."""
import argparse
import ctypes as C
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keep", default="", help="directory for the dumped .hip / .co files (default: a temp dir)")
    ap.add_argument("workloads", nargs="+")
    a = ap.parse_args()
    dump = a.keep or tempfile.mkdtemp(prefix="cloudy_jit_")
    os.makedirs(dump, exist_ok=True)
    os.environ["CLOUDY_HIP_JIT_DUMP"] = dump
    import numpy as np

    import bench
    from __graft_entry__ import load_package

    pkg = load_package()
    L = pkg.lib()
    for name in a.workloads:
        before = set(glob.glob(os.path.join(dump, "*.co")))
        if name.startswith(("conv:", "quad:")):
            # conv:<kernel>:<dist,dist,...>  e.g. conv:long:1,1,1  conv:linear:3,3  (1 Gamma, 0 Exponential, 3 Lognormal)
            mode, kname, dists = name.split(":")
            kf = {"constant": pkg.ConstantKernelFunction(1e-4), "linear": pkg.LinearKernelFunction(5.0),
                  "hydrodynamic": pkg.HydrodynamicKernelFunction(1e2 * np.pi),
                  "long": pkg.LongKernelFunction(5.236e-10, 9.44e9, 5.78)}[kname]
            conv = mode == "conv"
            d = pkg.NumericalPlan.make_desc([int(x) for x in dists.split(",")], kf, bench.NORMS, 8 if conv else 10,
                                            kernel_func_is_normalized=False,
                                            quad_mode=pkg.QUAD_CONVERGED if conv else pkg.QUAD_FIXED)
            keep = None
        elif name in ("cfg4q", "cfg4q_converged"):
            conv = name.endswith("converged")
            d = pkg.NumericalPlan.make_desc([1, 1, 1], pkg.HydrodynamicKernelFunction(1e2 * np.pi), bench.NORMS,
                                            8 if conv else 10, kernel_func_is_normalized=False,
                                            quad_mode=pkg.QUAD_CONVERGED if conv else pkg.QUAD_FIXED)
            keep = None
        else:
            spec = bench.workload_spec(name)
            # (one sedimentation velocity term, so that the fused column integrator is compiled as well)
            d, keep = pkg.Plan.make_desc([1] * spec["n_modes"], bench.kernel_matrix(spec), spec["thresholds"], bench.NORMS,
                                         1 if spec.get("moving") else 0, vel=((50.0, 1.0 / 6),))
        if L.cloudy_jit_selfcheck(C.byref(d), b"gfx950") != 0:
            raise SystemExit(f"{name}: " + L.cloudy_last_error().decode())
        for co in sorted(set(glob.glob(os.path.join(dump, "*.co"))) - before):
            notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count")[1:]:
                get = lambda key: (re.search(rf"\.{key}:\s*(\S+)", blk) or [None, "?"])[1]
                print(f"{name}: {get('name')}: {get('vgpr_count')} VGPRs, {get('vgpr_spill_count')} spilled, "
                      f"scratch {get('private_segment_fixed_size')} B, LDS {get('group_segment_fixed_size')} B, "
                      f"{get('sgpr_count')} SGPRs  [{os.path.basename(co)}]")


if __name__ == "__main__":
    main()
