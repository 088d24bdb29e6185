#!/usr/bin/env python3
"""Where do the plan-time compiled and the ahead-of-time kernels of ONE fuzzer configuration differ?  (round 6)
usage: python tools/jit_aot_diff.py <config index> [--seed 1] [--wild]   -- converged-mode configurations of tools/fuzz_parity.py"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

import fuzz_parity as F
from __graft_entry__ import load_package
from oracle import cloudy_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("index", type=int)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--wild", action="store_true")
ap.add_argument("--parcels", type=int, default=400)
a = ap.parse_args()
pkg = load_package()
rng = np.random.default_rng(a.seed)
for c in range(a.index + 1):
    cfg = F.random_numerical_config(rng, a.wild, False)
kf_cls = [pkg.ConstantKernelFunction, pkg.LinearKernelFunction, pkg.HydrodynamicKernelFunction, pkg.LongKernelFunction]
kfn = pkg.get_normalized_kernel_func(kf_cls[cfg["kind"]](*cfg["params"]), cfg["norms"])
okf = O.get_normalized_kernel_func(O.kernel_func(cfg["kind"], *cfg["params"]), cfg["norms"])
op = O.make_params(cfg["dist"], np.zeros((1, 1)), (F.INF,) * cfg["N"], norms=cfg["norms"], k_range=cfg["k_range"])
q = int(min(max(cfg["nq"], 4), 16))
mom = F.moments_for(cfg["dist"], a.parcels, 5000 + a.index)
jit = pkg.NumericalPlan(cfg["dist"], kfn, cfg["norms"], q, k_range=cfg["k_range"], specialize=1, quad_mode=1)
aot = pkg.NumericalPlan(cfg["dist"], kfn, cfg["norms"], q, k_range=cfg["k_range"], specialize=-1, quad_mode=1)
x, y = F.run(pkg, jit, mom, np.float64), F.run(pkg, aot, mom, np.float64)
want, scale = O.rhs_coal_numerical_converged_batch(op, okf, q, mom, with_scale=True)
d = np.abs(x - y) / np.maximum(scale, 1e-300)
d[~np.isfinite(d)] = 0.0
per = d.max(axis=0)
print(cfg)
print("parcels with |jit - aot| / scale > 1e-14:", int((per > 1e-14).sum()), "of", per.size, "; max", per.max())
y2 = F.run(pkg, aot, mom, np.float64)
print("ahead-of-time kernel, second call (ranked by the first call's hints) against its first: max |diff| / scale",
      float(np.nanmax(np.abs(y2 - y) / np.maximum(scale, 1e-300))))
y3 = F.run(pkg, aot, mom, np.float64)
print("third call against the second:", float(np.nanmax(np.abs(y3 - y2) / np.maximum(scale, 1e-300))),
      "; jit second call against its first:", float(np.nanmax(np.abs(F.run(pkg, jit, mom, np.float64) - x) / np.maximum(scale, 1e-300))))
for name, plan in (("jit", jit), ("aot", aot)):
    fresh = [F.run(pkg, pkg.NumericalPlan(cfg["dist"], kfn, cfg["norms"], q, k_range=cfg["k_range"], specialize=1 if name == "jit" else -1, quad_mode=1),
                   mom, np.float64) for _ in range(3)]
    print(name, "three fresh plans, first calls: max |diff| / scale", float(np.nanmax(np.abs(fresh[1] - fresh[0]) / np.maximum(scale, 1e-300))),
          float(np.nanmax(np.abs(fresh[2] - fresh[0]) / np.maximum(scale, 1e-300))), "; against the oracle",
          float(np.nanmax(np.abs(fresh[0] - want) / np.maximum(scale, 1e-300))))
i0 = int(np.argmax(per))
np.set_printoptions(linewidth=200)
print("worst parcel", i0, "\n jit   ", x[:, i0], "\n aot   ", y[:, i0], "\n oracle", want[:, i0], "\n scale ", scale[:, i0])
for i in np.argsort(-per)[:5]:
    ej, ea = np.abs(x[:, i] - want[:, i]) / scale[:, i], np.abs(y[:, i] - want[:, i]) / scale[:, i]
    print(f"parcel {i}: |jit-aot| {per[i]:.2e}  |jit-oracle| {np.nanmax(ej):.2e}  |aot-oracle| {np.nanmax(ea):.2e}  prm",
          np.array2string(O.update_dist_batch(op, mom[:, i:i + 1])[:, 0], precision=4))
