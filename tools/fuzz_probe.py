#!/usr/bin/env python3
"""Which parcels carry the worst error of one configuration of tools/fuzz_parity.py: python tools/fuzz_probe.py --wild --index 26"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bench
import fuzz_parity as F
from __graft_entry__ import load_package
from oracle import cloudy_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--wild", action="store_true")
ap.add_argument("--big", action="store_true")
ap.add_argument("--index", type=int, required=True)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--parcels", type=int, default=400)
a = ap.parse_args()
pkg = load_package()
rng = np.random.default_rng(a.seed)
for c in range(a.index + 1):
    cfg = F.random_config(rng, a.wild, a.big)
N = cfg["N"]
kernels = tuple(tuple(pkg.CoalescenceTensor(cfg["kc"][j, k]) for k in range(N)) for j in range(N))
npm = tuple({0: 2, 1: 3, 2: 2, 3: 3}[t] for t in cfg["dist"])
ts = pkg.MovingThreshold() if cfg["moving"] else pkg.FixedThreshold()
cd = pkg.CoalescenceData(kernels, npm, cfg["thr"], cfg["norms"], ts)
op = O.make_params(cfg["dist"], cfg["kc"], cfg["thr"], norms=cfg["norms"], k_range=cfg["k_range"],
                   threshold_style=O.MOVING_THRESHOLD if cfg["moving"] else O.FIXED_THRESHOLD)
mom = F.moments_for(cfg["dist"], a.parcels, 1000 + a.index)
plan = cd.plan(cfg["dist"], k_range=cfg["k_range"], specialize=1)
got = F.run(pkg, plan, mom, np.float64)
want, scale = O.rhs_coal_batch(op, mom, with_scale=True)
prm = O.update_dist_batch(op, mom)
with np.errstate(invalid="ignore"):
    e = np.where(np.isfinite(want), np.abs(got - want) / np.maximum(scale, 1e-300), 0.0).max(axis=0)
print("config", {k: v for k, v in cfg.items() if k != "kc"}, "thr normalised", [t / cfg["norms"][1] for t in cfg["thr"]])
for i in np.argsort(e)[::-1][:6]:
    print(f"parcel {i}: err {e[i]:.2e}  (n, theta|mu, k|sigma) per mode:", [tuple(float(f"{prm[3 * m + q, i]:.4g}") for q in range(3)) for m in range(N)])
