#!/usr/bin/env python3
"""A/B timing of libcloudy_hip.so builds on the GPU box (one bench.py subprocess per build, same device).
usage: python tools/time_variants.py [--parcels N] [--steps K] lib1.so lib2.so ...   (path or 'default')"""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--parcels", type=int, default=10_000_000)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
res = {}
for rnd in range(a.rounds):
    for lib in a.libs:
        env = dict(os.environ)
        if lib != "default":
            env["CLOUDY_HIP_LIB"] = os.path.abspath(lib)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", str(a.steps),
                            "--warmup", "3", "--parcels", str(a.parcels)], env=env, capture_output=True, text=True)
        try:
            j = json.loads(p.stdout.strip().splitlines()[-1])
            res.setdefault(lib, []).append((j["roofline"]["kernel_ms"], j["variants"]["cfg3b"]["kernel_ms"]))
        except Exception as e:
            print(lib, "FAILED", p.stderr[-400:])
for lib, v in res.items():
    print(f"{os.path.basename(lib):40s} cfg3a kernel_ms {min(x[0] for x in v):.4f}  cfg3b kernel_ms {min(x[1] for x in v):.3f}   all={v}")
