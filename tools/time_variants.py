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
            v = j["variants"]
            res.setdefault(lib, []).append((j["roofline"]["kernel_ms"], v["cfg3b"]["kernel_ms"], v["cfg4"]["kernel_ms"],
                                            v["cfg3b_f32_fast"]["kernel_ms"], v["cfg2"]["kernel_ms"],
                                            v["cfg3a_fused_ssprk33"]["ms_per_call"], v["cfg3a_f32_planes"]["kernel_ms"]))
        except Exception as e:
            print(lib, "FAILED", p.stderr[-400:])
for lib, v in res.items():
    print(f"{os.path.basename(lib):40s} cfg3a {min(x[0] for x in v):.4f}  cfg3b {min(x[1] for x in v):.3f}  cfg4 {min(x[2] for x in v):.2f}  "
          f"cfg3b_f32_fast {min(x[3] for x in v):.3f}  cfg2 {min(x[4] for x in v):.4f}  fused_ssprk33 {min(x[5] for x in v):.3f}  "
          f"cfg3a_f32_planes {min(x[6] for x in v):.4f} ms")
