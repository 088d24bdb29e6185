"""The plain-C example of the boundary (examples/box_single_gamma.c): compiles with gcc against include/cloudy_hip.h
and libcloudy_hip.so alone (no Python, no HIP headers).  CPU: builds, links and reports CLOUDY_ENODEVICE cleanly;
GPU: runs the reference's box_single_gamma driver staged and fused and checks them against each other."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "box_single_gamma")
    libdir = os.path.join(ROOT, "cloudy.jl_amd")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "box_single_gamma.c"), "-o", exe, "-L" + libdir, "-lcloudy_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_example_builds_and_fails_loudly_without_gpu(tmp_path, cloudy):
    exe = _build(tmp_path)
    if cloudy.device_count() > 0:
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "no HIP device" in r.stderr   # CLOUDY_ENODEVICE, no CPU fallback


@pytest.mark.gpu
def test_c_example_runs_on_gpu(tmp_path, gpu_cloudy):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK")
    print(r.stdout)
