"""The CPU oracle under AddressSanitizer + UBSan (GPU sanitizers are not available on the pool, so memory safety of
the C restatement -- fixed-size per-mode arrays indexed by N, P, M -- is checked on the CPU build)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = textwrap.dedent("""
    import ctypes, sys, numpy as np
    sys.path.insert(0, {root!r})
    from oracle import cloudy_oracle as O
    O._LIB_PATH = {lib!r}
    O._lib = None
    rng = np.random.default_rng(0)
    inf = float("inf")
    for N, P, types, thr in [(1, 1, [1], (inf,)), (2, 3, [1, 1], (5e-10, inf)), (3, 5, [0, 1, 3], (1e-9, inf, inf)),
                             (4, 5, [2, 1, 0, 1], (1e-10, 1e-9, 1e-7, inf)), (4, 2, [1, 1, 1, 1], (inf,) * 4)]:
        kc = rng.uniform(0, 1, (N, N, P, P)); kc = kc + kc.transpose(0, 1, 3, 2); kc = kc + kc.transpose(1, 0, 2, 3)
        p = O.make_params(types, kc, thr, norms=(1e6, 1e-9), vel=((50.0, 1 / 6),))
        nm = O.nmom_of(p)
        mom = np.abs(rng.normal(size=(nm, 40))) * 1e3
        mom[:, :3] = 0.0
        d, s = O.rhs_coal_batch(p, mom, with_scale=True, n_threads=1)
        O.rainshaft_cell_batch(p, mom, n_threads=1)
        O.rhs_condensation_batch(p, 1e-8, 0.05, mom)
        O.update_dist_batch(p, mom)
    pm = O.make_params([1, 0, 1], np.array([[0, 5.0], [5.0, 0]]), (0.99, 0.5, 1.0), norms=(1e6, 1e-9), threshold_style=1)
    O.rhs_coal_batch(pm, np.abs(rng.normal(size=(8, 16))) * 1e3, n_threads=1)
    print("sanitized run ok")
""")


def test_oracle_under_asan_ubsan(tmp_path):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build not available: " + r.stderr[-200:])
    lib = os.path.join(ROOT, "oracle", "libcloudy_oracle_asan.so")
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    script = tmp_path / "drv.py"
    script.write_text(DRIVER.format(root=ROOT, lib=lib))
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "sanitized run ok" in p.stdout
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr
