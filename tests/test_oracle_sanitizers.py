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
    # the NumericalCoalStyle restatements (cloudy_oracle_quad.c, cloudy_oracle_adaptive.c) up to eight modes: per-rule mark
    # lists, per-mode tables (round 5: a six-mode Long-kernel case of tools/fuzz_parity.py --converged --big wrote past the 96
    # marks that four modes needed)
    for types, kind, args in [([0, 1, 1, 1, 3, 1], O.KF_LONG, (3.48e-9, 6.66e9, 29.6)), ([1] * 8, O.KF_HYDRODYNAMIC, (3e2,)),
                              ([1, 3, 1, 0, 1, 1, 1], O.KF_LINEAR, (5.0,)), ([1, 1, 1], O.KF_LONG, (5.236e-10, 9.44e9, 5.78)),
                              ([3, 1, 0, 1, 1], O.KF_CONSTANT, (1e-4,))]:
        N = len(types)
        p = O.make_params(types, np.zeros((1, 1)), (inf,) * N, norms=(1e6, 1e-9))
        kf = O.get_normalized_kernel_func(O.kernel_func(kind, *args), (1e6, 1e-9))
        edges = np.logspace(-12, -4, N + 1)
        rows = []
        n = 12
        for i, t in enumerate(types):   # one size class per mode, number densities falling with size, wide and narrow shapes
            nn = 10.0 ** rng.uniform(6 - 12.0 * i / N, 9 - 12.0 * i / N, n)
            k = rng.uniform(0.5, 7.0, n)
            th = 10.0 ** rng.uniform(np.log10(edges[i]), np.log10(edges[i + 1]), n) / k
            m = [nn, nn * k * th, nn * k * (k + 1) * th * th]
            m[2][:2] = m[1][:2] ** 2 / m[0][:2]        # zero variance: the shape clamp (narrow cores: the longest mark ladders)
            rows += m[:2] + ([m[2]] if t in (1, 3) else [])
        mom = np.ascontiguousarray(np.stack(rows))
        mom[:, 2] = 0.0
        O.rhs_coal_numerical_converged_batch(p, kf, 8, mom, with_scale=True)
        O.rhs_coal_numerical_batch(p, kf, 6, mom, with_noise=True)
    pd = [O.make_dist(1, 100.0, 0.02, 2.0), O.make_dist(1, 20.0, 0.2, 3.0), O.make_dist(0, 4.0, 1.5, 1.0), O.make_dist(1, 0.5, 12.0, 3.5),
          O.make_dist(1, 0.05, 100.0, 4.0)]
    O.get_coal_ints_numerical_adaptive(pd, O.kernel_func(O.KF_LINEAR, 5e-3), 1e-6, 1e-8)
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
