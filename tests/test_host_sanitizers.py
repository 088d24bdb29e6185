"""The HOST half of libcloudy_hip.so under AddressSanitizer + UBSan (CPU container only -- GPU sanitizers are
not available on the pool): `make -C cloudy.jl_amd/csrc asan` compiles the C ABI, the descriptor checks and plan building
(host_plan.hpp, host_quad.hpp), the string assembly of the plan-time translation units (jit.hpp) and the communicator glue
with -fsanitize=address,undefined -fno-gpu-sanitize (device code untouched) and links them with the ordinary kernel objects.
Round 5 raised CLOUDY_MAX_MODES from 4 to 8; a mark list of the ORACLE sized for four modes was then written by a six-mode plan
("stack smashing detected", found by a fuzz run).  The same class of bug in the library would run in the caller's process."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cloudy.jl_amd", "libcloudy_hip_asan.so")


def _asan_env():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "cloudy.jl_amd", "csrc"), "-j", "8", "asan"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(LIB):
        pytest.skip("host sanitizer build not available: " + (r.stderr or r.stdout)[-300:])
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True,
                        text=True).stdout.strip()
    if not os.path.isfile(rt):
        pytest.skip("clang's ASan runtime not found")
    return dict(os.environ, CLOUDY_HIP_LIB=LIB, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
                UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="1", CLOUDY_HIP_CACHE_DIR="")


def _clean(p):
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-4000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out and "Sanitizer" not in p.stderr, out[-4000:]


DRIVER = textwrap.dedent("""
    import ctypes as C, sys, numpy as np
    sys.path.insert(0, {root!r})
    import __graft_entry__ as ge
    import bench
    pkg = ge.load_package()
    L, E = pkg.lib(), pkg._lib
    assert "libcloudy_hip_asan.so" in open("/proc/self/maps").read()
    inf, rng = float("inf"), np.random.default_rng(1)
    SRC = b"source-only"

    def tensor_desc(N, P, thr, moving=False, n_vel=0, dtype=0, types=None):
        kc = rng.uniform(0.0, 1.0, (N, N, P, P))
        kc = kc + kc.transpose(0, 1, 3, 2)
        kc = kc + kc.transpose(1, 0, 2, 3)
        vel = tuple((50.0 / (q + 1), (q + 1) / 6.0) for q in range(n_vel))
        return pkg.Plan.make_desc(types or [1] * N, kc, thr, bench.NORMS, 1 if moving else 0, vel=vel, dtype=dtype)

    # ---- tensor plans up to the limits: N = 8 / P = 8, thresholds on every mode but the last, four velocity terms, every plane type
    for N in (1, 2, 5, 8):
        for P in (1, 3, 8):
            for thr in ((inf,) * N, tuple(10.0 ** (-11 + i) for i in range(N - 1)) + (inf,)):
                for dtype in (0, 1, 2, 3):
                    d, keep = tensor_desc(N, P, thr, n_vel=4 if dtype == 0 else 1, dtype=dtype)
                    rc = L.cloudy_jit_selfcheck(C.byref(d), SRC)
                    assert rc == 0, (N, P, thr, dtype, L.cloudy_last_error())
    # MovingThreshold, mixed closures (Exponential / Gamma / Monodisperse / Lognormal)
    for N in (2, 4, 8):
        d, keep = tensor_desc(N, 2, tuple([0.97] * (N - 1) + [1.0]), moving=True, types=[(0, 1)[i % 2] for i in range(N - 1)] + [3])
        assert L.cloudy_jit_selfcheck(C.byref(d), SRC) == 0, L.cloudy_last_error()
    # (compute_threshold has no method for a Lognormal mode that is not the last: the refusal whose message read the plan after
    # deleting it -- the first finding of this build)
    d, keep = tensor_desc(4, 2, (0.9, 0.9, 0.9, 1.0), moving=True, types=[1, 3, 1, 1])
    assert L.cloudy_jit_selfcheck(C.byref(d), SRC) == E.EINVAL and b"dist_type[1] = 3" in L.cloudy_last_error()
    d, keep = tensor_desc(3, 2, (inf,) * 3, types=[2, 1, 3])
    assert L.cloudy_jit_selfcheck(C.byref(d), SRC) == 0, L.cloudy_last_error()
    # ---- NumericalCoalStyle plans of 1 ... 8 modes, both modes, every kernel function
    kfs = [pkg.ConstantKernelFunction(1e-4), pkg.LinearKernelFunction(5.0), pkg.HydrodynamicKernelFunction(3e2),
           pkg.LongKernelFunction(5.236e-10, 9.44e9, 5.78)]
    for N in (1, 2, 3, 4, 6, 8):
        for kf in kfs:
            for mode, q in ((pkg.QUAD_CONVERGED, 8), (pkg.QUAD_FIXED, 10), (pkg.QUAD_FIXED, 32)):
                types = [(1, 0, 3, 1)[i % 4] for i in range(N)]
                d = pkg.NumericalPlan.make_desc(types, kf, bench.NORMS, q, kernel_func_is_normalized=False, quad_mode=mode)
                assert L.cloudy_jit_selfcheck(C.byref(d), SRC) == 0, (N, mode, q, L.cloudy_last_error())
    # ---- every refusal of a descriptor the header names (status codes, never a crash)
    def expect(d, code):
        # (which of EINVAL / EUNSUPPORTED a limit answers with is the header's business and tests/test_host_abi.py's; here: a
        # refusal, the same from both entry points, a message, no handle -- and no sanitizer report on the way)
        h = C.c_void_p()
        rc = L.cloudy_plan_create(C.byref(d), C.byref(h))
        ok = (E.EINVAL, E.EUNSUPPORTED) if code in (E.EINVAL, E.EUNSUPPORTED) else (code,)
        assert rc in ok and not h.value and L.cloudy_last_error(), (rc, code, L.cloudy_last_error())
        assert L.cloudy_jit_selfcheck(C.byref(d), SRC) == rc

    d, keep = tensor_desc(2, 2, (inf, inf)); d.n_modes = 9; expect(d, E.EUNSUPPORTED)
    d, keep = tensor_desc(2, 2, (inf, inf)); d.n_modes = 0; expect(d, E.EUNSUPPORTED)
    d, keep = tensor_desc(2, 2, (inf, inf)); d.tensor_p = 9; expect(d, E.EUNSUPPORTED)
    d, keep = tensor_desc(2, 2, (inf, inf)); d.tensor_p = 0; expect(d, E.EUNSUPPORTED)
    d, keep = tensor_desc(2, 2, (inf, inf)); d.struct_size = 8; expect(d, E.EINVAL)
    d, keep = tensor_desc(2, 2, (inf, inf)); d.dist_type[1] = 7; expect(d, E.EINVAL)
    d, keep = tensor_desc(2, 2, (inf, inf)); d.norms[0] = 0.0; expect(d, E.EINVAL)
    d, keep = tensor_desc(2, 2, (inf, inf)); d.k_range[0], d.k_range[1] = 5.0, 1.0; expect(d, E.EINVAL)
    d, keep = tensor_desc(2, 2, (inf, inf)); d.n_vel = 5; expect(d, E.EINVAL)
    d, keep = tensor_desc(2, 2, (inf, inf)); d.dtype = 9; expect(d, E.EINVAL)
    d, keep = tensor_desc(2, 2, (inf, inf)); d.kernel_c = None; expect(d, E.EINVAL)
    d, keep = tensor_desc(2, 2, (float("nan"), inf)); expect(d, E.EINVAL)
    d, keep = tensor_desc(2, 2, (1.5, 1.0), moving=True); expect(d, E.EINVAL)          # a percentile above 1
    d, keep = tensor_desc(2, 2, (inf, inf)); keep_arr = np.ctypeslib.as_array(d.kernel_c, shape=(16,)); keep_arr[1] += 1.0
    expect(d, E.ENOTSYMMETRIC)
    dq = pkg.NumericalPlan.make_desc([1, 1], kfs[2], bench.NORMS, 10, kernel_func_is_normalized=False, quad_mode=pkg.QUAD_FIXED)
    dq.quad_order = 1000; expect(dq, E.EINVAL)
    dq = pkg.NumericalPlan.make_desc([1, 1], kfs[2], bench.NORMS, 10, kernel_func_is_normalized=False, quad_mode=pkg.QUAD_FIXED)
    dq.kernel_func = 11; expect(dq, E.EINVAL)
    dq = pkg.NumericalPlan.make_desc([1, 2], kfs[1], bench.NORMS, 10, kernel_func_is_normalized=False, quad_mode=pkg.QUAD_FIXED)
    expect(dq, E.EINVAL)                                                               # Monodisperse has no normed density
    assert L.cloudy_plan_create(None, None) == E.EINVAL and L.cloudy_jit_selfcheck(None, SRC) == E.EINVAL
    # the layout table, the host rule builder, the no-device answers of the memory helpers and of the communicator glue
    n = L.cloudy_plan_desc_layout(None, None, None, 0)
    names, offs, sizes = (C.c_char_p * n)(), (C.c_uint32 * n)(), (C.c_uint32 * n)()
    assert L.cloudy_plan_desc_layout(names, offs, sizes, n) == n
    for q in (2, 10, 32):
        u, W = np.zeros(q), np.zeros(q)
        for k in (1e-3, 0.7, 9.99):
            assert L.cloudy_quad_rule_host(q, 10.0, k, u.ctypes.data_as(C.POINTER(C.c_double)), W.ctypes.data_as(C.POINTER(C.c_double))) == 0
            assert abs(W.sum() - 1.0) < 1e-12
    assert L.cloudy_quad_rule_host(64, 10.0, 1.0, None, None) != 0
    buf = C.create_string_buffer(128)
    if pkg.device_count() == 0:
        assert L.cloudy_comm_unique_id(buf) == E.ENODEVICE
        assert L.cloudy_device_pci_bus_id(0, buf, 64) == E.ENODEVICE
        p = C.c_void_p()
        assert L.cloudy_malloc(C.byref(p), 64) != 0
        cnt = (C.c_uint64 * 8)()
        assert L.cloudy_closure_stats(None, 1, 1, None, cnt, None) == E.EINVAL
    assert L.cloudy_device_pci_bus_id(0, buf, 4) == E.EINVAL
    print("host sanitizer run ok")
""")


def test_host_code_of_the_library_under_asan_ubsan(tmp_path):
    env = _asan_env()
    script = tmp_path / "drv.py"
    script.write_text(DRIVER.format(root=ROOT))
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    _clean(p)
    assert "host sanitizer run ok" in p.stdout


def test_host_abi_suite_under_asan_ubsan():
    """tests/test_host_abi.py -- every CPU-side check of the boundary (exports, descriptor layout, error behaviour, the bench line
    assembly) -- once more against the sanitized build"""
    env = _asan_env()
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_abi.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    _clean(p)
    assert " passed" in p.stdout
