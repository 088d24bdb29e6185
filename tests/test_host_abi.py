"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol include/cloudy_hip.h
declares; host-side logic (index maps, CoalescenceData derived fields, tensor checks, error behaviour)
matches the reference KATs and the oracle; no compute call is made (there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INF = float("inf")


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "cloudy_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cloudy_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(cloudy):
    L = cloudy.lib()
    names = _header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/cloudy_hip.h but not exported"
    assert sorted(cloudy._lib.SYMBOLS) == names, "ctypes table and header disagree"
    assert L.cloudy_version() == 100


def test_plan_desc_layout_and_defaults(cloudy):
    d = cloudy._lib.PlanDesc()
    cloudy.lib().cloudy_plan_desc_init(C.byref(d))
    assert d.struct_size == C.sizeof(cloudy._lib.PlanDesc)
    assert (d.k_range[0], d.k_range[1]) == (np.finfo(np.float64).eps, 10.0)
    assert d.n_bins_per_log_unit == 15 and d.device == -1
    assert all(np.isinf(d.dist_thresholds[i]) for i in range(4))


def test_plan_desc_layout_matches_every_binding(cloudy):
    """cloudy_plan_desc_layout() (offsets / sizes as the library was compiled) against the two bindings that mirror the
    struct by hand: the ctypes PlanDesc and `mutable struct PlanDesc` of julia/CloudyHIP.jl (parsed here -- julia is not
    installed, so the shim cannot be run; its field order, types and hence C layout can still be checked)."""
    import re

    L = cloudy.lib()
    cap = 64
    names, offs, sizes = (C.c_char_p * cap)(), (C.c_uint32 * cap)(), (C.c_uint32 * cap)()
    nf = L.cloudy_plan_desc_layout(names, offs, sizes, cap)
    lib_fields = [(names[i].decode(), offs[i], sizes[i]) for i in range(nf)]
    # ctypes mirror
    PD = cloudy._lib.PlanDesc
    assert [f[0] for f in PD._fields_] == [f[0] for f in lib_fields]
    for name, off, size in lib_fields:
        fld = getattr(PD, name)
        assert (fld.offset, fld.size) == (off, size), name
    assert lib_fields[-1][1] + lib_fields[-1][2] <= C.sizeof(PD) < lib_fields[-1][1] + lib_fields[-1][2] + 8
    # Julia mirror: parse the struct, lay it out with C rules (natural alignment), compare
    src = open(os.path.join(ROOT, "julia", "CloudyHIP.jl")).read()
    consts = {k: int(v) for k, v in zip(("MAX_MODES", "MAX_P", "MAX_VEL"),
                                        re.search(r"const MAX_MODES, MAX_P, MAX_VEL = (\d+), (\d+), (\d+)", src).groups())}
    body = re.search(r"mutable struct PlanDesc\n(.*?)\nend", src, re.S).group(1)
    prim = {"UInt32": 4, "Int32": 4, "Float64": 8, "Ptr{Float64}": 8}
    off, jl = 0, []
    for line in body.splitlines():
        line = line.split("#")[0].strip()
        if not line:
            continue
        name, typ = [x.strip() for x in line.split("::")]
        m = re.fullmatch(r"NTuple\{(.+),\s*(\w+)\}", typ)
        if m:
            count = eval(m.group(1).replace("2 * MAX_VEL", str(2 * consts["MAX_VEL"])), {}, consts)
            esz = prim[m.group(2)]
            size = count * esz
        else:
            esz = size = prim[typ]
        off = (off + esz - 1) // esz * esz
        jl.append((name, off, size))
        off += size
    assert jl == lib_fields
    assert (consts["MAX_MODES"], consts["MAX_P"], consts["MAX_VEL"]) == (cloudy._lib.MAX_MODES, cloudy._lib.MAX_P, cloudy._lib.MAX_VEL)
    # every ccall in the shim names an exported symbol
    for sym in set(re.findall(r"ccall\(\(:(\w+), lib\)", src)):
        assert hasattr(L, sym), sym


def _c_prototypes():
    """{name: (return C type, [argument C types])} of every function include/cloudy_hip.h declares"""
    import re

    text = open(os.path.join(ROOT, "include", "cloudy_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = text[text.index("typedef struct cloudy_plan cloudy_plan;"):]
    text = re.sub(r"typedef struct cloudy_plan_desc \{.*?\} cloudy_plan_desc;", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(cloudy_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), m.group(3).strip()
        if "typedef" in ret or "define" in ret:
            continue
        argt = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                argt.append("ptr" if "*" in a else a.rsplit(" ", 1)[0])   # drop the parameter name
        protos[name] = ("ptr" if "*" in ret else ret, argt)
    return protos


def _julia_ccalls(src):
    """[(symbol, return type, [argument types])] of every ccall((:sym, lib), ...) in the shim"""
    import re

    out = []
    for m in re.finditer(r"ccall\(\(:(\w+), lib\),\s*", src):
        i = m.end()
        j = src.index(",", i)
        ret = src[i:j].strip()
        k = src.index("(", j)                     # the argument-type tuple
        depth, e = 0, k
        while True:
            depth += src[e] == "("
            depth -= src[e] == ")"
            if depth == 0:
                break
            e += 1
        inner = src[k + 1:e]
        args, cur, br = [], "", 0
        for ch in inner:
            br += ch == "{"
            br -= ch == "}"
            if ch == "," and br == 0:
                args.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            args.append(cur.strip())
        out.append((m.group(1), ret, args))
    return out


def test_julia_shim_ccalls_match_the_header_and_drop_in_branches_exist(cloudy):
    """julia is not installed, so the shim cannot run; what a parser can check: (i) every ccall's argument count and C
    types against the prototype in include/cloudy_hip.h, (ii) the branches that make the reference drivers run unchanged
    (VERDICT r2 weak #4): Vector state, host arrays through cloudy_coal_rhs_host, stride checks, a lazily built and cached
    plan from `par`, make_rainshaft_rhs's 3-argument rhs, the specialisation warning, the RCCL entry points."""
    import re

    src = open(os.path.join(ROOT, "julia", "CloudyHIP.jl")).read()
    protos = _c_prototypes()
    assert len(protos) >= 50 and set(protos) == set(cloudy._lib.SYMBOLS), set(protos) ^ set(cloudy._lib.SYMBOLS)
    jl_of_c = {"int": {"Cint"}, "size_t": {"Csize_t"}, "double": {"Cdouble"}, "void": {"Cvoid"}, "float": {"Cfloat"},
               "uint32_t": {"UInt32", "Cuint"}, "int32_t": {"Int32", "Cint"}}
    calls = _julia_ccalls(src)
    assert len(calls) >= 25
    for sym, ret, args in calls:
        assert sym in protos, f"ccall of {sym}: not declared in include/cloudy_hip.h"
        cret, cargs = protos[sym]
        assert len(args) == len(cargs), f"{sym}: {len(args)} ccall arguments, header has {len(cargs)}"
        if cret == "ptr":
            assert ret in ("Cstring", "Ptr{Cvoid}"), (sym, ret)
        else:
            assert ret in jl_of_c[cret], (sym, ret, cret)
        for a, c in zip(args, cargs):
            if c == "ptr":
                assert re.match(r"(Ptr|Ref)\{.+\}$", a) or a == "Cstring", (sym, a)
            else:
                assert a in jl_of_c[c.replace("const ", "")], (sym, a, c)
    used = {c[0] for c in calls}
    for need in ("cloudy_coal_rhs", "cloudy_coal_rhs_host", "cloudy_rainshaft_rhs", "cloudy_plan_specialized",
                 "cloudy_plan_jit_log", "cloudy_plan_nmom", "cloudy_ssprk33_steps", "cloudy_rainshaft_ssprk33_steps",
                 "cloudy_comm_unique_id", "cloudy_comm_create", "cloudy_moment_sums_allreduce", "cloudy_plan_destroy"):
        assert need in used, need
    # the factory keeps the reference's positional signature and needs no plan keyword
    assert re.search(r"function make_box_model_rhs\(coal_type::CoalescenceStyle, ts::ThresholdStyle = FixedThreshold\(\); "
                     r"plan = nothing", src)
    assert "plan === nothing ? cached_plan!(cache, coal_type, ts, par) : plan" in src
    assert re.search(r"function plan_from_parameters\(.*\bpar\)", src) and "par.coal_data" in src and "par.kernel_func" in src
    assert "cd.kernels" in src and "cd.dist_thresholds" in src and "normalized = true" in src
    # a Vector state is ONE parcel with leading dimension 1; matrices must have unit stride and matching dm
    assert re.search(r"function batch_shape\(m::AbstractVector, nmom\)", src) and "return 1, 1" in src
    assert "batch_shape(dm, p.nmom) || error" in src and "stride(m, 1) == 1 || error" in src
    assert "is_host(m) && is_host(dm)" in src
    # make_rainshaft_rhs: out-of-place rhs(m, p, t), negatives clamped in place first (rainshaft_helpers.jl:52)
    assert re.search(r"function make_rainshaft_rhs\(coal_type::CoalescenceStyle;", src)
    assert re.search(r"function rhs\(m, p, t\)", src) and "m .= max.(m, zero(eltype(m)))" in src and "p.dz" in src
    assert "@warn" in src and "p.specialized ||" in src
    # the stream of a device array is AMDGPU.jl's task-local one, found among the LOADED modules (not
    # `isdefined(Main, :AMDGPU)`, false when the host loads AMDGPU.jl inside its own package); no stream -> an error
    code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    assert "Base.loaded_modules" in code and "isdefined(Main, :AMDGPU)" not in code
    assert re.search(r"mod === nothing && error\(", code) and "return C_NULL\nend" not in code
    # the drop-in's NumericalCoalStyle plan is the converged mode unless the host asks for the fixed rule
    assert re.search(r"function numerical_plan\(pdists, kernel_func_normalized, norms; quad_order = 0, quad_mode = 1\)", code)
    assert "return numerical_plan(par.pdists, par.kernel_func, par.norms)" in code


def test_thresholds_are_normalized_flag_reproduces_the_coalescence_data_fields(cloudy):
    """a host that builds the plan from an existing CoalescenceData passes its (already normalised) kernels and
    dist_thresholds through unchanged: same derived plan as from the constructor arguments (no device needed: the
    validation / derivation half of cloudy_plan_create is what cloudy_jit_selfcheck runs too)"""
    L = cloudy.lib()
    norms = (1e6, 1e-9)
    c = np.array([[0.0, 5.78], [5.78, 0.0]])
    cn = c * norms[0] * norms[1] ** np.add.outer(np.arange(2), np.arange(2))

    def desc(kernel, thr, flag):
        d = cloudy._lib.PlanDesc()
        L.cloudy_plan_desc_init(C.byref(d))
        d.n_modes, d.tensor_p = 2, 2
        d.dist_type[0] = d.dist_type[1] = 1
        d.kernel_c = kernel.ctypes.data_as(C.POINTER(C.c_double))
        d.kernel_is_normalized = d.thresholds_are_normalized = flag
        d.norms[0], d.norms[1] = norms
        d.dist_thresholds[0] = thr
        return d

    for d in (desc(c, 5e-10, 0), desc(cn, 5e-10 / 1e-9, 1)):
        assert L.cloudy_jit_selfcheck(C.byref(d), b"gfx950") == 0, L.cloudy_last_error()
    if cloudy.device_count():
        outs = []
        for d in (desc(c, 5e-10, 0), desc(cn, 5e-10 / 1e-9, 1)):
            h = C.c_void_p()
            cloudy._lib.check(L.cloudy_plan_create(C.byref(d), C.byref(h)))
            thr, kc = (C.c_double * 2)(), (C.c_double * 16)()
            cloudy._lib.check(L.cloudy_plan_get(h, None, None, thr, kc, None))
            outs.append((list(thr), list(kc)))
            L.cloudy_plan_destroy(h)
        assert outs[0] == outs[1]


def test_moment_sums_workspace_size(cloudy):
    L = cloudy.lib()
    assert L.cloudy_moment_sums_workspace_bytes(6) == 8 * 1024 * 6 and L.cloudy_moment_sums_workspace_bytes(0) == 0


def test_plan_validation_errors_precede_device_lookup(cloudy):
    L = cloudy.lib()

    def create(mut):
        d = cloudy._lib.PlanDesc()
        L.cloudy_plan_desc_init(C.byref(d))
        c = np.array([[0.0, 5.0], [5.0, 0.0]])
        d.n_modes, d.tensor_p = 1, 2
        d.dist_type[0] = 1
        d.kernel_c = c.ctypes.data_as(C.POINTER(C.c_double))
        mut(d, c)
        h = C.c_void_p()
        rc = L.cloudy_plan_create(C.byref(d), C.byref(h))
        msg = L.cloudy_last_error().decode()
        if rc == 0:
            L.cloudy_plan_destroy(h)
        return rc, msg

    E = cloudy._lib
    rc, msg = create(lambda d, c: c.__setitem__((0, 1), 4.0))
    assert rc == E.ENOTSYMMETRIC and "not symmetric" in msg          # KernelTensors.jl:157-171
    rc, msg = create(lambda d, c: d.norms.__setitem__(0, 0.0))
    assert rc == E.EINVAL and "norms must be positive" in msg         # helper_functions.jl:44-46
    rc, _ = create(lambda d, c: setattr(d, "n_modes", 9))
    assert rc == E.EUNSUPPORTED
    rc, _ = create(lambda d, c: setattr(d, "tensor_p", 9))
    assert rc == E.EUNSUPPORTED
    rc, _ = create(lambda d, c: d.dist_type.__setitem__(0, 9))
    assert rc == E.EINVAL
    rc, _ = create(lambda d, c: setattr(d, "struct_size", 8))
    assert rc == E.EINVAL
    rc, _ = create(lambda d, c: setattr(d, "dtype", 7))
    assert rc == E.EINVAL
    if cloudy.device_count() == 0:
        rc, msg = create(lambda d, c: None)
        assert rc == E.ENODEVICE and "no HIP device" in msg           # fails loudly, no CPU fallback


def test_no_cpu_fallback_and_no_oracle_in_product(cloudy):
    """The product package never imports the oracle, and batched calls without a GPU raise."""
    pkg_dir = os.path.join(ROOT, "cloudy.jl_amd")
    pat = re.compile(r"^\s*(import|from)\s+\S*oracle|cloudy_oracle|libcloudy_oracle|co_rhs_coal", re.M)
    for fn in os.listdir(pkg_dir):
        if fn.endswith(".py"):
            assert not pat.search(open(os.path.join(pkg_dir, fn)).read()), fn
    for fn in os.listdir(os.path.join(pkg_dir, "csrc")):
        if fn.endswith((".hip", ".hpp", ".h", "Makefile")):
            assert not pat.search(open(os.path.join(pkg_dir, "csrc", fn)).read()), fn
    if cloudy.device_count() == 0:
        cd = cloudy.CoalescenceData(cloudy.CoalescenceTensor([[1.0]]), (3,), (INF,))
        with pytest.raises(cloudy.CloudyError) as e:
            cd.plan([1])
        assert e.value.code == cloudy._lib.ENODEVICE


def test_helper_function_kats(cloudy, kats):
    e = kats["helper_functions"]
    npm = tuple(e["NProgMoms"])
    for i, m, ex in e["moment_ind"]:
        assert cloudy.get_dist_moment_ind(npm, i, m) == ex
    for i, m in e["moment_ind_throws"]:
        with pytest.raises((ValueError, IndexError)):
            cloudy.get_dist_moment_ind(npm, i, m)
    for i, a, b in e["ind_range"]:
        assert cloudy.get_dist_moments_ind_range(npm, i) == range(a, b + 1)
    with pytest.raises(IndexError):
        cloudy.get_dist_moments_ind_range(npm, 4)
    assert np.allclose(cloudy.get_moments_normalizing_factors(npm, e["norms"]), e["norm_factors"], atol=e["atol"], rtol=0)
    assert cloudy.rflatten(((1, 2), (3.2, (1.2, 1.0)), (1,))) == (1, 2, 3.2, 1.2, 1.0, 1)


def test_kernel_tensor_and_function_kats(cloudy, kats, oracle):
    e = kats["normalized_kernel_tensor"]
    kn = cloudy.get_normalized_kernel_tensor(cloudy.CoalescenceTensor(e["c"]), e["norms"])
    assert np.allclose(kn.c, e["expected"], atol=e["atol"], rtol=0)
    assert np.array_equal(kn.c, oracle.get_normalized_kernel_tensor(e["c"], e["norms"]))
    for c in kats["check_symmetry"]["ok"]:
        cloudy.check_symmetry(c)
    for c in kats["check_symmetry"]["throws"]:
        with pytest.raises(ValueError):
            cloudy.CoalescenceTensor(c)
    kn = kats["kernel_normalization"]
    norms = kn["norms"]
    assert cloudy.get_normalized_kernel_func(cloudy.ConstantKernelFunction(1.0), norms).coll_coal_rate == 100.0
    assert cloudy.get_normalized_kernel_func(cloudy.LinearKernelFunction(1.0), norms).coll_coal_rate == 1.0 * 100.0 * 0.001
    h = cloudy.get_normalized_kernel_func(cloudy.HydrodynamicKernelFunction(1.0), norms)
    assert abs(h.coal_eff - 100.0 * 0.001 ** (4.0 / 3)) < 1e-12
    lg = cloudy.get_normalized_kernel_func(cloudy.LongKernelFunction(1.0, 10.0, 5.0), norms)
    assert (lg.x_threshold, lg.coal_rate_below_threshold, lg.coal_rate_above_threshold) == (
        1.0 / 0.001, 10.0 * 100.0 * 0.001**2, 5.0 * 100.0 * 0.001)
    for e in kats["kernel_functions"]:
        cls = {"constant": cloudy.ConstantKernelFunction, "linear": cloudy.LinearKernelFunction,
               "hydrodynamic": cloudy.HydrodynamicKernelFunction, "long": cloudy.LongKernelFunction}[e["kind"]]
        k = cls(*e["params"])
        if "expected" in e:
            assert k(e["x"], e["y"]) == e["expected"]
        assert k(e["x"], e["y"]) == k(e["y"], e["x"])
    # exact tensors of polynomial kernels, with the forced C_1_1 = max(eps, K(0,0)) of KernelTensors.jl:115
    eps = np.finfo(np.float64).eps
    # (the constant is forced in normalised units, default norms (1e6, 1e-9): eps / 1e6 after de-normalisation)
    t = cloudy.CoalescenceTensor(cloudy.LinearKernelFunction(5e-3), 1, 10.0)
    assert np.array_equal(t.c, [[eps / 1e6, 5e-3], [5e-3, 0.0]])
    lk = cloudy.LongKernelFunction(5.236e-10, 9.44e9, 5.78)   # box_gamma_mixture_long.jl:20-30
    e6 = eps / 1e6
    assert np.array_equal(cloudy.CoalescenceTensor(lk, 2, 5e-10).c, [[e6, 0, 9.44e9], [0, 0, 0], [9.44e9, 0, 0]])
    assert np.array_equal(cloudy.CoalescenceTensor(lk, 2, 1e-6, 5e-10).c, [[e6, 5.78, 0], [5.78, 0, 0], [0, 0, 0]])
    assert cloudy.CoalescenceTensor(cloudy.ConstantKernelFunction(1.0), 0, 100.0).c[0, 0] == 1.0


def test_coalescence_data_fields_match_oracle(cloudy, oracle):
    rng = np.random.default_rng(5)
    for N, P, npm in [(1, 1, (3,)), (2, 2, (3, 2)), (3, 3, (2, 2, 3)), (4, 5, (3, 3, 3, 3)), (2, 3, (2, 2))]:
        kc = rng.uniform(0, 1, (N, N, P, P))
        kc = kc + kc.transpose(0, 1, 3, 2)
        thr = tuple([0.5e-9 * (i + 1) for i in range(N - 1)] + [INF])
        kernels = tuple(tuple(cloudy.CoalescenceTensor(kc[j, k]) for k in range(N)) for j in range(N))
        cd = cloudy.CoalescenceData(kernels, npm, thr, (1e6, 1e-9))
        ocd = oracle.coalescence_data(kc, npm, thr, (1e6, 1e-9))
        assert cd.N_mom_max == ocd.N_mom_max
        assert list(cd.N_2d_ints) == [ocd.N_2d_ints[i] for i in range(N)]
        assert np.array_equal(cd.dist_thresholds, [ocd.dist_thresholds[i] for i in range(N)])
    with pytest.raises(ValueError):
        cloudy.CoalescenceData(cloudy.CoalescenceTensor([[1.0]]), (3, 3), (INF,))


def test_distribution_constructors(cloudy):
    G, E = cloudy.GammaPrimitiveParticleDistribution, cloudy.ExponentialPrimitiveParticleDistribution
    for bad in ((-1.0, 2.0, 3.0), (1.0, -2.0, 3.0), (1.0, 2.0, -3.0)):
        with pytest.raises(ValueError):
            G(*bad)
    for bad in ((-1.0, 2.0), (1.0, -2.0)):
        with pytest.raises(ValueError):
            E(*bad)
    assert cloudy.nparams(G(1.0, 1.0, 2.0)) == 3 and cloudy.nparams(E(1.0, 1.0)) == 2
    assert cloudy.get_moments(G(1.0, 1.0, 2.0)) == [1.0, 2.0, 6.0]
    assert cloudy.get_moments(E(1.0, 2.0)) == [1.0, 2.0]


def test_box_model_rhs_factory_signature(cloudy):
    rhs = cloudy.make_box_model_rhs(cloudy.AnalyticalCoalStyle())
    import inspect

    assert list(inspect.signature(rhs).parameters) == ["dm", "m", "par", "t"]  # rhs!(dm, m, par, t)
    with pytest.raises(TypeError):
        cloudy.make_box_model_rhs("analytical")
    assert isinstance(cloudy.AnalyticalCoalStyle(), cloudy.CoalescenceStyle)
    assert not isinstance(cloudy.AnalyticalCoalStyle(), cloudy.NumericalCoalStyle)


def test_synthetic_workload_is_deterministic_and_degenerate_cases_present():
    import bench

    a = bench.synth_moments(2, 5000, seed=11)
    b = bench.synth_moments(2, 5000, seed=11)
    assert np.array_equal(a, b) and a.shape == (6, 5000)
    assert (a[0] == 0).sum() > 0                      # empty parcels
    var = a[2] / a[1] - a[1] / a[0]
    with np.errstate(invalid="ignore", divide="ignore"):
        assert (var < 0).sum() > 0 and (np.abs(var) < 1e-25).sum() > 0


def test_polyfit_reference_kats(cloudy):
    """test_KernelTensors_correctness.jl:17-20, 36-46 (rtol 1e-5 there: Nelder-Mead; the direct least-squares solve
    here is exact for polynomial targets)."""
    t = cloudy.CoalescenceTensor(lambda x, y: 0.02 + x + y, 1, 10.0)
    assert np.allclose(t.c, [[0.02, 1.0], [1.0, 0.0]], rtol=1e-5, atol=1e-12)
    assert np.allclose(cloudy.polyfit(lambda x, y: 0.1 + 0.2 * x * y, 1, 10.0), [[0.1, 0.0], [0.0, 0.2]], rtol=1e-5, atol=1e-12)
    f = lambda x, y: 0.1 - 0.23 * x - 0.23 * y + 0.2 * x * y  # noqa: E731
    for lim in (10.0, 100.0, 1000.0):
        assert np.allclose(cloudy.polyfit(f, 1, lim), [[0.1, -0.23], [-0.23, 0.2]], rtol=1e-5, atol=0)
    with pytest.raises(ValueError):
        cloudy.polyfit(lambda x, y: x - y, 1, 10.0)        # "function likely not symmetric."
    with pytest.raises(ValueError):
        cloudy.polyfit(f, 1, 1.0, 2.0)                      # limits improperly specified
    # type-stability KAT (:56-61): a linear kernel fitted at order 1 is a 2 x 2 symmetric tensor
    t = cloudy.CoalescenceTensor(cloudy.LinearKernelFunction(5.0), 1, 5e-10)
    assert t.c.shape == (2, 2) and t.c[0, 1] == t.c[1, 0] == 5.0
    # the hydrodynamic kernel of box_gamma_mixture_hydro.jl:22-23 (order 4): symmetric, C_1_1 = eps / n0
    h = cloudy.CoalescenceTensor(cloudy.HydrodynamicKernelFunction(1e2 * np.pi), 4, 1e-6)
    assert h.c.shape == (5, 5) and np.array_equal(h.c, h.c.T)
    assert h.c[0, 0] == np.finfo(np.float64).eps / 1e6


@pytest.mark.parametrize("case", ["cfg2", "cfg3a", "cfg3a_f32", "cfg3b", "cfg3b_f32fast", "cfg3_moving", "cfg4",
                                  "n4p5_mixed", "n6p2_beyond_aot", "n5p3_fixed_beyond_aot", "n2p8_fixed_beyond_aot",
                                  "n6p2_moving_beyond_aot"])
def test_plan_time_translation_units_compile_without_a_gpu(cloudy, case):
    """cloudy_jit_selfcheck: the kernel sources embedded in libcloudy_hip.so plus the generated constexpr plan compile
    with hiprtc for gfx950 here, on the CPU -- the build check of the plan-time specialisation (jit.hpp).  Thresholded
    plans compile two units (single-pass kernel; fused integrator)."""
    import bench

    L = cloudy.lib()
    moving, dtype = 0, 0
    if case == "n4p5_mixed":
        rng = np.random.default_rng(7)
        N, P = 4, 5
        kc = np.zeros((N, N, P, P))
        for j in range(N):
            for k in range(j, N):
                c = rng.uniform(0.1, 1.0, (P, P)) * (rng.random((P, P)) < 0.5)
                kc[j, k] = kc[k, j] = np.triu(c) + np.triu(c, 1).T
        dist_types, thr, dtype = [0, 1, 2, 1], (1e-9, 1e-8, 1e-7, INF), 1
    elif case.endswith("beyond_aot"):   # more than CLOUDY_AOT_MAX_MODES modes / order > 4: plan-time compiled kernels only
        N, P, thr = {"n6p2_beyond_aot": (6, 2, (INF,) * 6), "n5p3_fixed_beyond_aot": (5, 3, (1e-10, 1e-9, 1e-8, 1e-7, INF)),
                     "n2p8_fixed_beyond_aot": (2, 8, (5e-10, INF)), "n6p2_moving_beyond_aot": (6, 2, (0.9,) * 5 + (1.0,))}[case]
        rng = np.random.default_rng(N * 10 + P)
        kc = np.zeros((N, N, P, P))
        for j in range(N):
            for k in range(j, N):
                c = rng.uniform(0.1, 1.0, (P, P)) * (rng.random((P, P)) < 0.6)
                kc[j, k] = kc[k, j] = np.triu(c) + np.triu(c, 1).T
        dist_types, moving = [1] * N, int("moving" in case)
    else:
        name = {"cfg3a_f32": "cfg3a", "cfg3b_f32fast": "cfg3b", "cfg3_moving": "cfg3b"}.get(case, case)
        spec = bench.workload_spec(name)
        kc, dist_types, thr = bench.kernel_matrix(spec), [1] * spec["n_modes"], spec["thresholds"]
        dtype = {"cfg3a_f32": 1, "cfg3b_f32fast": 2}.get(case, 0)
        if case == "cfg3_moving":
            moving, thr = 1, (0.9, 1.0)
    d, keep = cloudy.Plan.make_desc(dist_types, kc, thr, bench.NORMS, moving, dtype=dtype)
    rc = L.cloudy_jit_selfcheck(C.byref(d), b"gfx950")
    assert rc == 0, L.cloudy_last_error().decode()
    assert keep is not None


def test_jit_selfcheck_reports_descriptor_errors(cloudy):
    L = cloudy.lib()
    d, keep = cloudy.Plan.make_desc([1], [[1.0, 2.0], [3.0, 4.0]], (INF,), (1.0, 1.0), 0)
    assert L.cloudy_jit_selfcheck(C.byref(d), None) == cloudy._lib.ENOTSYMMETRIC
    assert b"not symmetric" in L.cloudy_last_error()


def _strict_loads(text):
    """json.loads that rejects NaN / Infinity (what a strict parser on the driver's side does)"""
    import json

    def bad(name):
        raise ValueError("non-standard JSON constant " + name)

    return json.loads(text, parse_constant=bad)


def test_bench_contract_line_is_compact_strict_json():
    """BENCH_r04.parsed was null because bench.py printed one 20 KB line.  The LAST stdout line is now
    assembled by bench.compact_line from the full result object: < 4 KB, strict JSON, headline + roofline + cpu_baseline +
    per variant {value, kernel_ms, frac}.  Canned input: round 4's full 20 KB object (profiles/r04_bench.json), then the
    same with 8 ranks, NaN / Inf values and three dozen variants."""
    import copy
    import json

    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    assert len(json.dumps(full)) > 15000
    line = bench.compact_line(full)
    assert len(line) < 4096 and "\n" not in line
    out = _strict_loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "collective", "variants", "variants_file"):
        assert key in out, key
    assert out["value"] == pytest.approx(full["value"], rel=1e-6) and out["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-6)
    assert out["config"]["workload"].startswith("cfg3a: 10000000 parcels/GPU")
    rl = out["roofline"]
    assert rl["bound"] == "hbm" and rl["unit"] == "GB/s" and rl["peak"] == 8000.0 and rl["kernel"] == "cloudy_jit_allinf2_n2p3_f64"
    assert rl["frac"] == pytest.approx(rl["achieved"] / rl["peak"], rel=1e-5) and rl["traffic"] > 9e8 and rl["kernel_ms"] > 0
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 16 and cb["value"] > 0 and cb["unit"] == "parcel-RHS/s" and cb["sample"]
    assert set(out["variants"]) == set(full["variants"])
    for name, v in out["variants"].items():
        assert set(v) <= {"value", "kernel_ms", "frac", "unit"} and v["value"] > 0, name
    assert out["variants"]["cfg4q_converged"]["frac"] == pytest.approx(full["variants"]["cfg4q_converged"]["roofline"]["frac"], rel=1e-3)
    # a multi-rank line: one kernel_ms per rank; non-finite numbers become null; a flood of variants is dropped, not printed
    big = copy.deepcopy(full)
    big["n_gpus"] = 8
    big["roofline"]["per_rank"] = [{"rank": r, "kernel_ms": 0.15 + 0.001 * r, "achieved": 6000.0, "frac": 0.75} for r in range(8)]
    big["mass_rate_residual"] = float("nan")
    big["variants"]["cfg2"]["value"] = float("inf")
    big["collective_fallback"] = True
    line8 = bench.compact_line(big)
    out8 = _strict_loads(line8)
    assert len(line8) < 4096 and out8["roofline"]["per_rank_kernel_ms"] == [pytest.approx(0.15 + 0.001 * r) for r in range(8)]
    assert out8["mass_rate_residual"] is None and out8["variants"]["cfg2"]["value"] is None and out8["collective_fallback"] is True
    for i in range(40):
        big["variants"][f"extra_variant_with_a_long_name_{i:02d}"] = {"value": 1.0 * i, "kernel_ms": 2.0, "roofline": {"frac": 0.5}}
    line_big = bench.compact_line(big)
    out_big = _strict_loads(line_big)
    assert len(line_big) < 4096 and "variants" not in out_big and out_big["roofline"]["frac"] > 0 and out_big["cpu_baseline"]
    # the one-process form (no roofline block in its object) goes through the same assembly
    one = {"metric": full["metric"], "value": 1e10, "unit": "parcel-RHS/s", "n_gpus": 2, "steps": 5, "warmup": 2, "ms_per_step": 0.3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "cfg3a: 300000 parcels/GPU", "global_parcels": 600000, "launch": "ONE process driving all GPUs"},
           "collective": "ncclAllReduce", "mass_rate_residual": 1e-17}
    o1 = _strict_loads(bench.compact_line(one))
    assert o1["config"]["launch"].startswith("ONE process") and o1["cpu_baseline"] is None and o1["variants"] == {}
