# CloudyHIP.jl -- ccall binding of libcloudy_hip.so (include/cloudy_hip.h) for Cloudy.jl.
# SOURCE ONLY: julia is not installed in the build image or on the GPU box, so this file has never been run;
# the same C ABI is exercised from Python/ctypes (cloudy.jl_amd/_lib.py, tests/test_gpu_parity.py), and the field
# order / types of `PlanDesc` below are checked against the library's own cloudy_plan_desc_layout() by
# tests/test_host_abi.py::test_plan_desc_layout_matches_every_binding (a parser of this file, no Julia needed);
# `check_layout()` does the same check at load time once Julia runs it.
# See INTEGRATION.md.
module CloudyHIP

using Cloudy, Cloudy.Coalescence, Cloudy.ParticleDistributions, Cloudy.EquationTypes, Cloudy.KernelTensors,
      Cloudy.KernelFunctions

const lib = get(ENV, "CLOUDY_HIP_LIB", joinpath(@__DIR__, "..", "cloudy.jl_amd", "libcloudy_hip.so"))

const MAX_MODES, MAX_P, MAX_VEL = 4, 5, 4

# mirrors `struct cloudy_plan_desc` of include/cloudy_hip.h field by field
mutable struct PlanDesc
    struct_size::UInt32
    n_modes::Int32
    dist_type::NTuple{MAX_MODES,Int32}
    tensor_p::Int32
    kernel_layout::Int32            # 1 = [N][N][P][P]
    kernel_is_normalized::Int32
    kernel_c::Ptr{Float64}
    dist_thresholds::NTuple{MAX_MODES,Float64}
    threshold_style::Int32          # 0 FixedThreshold, 1 MovingThreshold
    norms::NTuple{2,Float64}
    k_range::NTuple{2,Float64}
    n_bins_per_log_unit::Int32
    dtype::Int32
    n_vel::Int32
    vel::NTuple{2 * MAX_VEL,Float64}
    device::Int32
    specialize::Int32               # plan-time compiled kernels for all-Inf thresholds: 0 auto, 1 required, -1 off
    coal_style::Int32               # 0 AnalyticalCoalStyle, 1 NumericalCoalStyle (fixed Gauss rule, csrc/quad.hpp)
    kernel_func::Int32              # 0 Constant, 1 Linear, 2 Hydrodynamic, 3 Long (KernelFunctions.jl:39-86)
    kernel_func_is_normalized::Int32
    quad_order::Int32
    kernel_func_params::NTuple{3,Float64}
end

"""
    check_layout()

Compares `fieldoffset` / `sizeof` of every `PlanDesc` field with what the loaded library reports for its
`cloudy_plan_desc` (cloudy_plan_desc_layout): a binding compiled against another header version fails here, not in a kernel.
"""
function check_layout()
    nf = fieldcount(PlanDesc)
    names = Vector{Cstring}(undef, nf); offs = Vector{UInt32}(undef, nf); sizes = Vector{UInt32}(undef, nf)
    n = ccall((:cloudy_plan_desc_layout, lib), Cint, (Ptr{Cstring}, Ptr{UInt32}, Ptr{UInt32}, Cint), names, offs, sizes, nf)
    n == nf || error("cloudy_plan_desc has $n fields, PlanDesc has $nf")
    for i in 1:nf
        (String(fieldname(PlanDesc, i)) == unsafe_string(names[i]) && fieldoffset(PlanDesc, i) == offs[i] &&
         sizeof(fieldtype(PlanDesc, i)) == sizes[i]) || error("PlanDesc field $i ($(fieldname(PlanDesc, i))) does not match the library")
    end
    return nothing
end

check(rc) = rc == 0 ? nothing : error("libcloudy_hip: ", unsafe_string(ccall((:cloudy_last_error, lib), Cstring, ())))

dist_code(::ExponentialPrimitiveParticleDistribution) = Int32(0)
dist_code(::GammaPrimitiveParticleDistribution) = Int32(1)
dist_code(::MonodispersePrimitiveParticleDistribution) = Int32(2)
dist_code(::LognormalPrimitiveParticleDistribution) = Int32(3)

pad(t, n, z) = ntuple(i -> i <= length(t) ? t[i] : z, n)

new_desc() = (d = Ref(PlanDesc(0, 0, pad((), MAX_MODES, Int32(0)), 0, 0, 0, C_NULL, pad((), MAX_MODES, Inf), 0,
                               (1.0, 1.0), (eps(Float64), 10.0), 15, 0, 0, pad((), 2MAX_VEL, 0.0), -1, 0,
                               0, 0, 0, 10, (0.0, 0.0, 0.0)));
              ccall((:cloudy_plan_desc_init, lib), Cvoid, (Ref{PlanDesc},), d); d)

kernel_func_code(k::ConstantKernelFunction) = (Int32(0), (k.coll_coal_rate, 0.0, 0.0))
kernel_func_code(k::LinearKernelFunction) = (Int32(1), (k.coll_coal_rate, 0.0, 0.0))
kernel_func_code(k::HydrodynamicKernelFunction) = (Int32(2), (k.coal_eff, 0.0, 0.0))
kernel_func_code(k::LongKernelFunction) =
    (Int32(3), (k.x_threshold, k.coal_rate_below_threshold, k.coal_rate_above_threshold))

"""
    numerical_plan(pdists, kernel_func_normalized, norms; quad_order = 10)

Plan of `make_box_model_rhs(NumericalCoalStyle())`: `kernel_func_normalized` is what the drivers put in
`p.kernel_func` (`get_normalized_kernel_func(kernel, norms)`, test/examples/Numerical/n_particles_gamma.jl:35).
The integrals of src/Sources/Coalescence.jl:503-708 are evaluated by one fixed `quad_order`-point Gauss rule per
distribution instead of nested adaptive quadgk.
"""
function numerical_plan(pdists, kernel_func_normalized, norms; quad_order = 10)
    d = new_desc()
    d[].n_modes = length(pdists)
    d[].dist_type = pad(map(dist_code, pdists), MAX_MODES, Int32(0))
    d[].norms = Float64.(norms)
    d[].coal_style = 1
    d[].kernel_func, d[].kernel_func_params = kernel_func_code(kernel_func_normalized)
    d[].kernel_func_is_normalized = 1
    d[].quad_order = quad_order
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:cloudy_plan_create, lib), Cint, (Ref{PlanDesc}, Ref{Ptr{Cvoid}}), d, h))
    return h[]
end

"""
    plan(pdists, kernels, NProgMoms, thresholds, norms, ts; vel = ())

`kernels[j][k]::CoalescenceTensor` un-normalised (as passed to `CoalescenceData`), `thresholds` in physical mass
units (FixedThreshold) or percentiles (MovingThreshold) -- the arguments of `CoalescenceData(...)`,
src/Sources/Coalescence.jl:55-87.
"""
function plan(pdists, kernels, NProgMoms, thresholds, norms, ts = FixedThreshold(); vel = ())
    N = length(pdists)
    P = size(kernels[1][1].c, 1)
    # row-major [N][N][P][P] with c[a][b] multiplying x^a y^b
    c = Float64[kernels[j][k].c[a, b] for b in 1:P, a in 1:P, k in 1:N, j in 1:N]
    d = new_desc()
    d[].n_modes = N
    d[].dist_type = pad(map(dist_code, pdists), MAX_MODES, Int32(0))
    d[].tensor_p = P
    d[].kernel_layout = 1
    d[].dist_thresholds = pad(Float64.(thresholds), MAX_MODES, Inf)
    d[].threshold_style = ts isa MovingThreshold ? 1 : 0
    d[].norms = Float64.(norms)
    d[].n_vel = length(vel)
    d[].vel = pad(Float64.(collect(Iterators.flatten(vel))), 2MAX_VEL, 0.0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve c begin
        d[].kernel_c = pointer(c)
        check(ccall((:cloudy_plan_create, lib), Cint, (Ref{PlanDesc}, Ref{Ptr{Cvoid}}), d, h))
    end
    return h[]      # finalize with ccall((:cloudy_plan_destroy, lib), Cvoid, (Ptr{Cvoid},), h)
end

"""
    make_box_model_rhs(coal_type, ts = FixedThreshold(); plan, stream = nothing, sync = true)

Same factory name and returned signature as test/examples/utils/box_model_helpers.jl:22-27 (`coal_type` =
`AnalyticalCoalStyle()` with a `plan(...)`, or `NumericalCoalStyle()` with a `numerical_plan(...)`).  `m`, `dm` are
device arrays (e.g. AMDGPU.ROCArray{Float64,2} of size (n_parcels, nmom)); `par` is passed through untouched.

Streams.  AMDGPU.jl arrays live on the task-local stream `AMDGPU.stream()`; the launch must go to THAT stream to be
ordered after the solver's own broadcasts on `m`.  `stream = nothing` (default) looks it up at every call;
pass a `hipStream_t` (as `Ptr{Cvoid}`) to pin one.  `sync = true` (default) waits for the launch before returning, so a
host-side solver (OrdinaryDiffEq stepping with host control flow) can read `dm` immediately -- SURVEY 8(b): "the shim
synchronises before returning unless the caller opts out".  With `sync = false` the call is asynchronous on the stream
and everything later enqueued on the same stream (AMDGPU.jl broadcasts of the solver) is ordered behind it.
"""
function make_box_model_rhs(::CoalescenceStyle, ts::ThresholdStyle = FixedThreshold(); plan, stream = nothing, sync::Bool = true)
    function rhs!(dm, m, par, t)
        n, ld = size(m, 1), stride(m, 2)
        s = stream === nothing ? current_stream() : stream
        check(ccall((:cloudy_coal_rhs, lib), Cint,
                    (Ptr{Cvoid}, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    plan, n, ld, pointer(m), pointer(dm), s))
        sync && check(ccall((:cloudy_stream_synchronize, lib), Cint, (Ptr{Cvoid},), s))
        return nothing
    end
    return rhs!
end

# the hipStream_t of the calling task when AMDGPU.jl is loaded (its arrays are ordered on that stream), else the
# default stream
function current_stream()
    if isdefined(Main, :AMDGPU)
        return Base.unsafe_convert(Ptr{Cvoid}, Main.AMDGPU.stream())
    end
    return C_NULL
end

"""
    solve_ssprk33!(u, plan, dt, n_steps; stream = nothing, sync = true)

`solve(prob, SSPRK33(), dt = dt)` for `n_steps` fixed steps on the device (final state only): the state stays in
registers over all stages, one read and one write of `u` per call.  `plan` from `plan(...)` or `numerical_plan(...)`.
"""
function solve_ssprk33!(u, plan, dt, n_steps; stream = nothing, sync::Bool = true)
    s = stream === nothing ? current_stream() : stream
    check(ccall((:cloudy_ssprk33_steps, lib), Cint,
                (Ptr{Cvoid}, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint, Ptr{Cvoid}),
                plan, size(u, 1), stride(u, 2), pointer(u), pointer(u), dt, n_steps, s))
    sync && check(ccall((:cloudy_stream_synchronize, lib), Cint, (Ptr{Cvoid},), s))
    return u
end

"""
    solve_rainshaft_ssprk33!(u, plan, nz, dz, dt, n_steps; stream = nothing, sync = true)

`solve(ODEProblem(make_rainshaft_rhs(AnalyticalCoalStyle()), m, tspan, p), SSPRK33(), dt = p.dt)` of
test/examples/Analytical/rainshaft_gamma_mixture.jl:59-60 for `size(u, 1) ÷ nz` independent columns of `nz <= 256`
cells stacked along the first axis (the reference's `m[nz, nmom]` layout for one column), final state only.
"""
function solve_rainshaft_ssprk33!(u, plan, nz, dz, dt, n_steps; stream = nothing, sync::Bool = true)
    s = stream === nothing ? current_stream() : stream
    check(ccall((:cloudy_rainshaft_ssprk33_steps, lib), Cint,
                (Ptr{Cvoid}, Csize_t, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cint, Ptr{Cvoid}),
                plan, nz, size(u, 1) ÷ nz, stride(u, 2), pointer(u), pointer(u), dz, dt, n_steps, s))
    sync && check(ccall((:cloudy_stream_synchronize, lib), Cint, (Ptr{Cvoid},), s))
    return u
end

# host-array convenience (copies over PCIe each call; for validation, not for production stepping)
function rhs_host!(dm::Matrix{Float64}, m::Matrix{Float64}, plan)
    check(ccall((:cloudy_coal_rhs_host, lib), Cint, (Ptr{Cvoid}, Csize_t, Csize_t, Ptr{Float64}, Ptr{Float64}),
                plan, size(m, 1), size(m, 1), m, dm))
end

function __init__()
    check_layout()
end

end # module
