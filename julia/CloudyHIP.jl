# CloudyHIP.jl -- ccall binding of libcloudy_hip.so (include/cloudy_hip.h) for Cloudy.jl.
# SOURCE ONLY: julia is not installed in the build image or on the GPU box, so this file has never been run;
# the same C ABI is exercised from Python/ctypes (cloudy.jl_amd/_lib.py, tests/test_gpu_parity.py).  What CAN be checked
# without Julia is checked by tests/test_host_abi.py (a parser of this file): the field order / types of `PlanDesc` against
# the library's own cloudy_plan_desc_layout(), the argument count and C types of EVERY ccall below against the prototypes
# of include/cloudy_hip.h, and the presence of the drop-in branches (Vector state, host arrays, lazy plan, rainshaft rhs).
# `check_layout()` repeats the struct check at load time once Julia runs it.  See INTEGRATION.md.
#
# Drop-in contract (reference file:line):
#   rhs = make_box_model_rhs(AnalyticalCoalStyle())           test/examples/utils/box_model_helpers.jl:22-27
#   prob = ODEProblem(rhs, moment_init, tspan, ODE_parameters)  test/examples/Analytical/box_single_gamma.jl:35
# run UNCHANGED with `using CloudyHIP: make_box_model_rhs` in place of the include of box_model_helpers.jl: the state may
# be the drivers' `Vector` (one parcel) or an (n_parcels, nmom) matrix, on the host (`Array`) or on the device
# (`AMDGPU.ROCArray`); the plan is built from `par` (`ODE_parameters`) on the first call and cached.
module CloudyHIP

using Cloudy, Cloudy.Coalescence, Cloudy.ParticleDistributions, Cloudy.EquationTypes, Cloudy.KernelTensors,
      Cloudy.KernelFunctions

const lib = get(ENV, "CLOUDY_HIP_LIB", joinpath(@__DIR__, "..", "cloudy.jl_amd", "libcloudy_hip.so"))

const MAX_MODES, MAX_P, MAX_VEL = 8, 8, 4
const COMM_ID_BYTES = 128

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
    specialize::Int32               # plan-time compiled kernels: 0 when available, 1 required, -1 off
    coal_style::Int32               # 0 AnalyticalCoalStyle, 1 NumericalCoalStyle
    kernel_func::Int32              # 0 Constant, 1 Linear, 2 Hydrodynamic, 3 Long (KernelFunctions.jl:39-86)
    kernel_func_is_normalized::Int32
    quad_order::Int32
    kernel_func_params::NTuple{3,Float64}
    thresholds_are_normalized::Int32  # 1: dist_thresholds is the CoalescenceData FIELD (already / norms[2])
    quad_mode::Int32                # 0 fixed rule, 1 converged (split along the kernel function's non-smooth sets)
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

last_error() = unsafe_string(ccall((:cloudy_last_error, lib), Cstring, ()))
check(rc) = rc == 0 ? nothing : error("libcloudy_hip: ", last_error())

dist_code(::ExponentialPrimitiveParticleDistribution) = Int32(0)
dist_code(::GammaPrimitiveParticleDistribution) = Int32(1)
dist_code(::MonodispersePrimitiveParticleDistribution) = Int32(2)
dist_code(::LognormalPrimitiveParticleDistribution) = Int32(3)

pad(t, n, z) = ntuple(i -> i <= length(t) ? t[i] : z, n)

new_desc() = (d = Ref(PlanDesc(0, 0, pad((), MAX_MODES, Int32(0)), 0, 0, 0, C_NULL, pad((), MAX_MODES, Inf), 0,
                               (1.0, 1.0), (eps(Float64), 10.0), 15, 0, 0, pad((), 2MAX_VEL, 0.0), -1, 0,
                               0, 0, 0, 10, (0.0, 0.0, 0.0), 0, 0));
              ccall((:cloudy_plan_desc_init, lib), Cvoid, (Ref{PlanDesc},), d); d)

kernel_func_code(k::ConstantKernelFunction) = (Int32(0), (k.coll_coal_rate, 0.0, 0.0))
kernel_func_code(k::LinearKernelFunction) = (Int32(1), (k.coll_coal_rate, 0.0, 0.0))
kernel_func_code(k::HydrodynamicKernelFunction) = (Int32(2), (k.coal_eff, 0.0, 0.0))
kernel_func_code(k::LongKernelFunction) =
    (Int32(3), (k.x_threshold, k.coal_rate_below_threshold, k.coal_rate_above_threshold))

# A plan handle that frees itself; `specialized` = the launches run kernels compiled for this plan (hiprtc).  Without
# them the ahead-of-time kernels serve the plan -- same results, ~20 % slower on the all-Inf path -- so say so once.
mutable struct Plan
    handle::Ptr{Cvoid}
    nmom::Int
    specialized::Bool
    function Plan(h::Ptr{Cvoid})
        p = new(h, Int(ccall((:cloudy_plan_nmom, lib), Cint, (Ptr{Cvoid},), h)),
                ccall((:cloudy_plan_specialized, lib), Cint, (Ptr{Cvoid},), h) == 1)
        p.specialized || @warn "libcloudy_hip: plan runs the ahead-of-time kernels (no plan-time specialisation)" reason =
            unsafe_string(ccall((:cloudy_plan_jit_log, lib), Cstring, (Ptr{Cvoid},), h))
        finalizer(q -> ccall((:cloudy_plan_destroy, lib), Cvoid, (Ptr{Cvoid},), q.handle), p)
        return p
    end
end
Base.unsafe_convert(::Type{Ptr{Cvoid}}, p::Plan) = p.handle

function create(d::Ref{PlanDesc})
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:cloudy_plan_create, lib), Cint, (Ref{PlanDesc}, Ref{Ptr{Cvoid}}), d, h))
    return Plan(h[])
end

"""
    numerical_plan(pdists, kernel_func_normalized, norms; quad_order = 0, quad_mode = 1)

Plan of `make_box_model_rhs(NumericalCoalStyle())`: `kernel_func_normalized` is what the drivers put in
`p.kernel_func` (`get_normalized_kernel_func(kernel, norms)`, test/examples/Numerical/n_particles_gamma.jl:35).
`quad_mode = 1` (converged, the default -- the drop-in meets the reference's `quadgk(rtol = 1e-8)` answer): closed forms plus an
adaptive Gauss-Kronrod rule per mode (`quad_order` is then only the points per panel of the inner rule a Lognormal mode needs;
0 = the default, 8).  `quad_mode = 0`: one fixed `quad_order`-point Gauss rule per distribution (0 = 10 points): faster, 1e-4 ... 4e-2
of scale away from the reference for the hydrodynamic and Long kernels -- the explicit opt-in.
"""
function numerical_plan(pdists, kernel_func_normalized, norms; quad_order = 0, quad_mode = 1)
    d = new_desc()
    d[].n_modes = length(pdists)
    d[].dist_type = pad(map(dist_code, pdists), MAX_MODES, Int32(0))
    d[].norms = Float64.(norms)
    d[].coal_style = 1
    d[].kernel_func, d[].kernel_func_params = kernel_func_code(kernel_func_normalized)
    d[].kernel_func_is_normalized = 1
    d[].quad_order = quad_order
    d[].quad_mode = quad_mode
    return create(d)
end

"""
    plan(pdists, kernels, NProgMoms, thresholds, norms, ts; vel = ())

`kernels[j][k]::CoalescenceTensor` un-normalised (as passed to `CoalescenceData`), `thresholds` in physical mass
units (FixedThreshold) or percentiles (MovingThreshold) -- the arguments of `CoalescenceData(...)`,
src/Sources/Coalescence.jl:55-87.
"""
function plan(pdists, kernels, NProgMoms, thresholds, norms, ts = FixedThreshold(); vel = (), normalized::Bool = false)
    N = length(pdists)
    P = size(kernels[1][1].c, 1)
    # row-major [N][N][P][P] with c[a][b] multiplying x^a y^b
    c = Float64[kernels[j][k].c[a, b] for b in 1:P, a in 1:P, k in 1:N, j in 1:N]
    d = new_desc()
    d[].n_modes = N
    d[].dist_type = pad(map(dist_code, pdists), MAX_MODES, Int32(0))
    d[].tensor_p = P
    d[].kernel_layout = 1
    d[].kernel_is_normalized = normalized ? 1 : 0
    d[].thresholds_are_normalized = normalized ? 1 : 0
    d[].dist_thresholds = pad(Float64.(thresholds), MAX_MODES, Inf)
    d[].threshold_style = ts isa MovingThreshold ? 1 : 0
    d[].norms = Float64.(norms)
    d[].n_vel = length(vel)
    d[].vel = pad(Float64.(collect(Iterators.flatten(vel))), 2MAX_VEL, 0.0)
    GC.@preserve c begin
        d[].kernel_c = pointer(c)
        return create(d)
    end
end

"""
    plan_from_parameters(coal_type, ts, par)

The plan of a driver's `ODE_parameters` NamedTuple, fields exactly as the reference drivers build them
(test/examples/Analytical/box_gamma_mixture.jl:29-35): `par.pdists` (types only), `par.NProgMoms`, `par.norms`, and
`par.coal_data::CoalescenceData` -- whose `kernels` are ALREADY normalised and whose `dist_thresholds` are already divided
by `norms[2]` (Coalescence.jl:55-84), so both go through unchanged (`normalized = true`) -- or `par.kernel_func` for
`NumericalCoalStyle` (box_model_helpers.jl:47-48); `par.vel` when present (rainshaft drivers).
"""
function plan_from_parameters(coal_type::CoalescenceStyle, ts::ThresholdStyle, par)
    if coal_type isa NumericalCoalStyle
        return numerical_plan(par.pdists, par.kernel_func, par.norms)
    elseif coal_type isa AnalyticalCoalStyle
        cd = par.coal_data
        return plan(par.pdists, cd.kernels, par.NProgMoms, cd.dist_thresholds, par.norms, ts;
                    vel = haskey(par, :vel) ? par.vel : (), normalized = true)
    end
    error("Invalid coal style!")   # box_model_helpers.jl:49-50
end

# one cached plan per factory call: rebuilt only if the driver hands over different parameters
mutable struct PlanCache
    key::Any
    plan::Union{Nothing,Plan}
end
# (The key is CONTENT-based although it is spelled objectid: CoalescenceData, CoalescenceTensor (an SMatrix inside) and the
# kernel-function structs are immutable isbits structs (src/Sources/Coalescence.jl:45-53, src/Kernels/KernelTensors.jl:44-52,
# KernelFunctions.jl:39-87), for which `===` compares bits and objectid hashes them.  A driver that rebuilds an EQUAL
# CoalescenceData in every step therefore hits this cache and pays neither cloudy_plan_create nor hiprtc again; a changed
# tensor, threshold, norm or velocity builds a new plan.)
function cached_plan!(c::PlanCache, coal_type, ts, par)
    key = (objectid(coal_type isa NumericalCoalStyle ? par.kernel_func : par.coal_data), map(typeof, par.pdists),
           par.NProgMoms, par.norms, haskey(par, :vel) ? par.vel : ())
    if c.plan === nothing || c.key != key
        c.plan = plan_from_parameters(coal_type, ts, par)
        c.key = key
    end
    return c.plan
end

is_host(a) = a isa Array || (a isa SubArray && parent(a) isa Array) || (a isa Base.ReshapedArray && parent(a) isa Array)

# (n_parcels, leading dimension) of a state in the library's moment-major layout: a Vector is ONE parcel whose nmom
# moments are planes of leading dimension 1 (every reference box driver, box_single_gamma.jl:15); a matrix is
# m[parcel, moment], Julia column-major, unit stride along parcels
function batch_shape(m::AbstractVector, nmom)
    length(m) == nmom || error("state has $(length(m)) moments, the plan has $nmom")
    stride(m, 1) == 1 || error("state vector must be contiguous")
    return 1, 1
end
# (views: `stride` of a SubArray is the parent's stride times the step of the range, so `@view big[1:2:end, :]` is refused by the
# unit-stride test below, `@view big[1001:2000, :]` passes with ld = the PARENT's leading dimension and the view's own pointer --
# the slice-of-batch case tests/test_gpu_numerical.py runs through the same (n, ld) pair)
function batch_shape(m::AbstractMatrix, nmom)
    size(m, 2) == nmom || error("state has $(size(m, 2)) moment columns, the plan has $nmom")
    stride(m, 1) == 1 || error("state must have unit stride along parcels (m[parcel, moment])")
    return size(m, 1), size(m, 1) == 1 ? max(stride(m, 2), 1) : stride(m, 2)
end

"""
    make_box_model_rhs(coal_type, ts = FixedThreshold(); plan = nothing, stream = nothing, sync = true)

Same factory name, positional arguments and returned `rhs!(dm, m, par, t)` as
test/examples/utils/box_model_helpers.jl:22-27: `rhs = make_box_model_rhs(AnalyticalCoalStyle())` stays as it is in the
drivers.  The plan is built from `par` on the first call (`plan_from_parameters`) and cached; pass `plan = ...` to share
one.  `m`, `dm`: a `Vector` (one parcel) or an (n_parcels, nmom) matrix with equal strides.  Host `Array`s go through
`cloudy_coal_rhs_host` (staged over PCIe: validation and single boxes); device arrays (AMDGPU.ROCArray) are used in
place via `cloudy_coal_rhs`.

Streams.  AMDGPU.jl arrays live on the task-local stream `AMDGPU.stream()`; the launch goes to THAT stream so that it
is ordered after the solver's own broadcasts on `m` (`stream = nothing` looks it up at every call; pass a `hipStream_t`
as `Ptr{Cvoid}` to pin one).  `sync = true` (default) waits for the launch before returning, so a host-side solver can
read `dm` immediately (SURVEY 8(b)); with `sync = false` the call is asynchronous on the stream.
"""
function make_box_model_rhs(coal_type::CoalescenceStyle, ts::ThresholdStyle = FixedThreshold(); plan = nothing,
                            stream = nothing, sync::Bool = true)
    cache = PlanCache(nothing, plan)
    function rhs!(dm, m, par, t)
        p = plan === nothing ? cached_plan!(cache, coal_type, ts, par) : plan
        n, ld = batch_shape(m, p.nmom)
        (n, ld) == batch_shape(dm, p.nmom) || error("dm and m must have the same shape and strides")
        if is_host(m) && is_host(dm)
            GC.@preserve m dm check(ccall((:cloudy_coal_rhs_host, lib), Cint,
                                          (Ptr{Cvoid}, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}),
                                          p.handle, n, ld, pointer(m), pointer(dm)))
            return nothing
        end
        (is_host(m) || is_host(dm)) && error("m and dm must both be host arrays or both device arrays")
        s = stream === nothing ? current_stream() : stream
        check(ccall((:cloudy_coal_rhs, lib), Cint,
                    (Ptr{Cvoid}, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    p.handle, n, ld, pointer(m), pointer(dm), s))
        sync && check(ccall((:cloudy_stream_synchronize, lib), Cint, (Ptr{Cvoid},), s))
        return nothing
    end
    return rhs!
end

"""
    make_rainshaft_rhs(coal_type; plan = nothing, stream = nothing)

Same factory name and returned out-of-place `rhs(m, p, t)` as test/examples/utils/rainshaft_helpers.jl:45-89 for
`m[nz, nmom]` (one column; `nz * n_columns` rows stack independent columns): negatives are clamped to zero IN PLACE as the
reference does (`:52`), then `cloudy_rainshaft_rhs` = coalescence source + upwind divergence of the sedimentation flux
with `p.vel`, `p.dz`.  Returns a new array like `m`.  Host arrays are staged through device buffers.
"""
function make_rainshaft_rhs(coal_type::CoalescenceStyle; plan = nothing, stream = nothing, nz = nothing)
    cache = PlanCache(nothing, plan)
    function rhs(m, p, t)
        pl = plan === nothing ? cached_plan!(cache, coal_type, FixedThreshold(), p) : plan
        n, ld = batch_shape(m, pl.nmom)
        cells = nz === nothing ? n : nz          # one column unless the caller stacks several
        n % cells == 0 || error("size(m, 1) must be a multiple of nz")
        m .= max.(m, zero(eltype(m)))            # rainshaft_helpers.jl:52 mutates the integrator's array
        out = similar(m)
        if is_host(m)
            bytes = sizeof(eltype(m)) * ld * pl.nmom
            buf = Ref{Ptr{Cvoid}}(C_NULL)
            check(ccall((:cloudy_malloc, lib), Cint, (Ref{Ptr{Cvoid}}, Csize_t), buf, 3 * bytes))
            d_m, d_flux, d_out = buf[], buf[] + bytes, buf[] + 2 * bytes
            try
                GC.@preserve m out begin
                    check(ccall((:cloudy_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                                d_m, pointer(m), bytes, C_NULL))
                    check(ccall((:cloudy_rainshaft_rhs, lib), Cint,
                                (Ptr{Cvoid}, Csize_t, Csize_t, Csize_t, Ptr{Cvoid}, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                                pl.handle, cells, n ÷ cells, ld, d_m, Float64(p.dz), d_flux, d_out, C_NULL))
                    check(ccall((:cloudy_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                                pointer(out), d_out, bytes, C_NULL))
                    check(ccall((:cloudy_stream_synchronize, lib), Cint, (Ptr{Cvoid},), C_NULL))
                end
            finally
                ccall((:cloudy_free, lib), Cint, (Ptr{Cvoid},), buf[])
            end
            return out
        end
        flux = similar(m)
        s = stream === nothing ? current_stream() : stream
        check(ccall((:cloudy_rainshaft_rhs, lib), Cint,
                    (Ptr{Cvoid}, Csize_t, Csize_t, Csize_t, Ptr{Cvoid}, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    pl.handle, cells, n ÷ cells, ld, pointer(m), Float64(p.dz), pointer(flux), pointer(out), s))
        check(ccall((:cloudy_stream_synchronize, lib), Cint, (Ptr{Cvoid},), s))
        return out
    end
    return rhs
end

# The hipStream_t of the calling task.  Device arrays come from AMDGPU.jl and are ordered on ITS task-local stream; a
# launch on any other stream (the NULL stream included) would race with the kernels that produced `m`, and `sync = true`
# would then wait on the wrong stream.  AMDGPU.jl is looked up among the LOADED modules -- it may have been loaded inside
# the host's own package, where `isdefined(Main, :AMDGPU)` is false -- and when it cannot be found a device array without
# an explicit `stream = ...` is an error, not a silent launch on the NULL stream.
function amdgpu_module()
    for (id, mod) in Base.loaded_modules     # (by name: whichever environment or package loaded it)
        id.name == "AMDGPU" && return mod
    end
    return nothing
end
function current_stream()
    mod = amdgpu_module()
    mod === nothing && error("CloudyHIP: a device array was passed but AMDGPU.jl is not among Base.loaded_modules, so the " *
                             "stream its kernels are ordered on cannot be determined; pass `stream = <hipStream_t>` explicitly")
    return Base.unsafe_convert(Ptr{Cvoid}, Base.invokelatest(mod.stream))
end

"""
    solve_ssprk33!(u, plan, dt, n_steps; stream = nothing, sync = true)

`solve(prob, SSPRK33(), dt = dt)` for `n_steps` fixed steps on the device (final state only): the state stays in
registers over all stages, one read and one write of `u` per call.  `plan` from `plan(...)` or `numerical_plan(...)`;
`u` a device array (Vector or (n_parcels, nmom) matrix).
"""
function solve_ssprk33!(u, plan::Plan, dt, n_steps; stream = nothing, sync::Bool = true)
    n, ld = batch_shape(u, plan.nmom)
    s = stream === nothing ? current_stream() : stream
    check(ccall((:cloudy_ssprk33_steps, lib), Cint,
                (Ptr{Cvoid}, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint, Ptr{Cvoid}),
                plan.handle, n, ld, pointer(u), pointer(u), dt, n_steps, s))
    sync && check(ccall((:cloudy_stream_synchronize, lib), Cint, (Ptr{Cvoid},), s))
    return u
end

"""
    solve_tsit5!(u, plan, dt, n_steps; stream = nothing, sync = true)

`solve(prob, Tsit5(), dt = dt, adaptive = false)` for `n_steps` fixed steps on the device (BASELINE configs[0] names Tsit5;
the reference drivers use SSPRK33): the Tsit5 tableau per parcel, state and stage derivatives in registers.
"""
function solve_tsit5!(u, plan::Plan, dt, n_steps; stream = nothing, sync::Bool = true)
    n, ld = batch_shape(u, plan.nmom)
    s = stream === nothing ? current_stream() : stream
    check(ccall((:cloudy_tsit5_steps, lib), Cint,
                (Ptr{Cvoid}, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint, Ptr{Cvoid}),
                plan.handle, n, ld, pointer(u), pointer(u), dt, n_steps, s))
    sync && check(ccall((:cloudy_stream_synchronize, lib), Cint, (Ptr{Cvoid},), s))
    return u
end

"""
    solve_rainshaft_ssprk33!(u, plan, nz, dz, dt, n_steps; stream = nothing, sync = true)

`solve(ODEProblem(make_rainshaft_rhs(AnalyticalCoalStyle()), m, tspan, p), SSPRK33(), dt = p.dt)` of
test/examples/Analytical/rainshaft_gamma_mixture.jl:59-60 for `size(u, 1) ÷ nz` independent columns of `nz`
cells (fused in one launch for `nz <= 1024`, stepped stage by stage inside the library above that) stacked along the first axis
(the reference's `m[nz, nmom]` layout for one column), final state only.
"""
function solve_rainshaft_ssprk33!(u, plan::Plan, nz, dz, dt, n_steps; stream = nothing, sync::Bool = true)
    n, ld = batch_shape(u, plan.nmom)
    s = stream === nothing ? current_stream() : stream
    check(ccall((:cloudy_rainshaft_ssprk33_steps, lib), Cint,
                (Ptr{Cvoid}, Csize_t, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cint, Ptr{Cvoid}),
                plan.handle, nz, n ÷ nz, ld, pointer(u), pointer(u), dz, dt, n_steps, s))
    sync && check(ccall((:cloudy_stream_synchronize, lib), Cint, (Ptr{Cvoid},), s))
    return u
end

# ---- multi-GPU: the conservation diagnostic summed over ranks (moments_sum, plotting_helpers.jl:240-252) ------------
# One Julia process per GPU (MPI.jl / Distributed.jl own the processes): rank 0 draws the id, the host broadcasts its
# 128 bytes, every rank forms the communicator; the all-reduce itself is RCCL inside the library.
#     id = rank == 0 ? CloudyHIP.comm_unique_id() : Vector{UInt8}(undef, CloudyHIP.COMM_ID_BYTES)
#     MPI.Bcast!(id, 0, MPI.COMM_WORLD)
#     comm = CloudyHIP.comm_create(MPI.Comm_size(MPI.COMM_WORLD), rank, id; device = local_rank)
#     sums = CloudyHIP.moment_sums_allreduce(plan, comm, dm)      # Vector{Float64}(nmom), identical on every rank
function comm_unique_id()
    id = Vector{UInt8}(undef, COMM_ID_BYTES)
    check(ccall((:cloudy_comm_unique_id, lib), Cint, (Ptr{Cvoid},), id))
    return id
end

function comm_create(world_size::Integer, rank::Integer, id::Vector{UInt8}; device::Integer = -1)
    length(id) == COMM_ID_BYTES || error("the unique id has $COMM_ID_BYTES bytes")
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:cloudy_comm_create, lib), Cint, (Cint, Cint, Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}),
                world_size, rank, id, device, h))
    return h[]      # release with comm_destroy
end

comm_destroy(comm::Ptr{Cvoid}) = ccall((:cloudy_comm_destroy, lib), Cvoid, (Ptr{Cvoid},), comm)

"""
    moment_sums_allreduce(plan, comm, arr; stream = nothing)

Plane sums of this rank's device array `arr` (n_parcels, planes) summed over all ranks: `cloudy_moment_sums` followed by
`ncclAllReduce(sum, Float64, planes)` on the stream, inside libcloudy_hip.so.  Returns a host `Vector{Float64}`.
"""
function moment_sums_allreduce(plan::Plan, comm::Ptr{Cvoid}, arr; stream = nothing)
    planes = size(arr, 2)
    s = stream === nothing ? current_stream() : stream
    dev = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:cloudy_malloc, lib), Cint, (Ref{Ptr{Cvoid}}, Csize_t), dev, 8 * planes))
    out = Vector{Float64}(undef, planes)
    try
        check(ccall((:cloudy_moment_sums_allreduce, lib), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Csize_t, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    plan.handle, comm, size(arr, 1), stride(arr, 2), planes, pointer(arr), dev[], s))
        check(ccall((:cloudy_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), out, dev[], 8 * planes, s))
        check(ccall((:cloudy_stream_synchronize, lib), Cint, (Ptr{Cvoid},), s))
    finally
        ccall((:cloudy_free, lib), Cint, (Ptr{Cvoid},), dev[])
    end
    return out
end

"""
    closure_stats(plan, m; stream = nothing) -> Matrix{Int} (N x 4)

How many parcels of the device array `m` (n_parcels, nmom; physical units) had a mode's closure inversion replaced or clamped:
columns = fallback distribution `(0, 1, 1)`, shape at the lower clamp, shape at the upper clamp, and moments for which
`check_moment_consistency` (ParticleDistributions.jl:437-449) would throw.  The reference clamps silently per call
(:456-541); this is the batch form of both, one pass on the device (`cloudy_closure_stats`).
"""
function closure_stats(plan::Plan, m; stream = nothing)
    N = Int(ccall((:cloudy_plan_nparams, lib), Cint, (Ptr{Cvoid},), plan.handle)) ÷ 3
    counts = Vector{UInt64}(undef, 4 * N)
    s = stream === nothing ? current_stream() : stream
    check(ccall((:cloudy_closure_stats, lib), Cint, (Ptr{Cvoid}, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{UInt64}, Ptr{Cvoid}),
                plan.handle, size(m, 1), stride(m, 2), pointer(m), counts, s))
    return permutedims(reshape(Int.(counts), 4, N))
end

function __init__()
    check_layout()
end

end # module
