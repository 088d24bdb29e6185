# CloudyHIP.jl -- ccall binding of libcloudy_hip.so (include/cloudy_hip.h) for Cloudy.jl.
# SOURCE ONLY: julia is not installed in the build image or on the GPU box, so this file has never been run;
# the same C ABI is exercised from Python/ctypes (cloudy.jl_amd/_lib.py, tests/test_gpu_parity.py).
# See INTEGRATION.md.
module CloudyHIP

using Cloudy, Cloudy.Coalescence, Cloudy.ParticleDistributions, Cloudy.EquationTypes, Cloudy.KernelTensors

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
end

check(rc) = rc == 0 ? nothing : error("libcloudy_hip: ", unsafe_string(ccall((:cloudy_last_error, lib), Cstring, ())))

dist_code(::ExponentialPrimitiveParticleDistribution) = Int32(0)
dist_code(::GammaPrimitiveParticleDistribution) = Int32(1)
dist_code(::MonodispersePrimitiveParticleDistribution) = Int32(2)
dist_code(::LognormalPrimitiveParticleDistribution) = Int32(3)

pad(t, n, z) = ntuple(i -> i <= length(t) ? t[i] : z, n)

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
    d = Ref(PlanDesc(0, 0, pad((), MAX_MODES, Int32(0)), 0, 0, 0, C_NULL, pad((), MAX_MODES, Inf), 0,
                     (1.0, 1.0), (eps(Float64), 10.0), 15, 0, 0, pad((), 2MAX_VEL, 0.0), -1, 0))
    ccall((:cloudy_plan_desc_init, lib), Cvoid, (Ref{PlanDesc},), d)
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
    make_box_model_rhs(::AnalyticalCoalStyle, ts = FixedThreshold(); plan, stream = C_NULL)

Same factory name and returned signature as test/examples/utils/box_model_helpers.jl:22-27.  `m`, `dm` are
device arrays (e.g. AMDGPU.ROCArray{Float64,2} of size (n_parcels, nmom)); `par` is passed through untouched.
"""
function make_box_model_rhs(::AnalyticalCoalStyle, ts::ThresholdStyle = FixedThreshold(); plan, stream = C_NULL)
    function rhs!(dm, m, par, t)
        n, ld = size(m, 1), stride(m, 2)
        check(ccall((:cloudy_coal_rhs, lib), Cint,
                    (Ptr{Cvoid}, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    plan, n, ld, pointer(m), pointer(dm), stream))
        return nothing
    end
    return rhs!
end

"""
    solve_ssprk33!(u, plan, dt, n_steps; stream = C_NULL)

`solve(prob, SSPRK33(), dt = dt)` for `n_steps` fixed steps on the device (final state only): the state stays in
registers over all stages, one read and one write of `u` per call.
"""
function solve_ssprk33!(u, plan, dt, n_steps; stream = C_NULL)
    check(ccall((:cloudy_ssprk33_steps, lib), Cint,
                (Ptr{Cvoid}, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint, Ptr{Cvoid}),
                plan, size(u, 1), stride(u, 2), pointer(u), pointer(u), dt, n_steps, stream))
    return u
end

"""
    solve_rainshaft_ssprk33!(u, plan, nz, dz, dt, n_steps; stream = C_NULL)

`solve(ODEProblem(make_rainshaft_rhs(AnalyticalCoalStyle()), m, tspan, p), SSPRK33(), dt = p.dt)` of
test/examples/Analytical/rainshaft_gamma_mixture.jl:59-60 for `size(u, 1) ÷ nz` independent columns of `nz <= 256`
cells stacked along the first axis (the reference's `m[nz, nmom]` layout for one column), final state only.
"""
function solve_rainshaft_ssprk33!(u, plan, nz, dz, dt, n_steps; stream = C_NULL)
    check(ccall((:cloudy_rainshaft_ssprk33_steps, lib), Cint,
                (Ptr{Cvoid}, Csize_t, Csize_t, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cint, Ptr{Cvoid}),
                plan, nz, size(u, 1) ÷ nz, stride(u, 2), pointer(u), pointer(u), dz, dt, n_steps, stream))
    return u
end

# host-array convenience (copies over PCIe each call; for validation, not for production stepping)
function rhs_host!(dm::Matrix{Float64}, m::Matrix{Float64}, plan)
    check(ccall((:cloudy_coal_rhs_host, lib), Cint, (Ptr{Cvoid}, Csize_t, Csize_t, Ptr{Float64}, Ptr{Float64}),
                plan, size(m, 1), size(m, 1), m, dm))
end

end # module
