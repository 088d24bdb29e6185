/*
 * cloudy_hip.h -- C ABI of libcloudy_hip.so: the MI355X (gfx950) batched collision-coalescence
 * moment-tendency operator for Cloudy.jl.
 *
 * Drop-in boundary.  Each entry point replaces one Julia-level interface of CliMA/Cloudy.jl v0.6.0
 * (file:line relative to the reference checkout) for a batch of independent 0-D parcels; the
 * Julia-side `ccall` binding a maintainer would add is shown in INTEGRATION.md.
 *
 *   cloudy_plan_create            <- CoalescenceData(kernel, NProgMoms, dist_thresholds, norms, ts)
 *                                    src/Sources/Coalescence.jl:55-104  (+ the ODE_parameters NamedTuple
 *                                    test/examples/Analytical/box_gamma_mixture.jl:29-35)
 *   cloudy_coal_rhs               <- rhs!(dm, m, par, t) = rhs_coal!(AnalyticalCoalStyle(), dm, m, par, ts)
 *                                    test/examples/utils/box_model_helpers.jl:22-53
 *   cloudy_get_coal_ints          <- get_coal_ints(::AnalyticalCoalStyle, pdists, coal_data[, ::MovingThreshold])
 *                                    src/Sources/Coalescence.jl:115-185
 *   (coal_style = CLOUDY_NUMERICAL_COAL: the same two entry points are
 *    rhs_coal!(NumericalCoalStyle(), dm, m, par, ts) with par.kernel_func, box_model_helpers.jl:47-48, and
 *    get_coal_ints(::NumericalCoalStyle, pdists, kernel_func), src/Sources/Coalescence.jl:470-489 -- the reference's
 *    nested adaptive quadgk replaced by a fixed Gauss rule (cloudy_plan_desc.quad_order) or, with quad_mode =
 *    CLOUDY_QUAD_CONVERGED, by closed forms + one 1-D rule per mode that reach quadgk's answer to <= 1e-8 of scale --
 *    the GUARANTEED bound, asserted on the device against the golden values; measured <= 5.5e-10)
 *   cloudy_update_dist_from_moments <- update_dist_from_moments(pdist, moments)
 *                                    src/ParticleDistributions/ParticleDistributions.jl:456-476, 512-523
 *   cloudy_finite_2d_integrals    <- get_finite_2d_integrals / moment_source_helper
 *                                    src/Sources/Coalescence.jl:200-244, ParticleDistributions.jl:567-612
 *   cloudy_compute_thresholds     <- compute_thresholds(pdists, percentiles)
 *                                    src/ParticleDistributions/ParticleDistributions.jl:734-761
 *   cloudy_sedimentation_flux     <- get_sedimentation_flux(pdists, vel)   src/Sources/Sedimentation.jl:22-37
 *   cloudy_rainshaft_sources      <- the per-cell body of make_rainshaft_rhs
 *                                    test/examples/utils/rainshaft_helpers.jl:52-78
 *   cloudy_rainshaft_rhs          <- rhs(m, p, t) of make_rainshaft_rhs incl. the flux divergence
 *                                    test/examples/utils/rainshaft_helpers.jl:45-89
 *   cloudy_rainshaft_ssprk33_steps <- solve(ODEProblem(make_rainshaft_rhs(...), m, tspan, p), SSPRK33(), dt = p.dt),
 *                                    test/examples/Analytical/rainshaft_gamma_mixture.jl:59-60
 *   cloudy_standard_N_q           <- get_standard_N_q(pdists, size_cutoff)  ParticleDistributions.jl:634-687
 *   cloudy_cond_evap              <- rhs_condensation!(dmom, mom, p, s) / get_cond_evap
 *                                    test/examples/utils/box_model_helpers.jl:55-67, src/Sources/Condensation.jl:22-37
 *   cloudy_ssprk33_steps          <- solve(ODEProblem(rhs, m0, tspan, p), SSPRK33(), dt = p.dt) of the drivers,
 *                                    test/examples/Analytical/box_single_gamma.jl:35-36 (OrdinaryDiffEq stepping)
 *   cloudy_tsit5_steps            <- solve(prob, Tsit5(), dt = ..., adaptive = false) (BASELINE configs[0]; OrdinaryDiffEq tableau)
 *   cloudy_moment_sums            <- moments_sum diagnostic, test/examples/utils/plotting_helpers.jl:240-252
 *   cloudy_moment_sums_allreduce  <- the same summed over the ranks / GPUs that share a batch (RCCL; the reference is
 *                                    single-process, its global sum is the local one)
 *
 * Data layout (all batched calls): moment-major structure-of-arrays, element (q, parcel) at
 * base[q * ld + parcel], ld >= n_parcels -- exactly Julia's column-major m[parcel, moment]
 * (test/examples/utils/rainshaft_helpers.jl:48-56).  Plane q is the flat moment index of
 * get_dist_moment_ind (src/helper_functions.jl:13-20), 0-based.  Values are in physical units;
 * normalisation by norms happens inside, as in rhs_coal!.
 *
 * Ownership: the caller owns every buffer; a plan owns only its constant block and a small reduction
 * workspace (used by cloudy_moment_sums: do not run that call concurrently on ONE plan from several
 * streams -- cloudy_moment_sums_ws takes the workspace from the caller and is re-entrant; every other batched
 * call only reads the plan and may run concurrently).
 *
 * Devices: a plan belongs to the device it was created on (desc.device, or the caller's current device): its constant
 * block, node tables and plan-time compiled code objects live there.  Every batched call makes that device current
 * for the duration of the call and restores the caller's current device before returning, as does
 * cloudy_plan_create; `stream` and all buffers must belong to the plan's device.  No call allocates, except the
 * host-pointer convenience cloudy_coal_rhs_host.  All device entry points are asynchronous on `stream` (a hipStream_t passed as
 * void*; NULL = the default stream).  Errors: int status, never a C++ exception; the message
 * of the last failure on the calling thread is cloudy_last_error().
 *
 * Plain C: no HIP or torch types appear in any signature.
 */
#ifndef CLOUDY_HIP_H
#define CLOUDY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLOUDY_HIP_VERSION 100 /* 0.1.0 */

#define CLOUDY_MAX_MODES 8  /* N: number of sub-distributions */
#define CLOUDY_MAX_P 8      /* P = tensor order + 1 */
/* Plans of up to CLOUDY_AOT_MAX_MODES modes and order <= 4 (P <= CLOUDY_AOT_MAX_P; the reference's examples stop at 4
 * modes and order 4, box_gamma_mixture_4modes.jl:23, box_gamma_mixture_hydro.jl:23) have ahead-of-time compiled kernels
 * behind every entry point.  Larger plans (the reference's types bound neither N nor the order, Coalescence.jl:55-104, and
 * get_coal_ints(::NumericalCoalStyle) is generic in the number of modes, :470-489) run kernels compiled for the plan with hiprtc: the RHS / rainshaft / integrator kernels at
 * cloudy_plan_create (CLOUDY_EUNSUPPORTED there when plan-time compilation is off or fails -- there is no other path), the
 * per-mode diagnostics and the parameter-plane entry points (cloudy_update_dist_from_moments, cloudy_get_coal_ints,
 * cloudy_get_finite_2d_integrals, cloudy_compute_thresholds, cloudy_sedimentation_flux, cloudy_cond_evap_rhs,
 * cloudy_standard_N_q) on their first call.  A NumericalCoalStyle plan takes up to CLOUDY_MAX_MODES modes as well (round 5; 15 to
 * 40 s of compilation per kernel for 5 to 8 modes, cached on disk; the order of its "tensor" is 0: P = 1). */
#define CLOUDY_AOT_MAX_MODES 4
#define CLOUDY_AOT_MAX_P 5
#define CLOUDY_MAX_VEL 4    /* terms of the terminal-velocity power series */
#define CLOUDY_MAX_MOMENTS (3 * CLOUDY_MAX_MODES)

/* distribution families, ParticleDistributions.jl:66-159.  Parameter planes are (n, theta, k); Monodisperse uses
 * (n, theta), Lognormal keeps (n, mu, sigma) in the same three slots. */
enum { CLOUDY_DIST_EXPONENTIAL = 0, CLOUDY_DIST_GAMMA = 1, CLOUDY_DIST_MONODISPERSE = 2, CLOUDY_DIST_LOGNORMAL = 3 };
/* EquationTypes.jl:20-22 */
enum { CLOUDY_FIXED_THRESHOLD = 0, CLOUDY_MOVING_THRESHOLD = 1 };
enum { CLOUDY_F64 = 0, CLOUDY_F32 = 1, CLOUDY_F32_FAST = 2, CLOUDY_F64_RELAXED = 3 };
/* EquationTypes.jl:15-16 */
enum { CLOUDY_ANALYTICAL_COAL = 0, CLOUDY_NUMERICAL_COAL = 1 };
/* CoalescenceKernelFunction families, KernelFunctions.jl:39-86; parameters in cloudy_plan_desc.kernel_func_params:
 * (coll_coal_rate) | (coll_coal_rate) | (coal_eff) | (x_threshold, coal_rate_below_threshold, coal_rate_above_threshold) */
enum { CLOUDY_KFUNC_CONSTANT = 0, CLOUDY_KFUNC_LINEAR = 1, CLOUDY_KFUNC_HYDRODYNAMIC = 2, CLOUDY_KFUNC_LONG = 3 };
#define CLOUDY_MAX_QUAD 32  /* points of the fixed Gauss rule per distribution */
/* cloudy_plan_desc.quad_mode: how a NumericalCoalStyle plan evaluates the integrals of Coalescence.jl:503-708.
 * FIXED: one quad_order-point Gauss rule per distribution (BASELINE configs[3] as worded: "10-pt Gauss quadrature").
 * CONVERGED: the integrals split along the non-smooth sets of the kernel function, converged to the reference's
 * quadgk(rtol = 1e-8) answer (DESIGN.md 3.6 states the measured error and cost). */
enum { CLOUDY_QUAD_FIXED = 0, CLOUDY_QUAD_CONVERGED = 1 };
/* layout of cloudy_plan_desc.kernel_c */
enum { CLOUDY_KERNEL_SINGLE = 0 /* [P][P] shared by all pairs, Coalescence.jl:89-104 */,
       CLOUDY_KERNEL_MATRIX = 1 /* [N][N][P][P], Coalescence.jl:55-87 */ };

enum {
    CLOUDY_OK = 0,
    CLOUDY_EINVAL = -1,       /* bad argument (the reference throws in a constructor) */
    CLOUDY_ENOTSYMMETRIC = -2,/* check_symmetry failed, KernelTensors.jl:157-171 */
    CLOUDY_EHIP = -3,         /* HIP runtime error (message has hipGetErrorString) */
    CLOUDY_ENOMEM = -4,
    CLOUDY_EUNSUPPORTED = -5, /* valid in the reference, outside this build */
    CLOUDY_ENODEVICE = -6,
    CLOUDY_ECOMM = -7         /* RCCL error (message has ncclGetErrorString) */
};

typedef struct cloudy_plan cloudy_plan;

typedef struct cloudy_plan_desc {
    uint32_t struct_size;                    /* sizeof(cloudy_plan_desc); set by cloudy_plan_desc_init */
    int32_t n_modes;                         /* N = length(pdists) */
    int32_t dist_type[CLOUDY_MAX_MODES];     /* type of p.pdists[i]; NProgMoms[i] = nparams (2 / 3) */
    int32_t tensor_p;                        /* P = order + 1 */
    int32_t kernel_layout;                   /* CLOUDY_KERNEL_SINGLE / _MATRIX */
    int32_t kernel_is_normalized;            /* 0: library applies get_normalized_kernel_tensor(c, norms) */
    const double *kernel_c;                  /* c[a][b] multiplies x^a y^b (KernelTensors.jl:44-52); row-major */
    double dist_thresholds[CLOUDY_MAX_MODES];/* FixedThreshold: mass in physical units (divided by norms[1] inside,
                                                Coalescence.jl:78-84); MovingThreshold: percentiles. +INFINITY ok. */
    int32_t threshold_style;
    double norms[2];                         /* (n0, m0), helper_functions.jl:40-53 */
    double k_range[2];                       /* param_range.k, ParticleDistributions.jl:459; default (eps, 10) */
    int32_t n_bins_per_log_unit;             /* ParticleDistributions.jl:594; default 15 */
    int32_t dtype;                           /* element type of the mom / dmom / flux planes: CLOUDY_F64, or CLOUDY_F32
                                                (float planes in HBM = half the traffic; arithmetic stays fp64 in
                                                registers; (n, theta, k) / F diagnostics planes are always fp64), or
                                                CLOUDY_F32_FAST (float planes AND single-precision arithmetic: in the
                                                per-node Simpson / incomplete-gamma pass of threshold plans -- ~1e-6
                                                relative on the thresholded integrals; and, for plans WITHOUT a
                                                threshold compiled for the plan, in the whole cloudy_coal_rhs kernel
                                                (packed v_pk_fma_f32, four parcels per lane): <= 1e-2 of scale,
                                                99.9 % of parcels <= 1e-4, median 1e-8, and parcels whose closure is
                                                clamped to k = eps may come out Inf / NaN where fp64 arithmetic stays
                                                finite.  Every layout of a batch takes that arithmetic (16-byte
                                                accesses where ld % 4 == 0 and the planes are aligned, scalar ones
                                                otherwise: same bits).  cloudy_ssprk33_steps / cloudy_tsit5_steps of
                                                such a plan run fp64 arithmetic on the float planes (~2e-7): their
                                                right-hand side is more accurate than cloudy_coal_rhs of the same
                                                plan; use CLOUDY_F32 where the two must agree), or
                                                CLOUDY_F64_RELAXED (fp64 planes and arithmetic; the power series and the
                                                continued fraction of the incomplete gamma function of the threshold
                                                plans stop at 1e-11 instead of 1e-17 / 1e-16: <= 1e-9 of scale against
                                                the oracle, an order inside north_star's 1e-8 for this path; opt-in;
                                                plan-time compiled kernels only -- the ahead-of-time kernels, and plans
                                                without a threshold, compute as CLOUDY_F64) */
    int32_t n_vel;                           /* 0 = no sedimentation term */
    double vel[CLOUDY_MAX_VEL][2];           /* p.vel: terminal velocity sum_k vel[k][0] * x^vel[k][1], physical units */
    int32_t device;                          /* HIP device ordinal, -1 = current */
    int32_t specialize;                      /* plan-time compilation (hiprtc) of the cloudy_coal_rhs kernel (and, for
                                                all-Inf thresholds, the cloudy_ssprk33_steps kernel) with the plan as
                                                compile-time constants: 0 = when available (default; the
                                                environment variable CLOUDY_HIP_JIT=0 turns it off), 1 = required
                                                (plan creation fails otherwise), -1 = off */
    /* ---- NumericalCoalStyle plans (make_box_model_rhs(NumericalCoalStyle()), Coalescence.jl:470-708) ----
     * coal_style = CLOUDY_NUMERICAL_COAL: the integrals of the kernel FUNCTION p.kernel_func over the densities, where
     * the reference nests adaptive quadgk(rtol = 1e-8).  quad_mode = CLOUDY_QUAD_CONVERGED (THE DEFAULT: the drop-in
     * answers within north_star's 1e-8 of the reference): the integrals split along the kernel function's non-smooth
     * sets -- closed forms for Q and R, one ADAPTIVE Gauss-Kronrod (7, 15) rule per mode for the weighting_fn split
     * (panels walked from large sizes down, bisected until |K15 - G7| <= 1e-7 of the accumulated value -- every kernel
     * function since round 5: the Long kernel's panel above x_t, where G(s) ~ (s - x_t)^k, is integrated in xi with
     * s - x_t ~ xi^4, its range [x_t, 2 x_t] in a second phase from a per-rule incomplete-beta table -- and ended by a
     * rigorous bound of what is left; quad_order is then only the points per panel of the inner rule a Lognormal mode's
     * self-collision integral needs under the hydrodynamic or Long kernel, default 8).  ERROR: the guaranteed bound is
     * 1e-8 of scale against nested adaptive quadrature of the reference integrals (asserted on the device on the golden
     * cases); measured <= 5.5e-10 there, and <= 1.1e-10 against the same rule at 1e-13 over a thousand random multi-scale
     * mixtures (a statement about the rule's convergence, not about the reference integrals).  Lognormal modes: under the
     * constant / linear kernels the inner integral is a trapezoidal rule of step min(sigma, 1/2) (<= 2e-13 of the density's
     * peak, any sigma); under the hydrodynamic / Long kernels it takes Gauss-Legendre panels of at most 3 sigma (up to 256):
     * <= 1.5e-12 of scale down to sigma = 0.003, 1e-9 ... 2e-6 at sigma = 0.001, growing below
     * (csrc/quad_conv.hpp).  quad_mode = CLOUDY_QUAD_FIXED (explicit opt-in; BASELINE configs[3] "via 10-pt Gauss
     * quadrature"): each integral by one fixed quad_order-point Gauss rule per distribution (generalised Gauss-Laguerre
     * for Gamma / Exponential modes, Gauss-Hermite in ln x for Lognormal modes; tensor product over a pair of modes after
     * the substitution x' = x - y; default 10 points) -- exact for the constant and linear kernels up to the weighting_fn
     * split, a discretisation for the hydrodynamic and Long kernels (1e-4 ... 4e-2 of scale at 10 points), several
     * times faster.  kernel_c, tensor_p,
     * dist_thresholds and threshold_style are ignored (the style has no thresholds: weighting_fn splits the self
     * collisions, Coalescence.jl:624-642).  Monodisperse modes: CLOUDY_EINVAL (no normed_density_func method).
     * Shape parameters: the per-parcel Gauss-Laguerre rules are staged for 0 < k <= max(k_range[1], 1), the range
     * update_dist_from_moments can produce.  (n, theta, k) planes handed to cloudy_get_coal_ints with a Gamma k outside
     * that range give NaN tendencies for the parcel (a batch cannot fail per parcel): widen k_range[1] instead. */
    int32_t coal_style;                      /* CLOUDY_ANALYTICAL_COAL (default) / CLOUDY_NUMERICAL_COAL */
    int32_t kernel_func;                     /* CLOUDY_KFUNC_* */
    int32_t kernel_func_is_normalized;       /* 0: library applies get_normalized_kernel_func(kernel, norms), :124-154 */
    int32_t quad_order;                      /* FIXED: points per distribution; CONVERGED: Gauss-Legendre points per
                                                panel; 2..CLOUDY_MAX_QUAD; 0 (default) = 10 (FIXED) / 8 (CONVERGED) */
    double kernel_func_params[3];            /* physical units unless kernel_func_is_normalized */
    int32_t thresholds_are_normalized;       /* FixedThreshold only: 1 = dist_thresholds are already divided by norms[1],
                                                i.e. the CoalescenceData.dist_thresholds FIELD (Coalescence.jl:78-84) rather
                                                than the constructor argument -- a host that builds the plan from an
                                                existing CoalescenceData passes its fields through unchanged */
    int32_t quad_mode;                       /* NumericalCoalStyle: CLOUDY_QUAD_CONVERGED (default) / CLOUDY_QUAD_FIXED */
} cloudy_plan_desc;

/* fills defaults: k_range = (eps, 10), n_bins_per_log_unit = 15, norms = (1, 1), thresholds = +Inf,
 * dtype = F64, device = -1, coal_style = ANALYTICAL, quad_mode = CONVERGED, quad_order = 0 (the mode's default) */
void cloudy_plan_desc_init(cloudy_plan_desc *desc);

int cloudy_plan_create(const cloudy_plan_desc *desc, cloudy_plan **out);
void cloudy_plan_destroy(cloudy_plan *plan);
/* 1 if cloudy_coal_rhs / cloudy_ssprk33_steps of this plan run kernels compiled for it at plan creation, else 0 with
 * the reason (or the compiler log) in cloudy_plan_jit_log.  Either way the arithmetic is the same (results agree to
 * rounding, bit for bit in most configurations). */
int cloudy_plan_specialized(const cloudy_plan *plan);
const char *cloudy_plan_jit_log(const cloudy_plan *plan);
/* Build check without a GPU: validates `desc`, generates the plan's specialised translation unit(s) and compiles them
 * with hiprtc for `arch` (NULL = "gfx950"); nothing is loaded or launched.  CLOUDY_OK, or CLOUDY_EUNSUPPORTED with the
 * compiler log in cloudy_last_error().  arch = "source-only": every translation unit of the plan is generated and none compiled
 * (the host-side sanitizer run of the tests). */
int cloudy_jit_selfcheck(const cloudy_plan_desc *desc, const char *arch);
/* Layout of cloudy_plan_desc as this library was compiled: for each field, in declaration order, name / byte offset /
 * byte size.  A binding in another language (julia/CloudyHIP.jl, the ctypes mirror) checks its own struct against this
 * before the first cloudy_plan_create.  Returns the number of fields; fills at most `cap` entries of each array (any of
 * which may be NULL).  Names are static strings. */
int cloudy_plan_desc_layout(const char **names, uint32_t *offsets, uint32_t *sizes, int cap);
/* The per-parcel Gauss rule of the NumericalCoalStyle kernels, evaluated on the HOST by the same source the device
 * compiles (quad.hpp: start values from the staged table, two Newton steps, Christoffel weights): nodes u[quad_order]
 * and normalised weights W[quad_order] for the weight u^(k-1) e^-u / Gamma(k), 0 < k <= k_hi.  A diagnostic for tests
 * and hosts (no device involved, nothing batched: this is not a compute path). */
int cloudy_quad_rule_host(int quad_order, double k_hi, double k, double *u, double *W);
int cloudy_plan_nmom(const cloudy_plan *plan);      /* sum(NProgMoms) */
int cloudy_plan_device(const cloudy_plan *plan);    /* HIP device ordinal the plan lives on */
int cloudy_plan_nparams(const cloudy_plan *plan);   /* 3 * N planes of (n, theta, k) */
/* copies of the derived CoalescenceData fields (Coalescence.jl:69-84), for tests and hosts */
int cloudy_plan_get(const cloudy_plan *plan, int32_t *N_mom_max, int32_t *N_2d_ints /*[N]*/,
                    double *thresholds /*[N]*/, double *kernel_c_normalized /*[N][N][P][P]*/,
                    double *mom_norms /*[nmom]*/);

/* dmom = d(mom)/dt by collision-coalescence; mom, dmom: nmom planes, physical units. */
int cloudy_coal_rhs(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *mom_dev, void *dmom_dev,
                    void *stream);
/* same through host buffers (allocates a staging buffer, copies, synchronises): convenience only */
int cloudy_coal_rhs_host(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *mom_host,
                         void *dmom_host);

/* n_steps explicit SSPRK33 steps of du/dt = rhs!(u) with fixed dt, the integrator of every reference driver
 * (solve(prob, SSPRK33(), dt = ...), test/examples/Analytical/box_single_gamma.jl:35-36), fused around the RHS:
 * each parcel's moments stay in registers over all stages and steps (one read + one write of the state per call,
 * no per-stage launches).  u_out_dev may equal u_in_dev.  NumericalCoalStyle plans are served as well (the Numerical
 * drivers integrate the same way, test/examples/Numerical/n_particles_gamma.jl:39-40).  Plans without a finite threshold
 * keep the state in normalised units (mom ./ norms) between load and store: the stage values differ from rhs!'s
 * normalise-in-every-call sequence by roundings only. */
int cloudy_ssprk33_steps(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *u_in_dev, void *u_out_dev,
                         double dt, int n_steps, void *stream);

/* n_steps explicit steps of the Tsit5 tableau (Tsitouras 2011; OrdinaryDiffEq's Tsit5()) with FIXED dt, fused around the
 * RHS like cloudy_ssprk33_steps (state and the six stage derivatives in registers, 6 RHS evaluations per step, FSAL).
 * BASELINE configs[0] names Tsit5 for the single-box Golovin case; no reference driver uses it (all of them call
 * solve(prob, SSPRK33(), dt = ...)), and OrdinaryDiffEq's adaptive step control needs a global error norm over the state,
 * which a batch of independent parcels does not have: this is the tableau applied per parcel with the caller's dt.
 * Served for every plan cloudy_ssprk33_steps serves (round 4): AnalyticalCoalStyle with thresholds Inf (state kept in
 * normalised units between load and store), fixed or MovingThreshold, and NumericalCoalStyle in either quad_mode; fp64 or
 * float planes (CLOUDY_F32_FAST: CLOUDY_EUNSUPPORTED).  The kernel is compiled for the plan on the first call
 * (cloudy_jit_tsit5_* / cloudy_jit_quad_tsit5_*); without hiprtc the tensor plans run the ahead-of-time kernel and a
 * NumericalCoalStyle plan answers CLOUDY_EUNSUPPORTED. */
int cloudy_tsit5_steps(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *u_in_dev, void *u_out_dev,
                       double dt, int n_steps, void *stream);

/* inner operator on given distributions: params = 3N planes (n, theta, k) per mode, normalised units
 * (k plane ignored for exponential modes); out = nmom planes, normalised units. */
int cloudy_get_coal_ints(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *params_dev,
                         void *coal_ints_dev, void *stream);

/* Batched validity diagnostic of the closure inversion (round 6): counts_host[4 * mode + c] = the number of parcels whose mode
 *   c = 0: took the fallback distribution (0, 1, 1) (moments at or below eps; ParticleDistributions.jl:461, :473-475, :495, :519);
 *   c = 1: has its shape at the lower clamp (Gamma: k = k_range[0], :459-469; Lognormal: sigma = eps, :499-502);
 *   c = 2: has its shape at the upper clamp (Gamma: k = k_range[1]);
 *   c = 3: fails check_moment_consistency (ParticleDistributions.jl:437-449) -- a negative moment, or a negative second central
 *          moment -- evaluated on the normalised moments the closure sees (box_model_helpers.jl:30-31).
 * The reference can only raise per call or clamp silently; a batch of 1e7 parcels cannot throw per parcel, so the counts are the
 * batch form of both.  mom_dev: the plan's plane type, physical units, as cloudy_coal_rhs.  Synchronises `stream`; counts_host
 * has 4 * n_modes entries (host memory). */
int cloudy_closure_stats(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *mom_dev, uint64_t *counts_host,
                         void *stream);

/* closure inversion: mom (physical) -> params planes (n, theta, k), normalised units */
int cloudy_update_dist_from_moments(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *mom_dev,
                                    void *params_dev, void *stream);

/* F[i][p1][p2] of get_finite_2d_integrals: N*M*M planes (M = P+2), row-major (i, p1, p2) */
int cloudy_finite_2d_integrals(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *params_dev,
                               void *F_dev, void *stream);

/* per-parcel thresholds actually used by the S terms: N planes (last = +Inf) */
int cloudy_compute_thresholds(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *params_dev,
                              void *thresholds_dev, void *stream);

/* sedimentation flux of every prognostic moment, physical units (plan must have n_vel > 0) */
int cloudy_sedimentation_flux(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *mom_dev,
                              void *flux_dev, void *stream);

/* condensation / evaporation tendency of every prognostic moment, rhs_condensation!(dmom, mom, p, s)
 * (test/examples/utils/box_model_helpers.jl:55-67 -> get_cond_evap, src/Sources/Condensation.jl:22-37).
 * xi = p.xi in physical units; supersaturation s per parcel (s_dev, fp64, n values) or, if s_dev is NULL, `s`. */
int cloudy_cond_evap(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *mom_dev, const double *s_dev,
                     double s, double xi, void *dmom_dev, void *stream);

/* cloud / rain diagnostics get_standard_N_q(pdists, size_cutoff) (ParticleDistributions.jl:634-687): 4 planes
 * (N_liq, N_rai, M_liq, M_rai) in physical units; size_cutoff in physical mass units (the examples pass 1e-6 kg). */
int cloudy_standard_N_q(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *mom_dev, double size_cutoff,
                        void *nq_dev, void *stream);

/* coalescence source and sedimentation flux of each cell in one fused pass
 * (negative moments clamped to zero, empty cells skipped) */
int cloudy_rainshaft_sources(const cloudy_plan *plan, size_t n_cells, size_t ld, const void *mom_dev,
                             void *coal_source_dev, void *sedi_flux_dev, void *stream);

/* the full right-hand side of make_rainshaft_rhs (rainshaft_helpers.jl:45-89) for n_columns columns of nz cells
 * (cell index fastest, bottom to top): coalescence source + upwind divergence of the sedimentation flux,
 * rhs[i] = coal[i] - (flux[i+1] - flux[i]) / dz with zero flux above the top cell.  flux_work_dev: nmom planes of
 * scratch (holds the cell fluxes on return).  One launch for columns of up to 1024 cells when the plan's kernels are compiled
 * for it (a workgroup holds whole columns, the flux of the cell above travels through LDS), the cell kernel and a divergence
 * launch otherwise; the same bits on fp64 planes (float planes: the one launch does not round the fluxes in between). */
int cloudy_rainshaft_rhs(const cloudy_plan *plan, size_t nz, size_t n_columns, size_t ld, const void *mom_dev, double dz,
                         void *flux_work_dev, void *rhs_dev, void *stream);

/* n_steps SSPRK33 steps of the rainshaft right-hand side above for n_columns independent columns of nz <= 1024 cells
 * (what `solve(ODEProblem(rhs, m, tspan, p), SSPRK33(), dt = p.dt)` does in rainshaft_single_gamma.jl:52-53,
 * rainshaft_gamma_mixture.jl:59-60), in ONE launch: a workgroup owns whole columns, the state stays in registers over
 * all stages and steps (parked in LDS across the Simpson passes of a thresholded plan) and the upwind flux of the cell above
 * is exchanged through LDS.  As in the reference, every RHS
 * evaluation first clamps negative moments of its argument to zero in place (rainshaft_helpers.jl:52), including the
 * FSAL evaluation on each step's result, so the returned state is clamped.  u_out_dev may equal u_in_dev.
 * A workgroup of 256, 512 or 1024 threads holds floor(threads / nz) whole columns; the kernel compiled for the plan picks the
 * size by nz (512 threads for the reference's 20 cells; results do not depend on it), the ahead-of-time kernel has 256 threads
 * and nz <= 256 (256 < nz <= 1024 without hiprtc: CLOUDY_EUNSUPPORTED).  nz > 1024 (round 5; the reference's cell
 * loop is unbounded in nz, rainshaft_helpers.jl:55-78): the same steps stage by stage on the stream -- cloudy_rainshaft_rhs and
 * one update launch per stage, three stream-ordered scratch arrays of the state's size per call -- not fused, any nz. */
int cloudy_rainshaft_ssprk33_steps(const cloudy_plan *plan, size_t nz, size_t n_columns, size_t ld,
                                   const void *u_in_dev, void *u_out_dev, double dz, double dt, int n_steps,
                                   void *stream);

/* sums_dev[q] = sum over parcels of plane q (fp64 accumulate); `planes` planes are reduced.
 * The multi-GPU conservation check all-reduces these nmom doubles (RCCL), see INTEGRATION.md. */
int cloudy_moment_sums(const cloudy_plan *plan, size_t n_parcels, size_t ld, int planes, const void *arr_dev,
                       double *sums_dev, void *stream);
/* the same with a caller-supplied workspace of cloudy_moment_sums_workspace_bytes(planes) bytes of device memory:
 * re-entrant (any number of streams may reduce with one plan concurrently, each with its own workspace) */
size_t cloudy_moment_sums_workspace_bytes(int planes);
int cloudy_moment_sums_ws(const cloudy_plan *plan, size_t n_parcels, size_t ld, int planes, const void *arr_dev,
                          double *sums_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- multi-GPU: the one collective of the path ----------------------------------------------------------
 * Parcels are independent: ranks (one process per GPU, or one process driving several GPUs) own contiguous parcel ranges
 * and exchange nothing per right-hand side.  The only exchange is the sum over ranks of the nmom plane sums of the
 * conservation diagnostic (moments_sum, plotting_helpers.jl:240-252): ncclAllReduce(sum, ncclDouble, planes values) over
 * RCCL / xGMI, issued on the caller's stream.  RCCL is bound at first use (dlopen librccl.so.1): hosts that never create a
 * communicator do not load it.  CLOUDY_EUNSUPPORTED if RCCL is absent, CLOUDY_ECOMM on an RCCL error.
 *
 * One process per GPU: rank 0 calls cloudy_comm_unique_id, the host hands the 128 bytes to every rank (MPI.bcast, a
 * file, a socket -- the library does no rendezvous of its own), every rank calls cloudy_comm_create (collective).
 * One process, several GPUs: cloudy_comm_create_all, and bracket the per-GPU cloudy_moment_sums_allreduce calls with
 * cloudy_comm_group_start / _end as RCCL requires for one thread driving several communicators. */
#define CLOUDY_COMM_ID_BYTES 128
typedef struct cloudy_comm cloudy_comm;
int cloudy_comm_rccl_version(void);                 /* NCCL_VERSION_CODE of the RCCL bound, 0 if none */
int cloudy_comm_unique_id(void *id_out /*[CLOUDY_COMM_ID_BYTES]*/);
/* device: HIP ordinal the communicator binds to, -1 = current */
int cloudy_comm_create(int world_size, int rank, const void *id /*[CLOUDY_COMM_ID_BYTES]*/, int device, cloudy_comm **out);
/* devices: n_devices ordinals, NULL = 0 .. n_devices-1; comms_out[i] has rank i */
int cloudy_comm_create_all(int n_devices, const int *devices, cloudy_comm **comms_out /*[n_devices]*/);
void cloudy_comm_destroy(cloudy_comm *comm);
int cloudy_comm_rank(const cloudy_comm *comm);
int cloudy_comm_world_size(const cloudy_comm *comm);
int cloudy_comm_device(const cloudy_comm *comm);
int cloudy_comm_group_start(void);
int cloudy_comm_group_end(void);
/* recv_dev[i] = sum over ranks of send_dev[i], i < count; in place allowed; asynchronous on `stream` */
int cloudy_allreduce_sum_f64(cloudy_comm *comm, const double *send_dev, double *recv_dev, size_t count, void *stream);
/* cloudy_moment_sums of this rank's parcels followed by the all-reduce, in place in sums_dev, both on `stream`:
 * sums_dev[q] = sum over ALL ranks' parcels of plane q.  With world_size 1 the result is cloudy_moment_sums' bit for bit. */
int cloudy_moment_sums_allreduce(const cloudy_plan *plan, cloudy_comm *comm, size_t n_parcels, size_t ld, int planes,
                                 const void *arr_dev, double *sums_dev, void *stream);

/* thin device-memory helpers so that a host without a HIP binding can keep state device-resident */
int cloudy_device_count(void);
int cloudy_set_device(int device);
/* "dddd:bb:dd.f" of a device ordinal (hipDeviceGetPCIBusId), len >= 13: bench.py --gpus N records it per rank so that a
 * scaling record shows N distinct devices */
int cloudy_device_pci_bus_id(int device, char *buf, int len);
int cloudy_malloc(void **dev_ptr, size_t bytes);
int cloudy_free(void *dev_ptr);
int cloudy_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int cloudy_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int cloudy_memset(void *dst_dev, int value, size_t bytes, void *stream);
int cloudy_stream_synchronize(void *stream);

/* launch-timing helper used by bench.py: runs `iters` back-to-back cloudy_coal_rhs launches on `stream`
 * bracketed by HIP events recorded on that stream; returns the average milliseconds per launch. */
int cloudy_time_coal_rhs(const cloudy_plan *plan, size_t n_parcels, size_t ld, const void *mom_dev,
                         void *dmom_dev, void *stream, int iters, float *ms_per_launch);

/* generic launch timing with HIP events recorded on `stream` (the stream the kernels are launched on): begin records an
 * event and hands out a timer, end records the second event, waits for it, returns the milliseconds between the two and
 * releases the timer.  bench.py brackets every variant that is not a plain cloudy_coal_rhs loop with these. */
int cloudy_timer_begin(void *stream, void **timer_out);
int cloudy_timer_end(void *timer, void *stream, float *ms_total);

const char *cloudy_last_error(void);
int cloudy_version(void);
/* FNV-1a of the kernel sources inside the library (which = 0: all of them; 1: those of the all-Inf kernels, the headline of
 * bench.py): a committed profile figure is quoted only for the sources it was collected on */
unsigned long long cloudy_source_hash(int which);

#ifdef __cplusplus
}
#endif
#endif /* CLOUDY_HIP_H */
