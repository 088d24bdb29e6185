// quad.hpp -- NumericalCoalStyle plans: the coalescence integrals of a kernel FUNCTION K(x, y) by a fixed Gauss rule.
//
// Reference: get_coal_ints(::NumericalCoalStyle, pdists, kernel_func), src/Sources/Coalescence.jl:470-489, behind
// make_box_model_rhs(NumericalCoalStyle()) (test/examples/utils/box_model_helpers.jl:22-53).  The reference evaluates
// its Q, R, S integrals (:503-622, integrands :644-708, weighting_fn :624-642) with nested ADAPTIVE quadgk; here every
// integral against a density is ONE fixed nq-point Gauss rule for that density, after the substitution x' = x - y that
// maps the triangular domain of Q and S onto (0, inf)^2:
//     Q_jk^(m) =     sum_ab  Kab (x^j_a + x^k_b)^m             Kab = K(x^j_a, x^k_b) (n_j W^j_a)(n_k W^k_b)
//     R_jk^(m) =     sum_ab  Kab (x^k_b)^m                     (the sink of mode k against mode j)
//     S_k^(m)  = 1/2 sum_ab  Kab (x^k_a + x^k_b)^m {w, 1 - w}(x^k_a + x^k_b)
// Gamma / Exponential modes: generalised Gauss-Laguerre in u = x / theta for the weight u^(k-1) e^-u; Lognormal modes:
// Gauss-Hermite in (ln x - mu) / (sqrt(2) sigma).  Exact for polynomial kernels (the rule integrates degree 2 nq - 1),
// a discretisation for the hydrodynamic and Long kernels (DESIGN.md states the measured error).
//
// The Laguerre rule depends on the shape parameter k, which is per parcel.  What is staged once per plan is a table of
// start values: the nodes u_a(k), a = 1..nq, are analytic in k on [0, k_max], so a Chebyshev fit of modest degree
// (u_1 / k for the node that vanishes with k) gives them to ~1e-5 of the node spacing; each lane then polishes its own
// nodes with two Newton steps on L_nq^(k-1) (quadratic: 1e-5 -> 1e-10 -> 1e-20) and takes the Christoffel weights
// from the same recurrence.  The table is read at wave-uniform addresses (compile-time constants when the kernel is
// compiled for its plan); nothing is searched or iterated to convergence, every lane does identical work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.hpp"

namespace cloudy {

enum { KF_CONSTANT = 0, KF_LINEAR = 1, KF_HYDRODYNAMIC = 2, KF_LONG = 3 };
constexpr int kQuadMax = 32;     // points per rule
constexpr int kQuadDegMax = 24;  // degree of the start-value polynomials

enum { QUAD_FIXED = 0, QUAD_CONVERGED = 1 };  // cloudy_plan_desc.quad_mode
struct QArgs {  // wave-uniform constants of a NumericalCoalStyle plan
    // mode QUAD_FIXED: nq points of the per-distribution Gauss rule; QUAD_CONVERGED (quad_conv.hpp): nq Gauss-Legendre
    // points per panel of the inner rule of a Lognormal mode's T_m, table = nq nodes on [-1, 1] then nq weights
    int32_t kind, nq, deg, mode;
    double kf[3];     // normalised kernel-function parameters (get_normalized_kernel_func, KernelFunctions.jl:124-154)
    double t_scale;   // t = k * t_scale - 1 maps [0, k_hi] onto [-1, 1]
};
// table layout (doubles): [nq][deg + 1] monomial coefficients in t, highest power first (row 0 fits u_1 / k);
// then nq Gauss-Hermite nodes and nq normalised Gauss-Hermite weights (Lognormal modes)
__host__ __device__ constexpr int quad_tab_size(int nq, int deg) { return nq * (deg + 1) + 2 * nq; }
// workgroup size of the plan-time compiled kernel: its self-collision stage parks one mode's nodes in per-lane LDS
// slots (2 nq doubles per lane), which must fit the 64-KiB static limit
__host__ __device__ constexpr int quad_block(int nq) { return nq <= 12 ? 256 : nq <= 24 ? 128 : 64; }

// Scheduling fence (device): the unrolled per-node work of a rule, and the per-row work of a pair sum, are independent
// dependency chains; left alone the scheduler interleaves all of them and the kernel needs 350-500 VGPRs (measured:
// 512 + 747 spilled for the Long family).  One chain at a time keeps the kernel inside 256.
__host__ __device__ __forceinline__ void quad_sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

__host__ __device__ __forceinline__ double quad_recip(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return recip_fast(x);
#else
    return 1.0 / x;  // host instance of the same template: cloudy_quad_rule_host, a CPU-side check of this algorithm
#endif
}

// L_n^(k-1)(x) and L_{n-1}^(k-1)(x): (j+1) L_{j+1} = (2j + k - x) L_j - (j - 1 + k) L_{j-1}
template <int NQ>
__host__ __device__ __forceinline__ void laguerre_pair(int nq_rt, double k, double x, double &Ln, double &Lm) {
    const int nq = NQ ? NQ : nq_rt;
    double p0 = 1.0, p1 = k - x;
    const double kx = k - x;
#pragma unroll
    for (int j = 1; j < nq; ++j) {
        const double p2 = (fma(2.0, double(j), kx) * p1 - (double(j - 1) + k) * p0) * (1.0 / double(j + 1));
        p0 = p1;
        p1 = p2;
    }
    Ln = p1;
    Lm = p0;
}

// nq-point Gauss rule for the normalised weight u^(k-1) e^-u / Gamma(k): u[a], W[a] (sum_a W[a] = 1).
// NQ > 0: compile-time point count (the arrays live in registers); NQ == 0: run-time nq <= kQuadMax.
template <int NQ>
__host__ __device__ __forceinline__ void gamma_rule(const QArgs &Q, const double *__restrict__ tab, double k, double *u,
                                                    double *W) {
    const int nq = NQ ? NQ : Q.nq;
    const int deg = Q.deg;
    const double t = fma(k, Q.t_scale, -1.0);
    // The start-value polynomials are a fit over k in (0, k_hi] (k_hi = max(k_range[1], 1), the range the closure
    // inversion can produce).  Parameters handed in directly (cloudy_get_coal_ints) may lie outside: extrapolated start
    // values converge to wrong nodes silently, so such a mode gets a NaN rule -- its tendencies are NaN, not wrong.
    const bool k_ok = k > 0.0 && t <= 1.0;
    double Cnk = 1.0;  // Gamma(nq + k) / (Gamma(k) nq!) = prod_j (k + j) / (j + 1)
#pragma unroll
    for (int j = 0; j < nq; ++j) Cnk *= (k + double(j)) * (1.0 / double(j + 1));
    const double nk = double(nq - 1) + k;
#pragma unroll
    for (int a = 0; a < nq; ++a) {
        const double *c = tab + a * (deg + 1);
        double x = c[0];
        for (int d = 1; d <= deg; ++d) x = fma(x, t, c[d]);
        if (a == 0) x *= k;
        double Ln, Lm, D;
#pragma unroll
        for (int it = 0; it < 2; ++it) {  // Newton: x L_n'(x) = n L_n - (n - 1 + k) L_{n-1}
            laguerre_pair<NQ>(nq, k, x, Ln, Lm);
            D = fma(double(nq), Ln, -(nk * Lm));
            x = fma(-x, Ln * quad_recip(D), x);
        }
        laguerre_pair<NQ>(nq, k, x, Ln, Lm);
        D = fma(double(nq), Ln, -(nk * Lm));
        const double rD = quad_recip(D);
        W[a] = k_ok ? Cnk * x * (rD * rD) : __builtin_nan("");  // Gamma(n+k) / (Gamma(k) n! x L_n'(x)^2)
        u[a] = k_ok ? fma(-x, Ln * rD, x) : __builtin_nan("");  // (the third Newton update comes free with the weight)
        quad_sched_fence();
    }
}

// ---- kernel functions K(x, y), KernelFunctions.jl:94-116, on per-node auxiliaries ---------------------------
// A node carries `aux` (what K needs of x) and x itself is recovered from it:
//   hydrodynamic: aux = x^(1/3);  K = E pi (3/(4 pi))^(4/3) (rx + ry)^2 |rx^2 - ry^2|   [= E (r1 + r2)^2 |A1 - A2|]
//   others:       aux = x
template <int KIND>
__device__ __forceinline__ double kf_aux(double x) {
    return KIND == KF_HYDRODYNAMIC ? cbrt(x) : x;
}
template <int KIND>
__device__ __forceinline__ double kf_x(double aux) {
    return KIND == KF_HYDRODYNAMIC ? aux * aux * aux : aux;
}
// K up to the constant factor kf_scale (applied once to the finished sums)
template <int KIND>
__device__ __forceinline__ double kf_eval(const QArgs &Q, double a, double b) {
    if (KIND == KF_CONSTANT) return 1.0;
    if (KIND == KF_LINEAR) return a + b;
    if (KIND == KF_HYDRODYNAMIC) {
        const double s = a + b;
        return (s * s) * fabs(fma(a, a, -(b * b)));
    }
    // Long: rate_below (x^2 + y^2) if both below the threshold, else rate_above (x + y)
    return (a < Q.kf[0] && b < Q.kf[0]) ? Q.kf[1] * fma(a, a, b * b) : Q.kf[2] * (a + b);
}
template <int KIND>
__device__ __forceinline__ double kf_scale(const QArgs &Q) {
    if (KIND == KF_HYDRODYNAMIC) return Q.kf[0] * 0.46526286817455001;  // pi (3 / (4 pi))^(4/3)
    if (KIND == KF_LONG) return 1.0;
    return Q.kf[0];
}

}  // namespace cloudy
