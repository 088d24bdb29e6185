// device_math.hpp -- fp64 special functions for the gfx950 coalescence kernels.
//
// Everything here runs one value per wavefront lane.  The incomplete-gamma routines are written for
// SIMT execution: division-free inner loops (one divide after the loop), convergence tested every
// four terms so that a wave leaves the loop soon after its slowest lane has converged.
//
// Replaces, on device, the SpecialFunctions.jl calls of the reference hot path
// (src/ParticleDistributions/ParticleDistributions.jl:575-577, 597-602, 760).
#pragma once
#include <hip/hip_runtime.h>

namespace cloudy {

constexpr double kEps = 2.220446049250313e-16;          // eps(Float64)
constexpr double kSqrtEps = 1.4901161193847656e-08;     // 2^-26: x >= 2^-26 && y >= 2^-26  =>  x*y >= eps

// 1/x for the per-node quotients of the Simpson pass (1/z, N'/D', A/B of the continued fraction): hardware reciprocal
// estimate + two Newton steps, ~1 ulp, 5 instructions -- without the scaling / fix-up sequence of an IEEE division
// (11-12 instructions) whose only purpose is denormal and overflow ranges these operands never reach
// (z = (x_t - x)/theta, D' in [1, 1e17], |B| <= 1e150 by the rescaling in the loop).
__device__ __forceinline__ double recip_fast(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// Regularised lower incomplete gamma P(a, z) for a > 0, z > 0, given
//   E = z^a e^-z / Gamma(a + 1)
// (the caller has E from one exp()).  z <= a+1: power series P = E * sum_n z^n / (a+1)_n evaluated as
// a ratio N_n / D_n (no division per term).  z > a+1: Legendre continued fraction for Q = 1 - P by the
// Wallis forward recurrence (no division per term).  *q_out gets Q on the same branch's accuracy.
// Stopping tolerances (relative size of the last term / of the last change of the continued fraction), tested every four
// terms.  Rounds 1-3 used 1e-17 / 1e-16 -- BELOW the rounding unit, so both loops ran one or two groups of four past the
// point where double precision stops changing.  Measured in round 4 (tools/timeit.py kernels --error; cfg3b / cfg4 / moving4,
// error against the oracle in units of the term scale):
//   1e-17 / 1e-16   2.05 / 6.56 / 1.375 ms   1.8e-14 / 1.9e-14 / 7.5e-15
//   1e-15           1.85 / 6.05 / 1.291 ms   the same errors to both printed digits
//   1e-14           1.83 / 5.98 / 1.281 ms   1.8e-14 / 1.9e-14 / 7.6e-15      <- the default
//   1e-13                                    the moving4 error doubles
// (the tail left behind a last term of 1e-14 is of that size: by then the term ratio is well below 1.)  CLOUDY_F64_RELAXED
// plans define both as 1e-11.
#ifndef CLOUDY_SERIES_TOL
#define CLOUDY_SERIES_TOL 1e-14
#endif
#ifndef CLOUDY_CF_TOL
#define CLOUDY_CF_TOL 1e-14
#endif
__device__ __forceinline__ double inc_gamma_p_from_E(double a, double z, double E, double *q_out) {
    if (z <= a + 1.0) {
        // S = sum_n z^n/(a+1)_n = N_n/D_n with both scaled by z^-n:  q_n = (a+n)/z,
        //   N'_n = N'_{n-1} q_n + 1,  D'_n = D'_{n-1} q_n;   term_n / S_n = 1 / N'_n   (3 VALU ops per term)
        const double invz = recip_fast(z);
        double q = a * invz, Nn = 1.0, Dn = 1.0;
#pragma unroll 1
        for (int it = 0; it < 100; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                q += invz;
                Nn = fma(Nn, q, 1.0);
                Dn *= q;
            }
            if (!(Nn < 1.0 / CLOUDY_SERIES_TOL)) break;  // D' <= N': no overflow before convergence
        }
        double p = E * (Nn * recip_fast(Dn));
        p = p > 1.0 ? 1.0 : p;
        if (q_out) *q_out = 1.0 - p;
        return p;
    } else {
        // Q = a E f with f < 1/2 for z > a+1: below 1e-18 it cannot change P = 1 - Q (nor any lower order)
        if (a * E < 1e-18) {
            if (q_out) *q_out = a * E * (1.0 / (z + 1.0 - a));
            return 1.0;
        }
        // f = a1/(b1 + a2/(b2 + ...)), a1 = 1, b1 = z+1-a, a_n = -(n-1)(n-1-a), b_n = b_{n-1} + 2
        double b = z + 1.0 - a;
        double Ap = 0.0, Bp = 1.0, Ac = 1.0, Bc = b;
        // a_{i+1} - a_i = -(2i + 1 - a) = z - b_i: the partial numerators follow by additions (7 VALU ops per step)
        double an = a - 1.0, c = a - 1.0;  // a_1 = -(1 - a); c_i = z - b_i, c_0 = a - 1
#pragma unroll 1
        for (int it = 0; it < 100; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                b += 2.0;
                const double An = fma(b, Ac, an * Ap);
                const double Bn = fma(b, Bc, an * Bp);
                Ap = Ac;
                Bp = Bc;
                Ac = An;
                Bc = Bn;
                c -= 2.0;
                an += c;
            }
            // f_n - f_{n-1} = (Ac*Bp - Ap*Bc) / (Bc*Bp)
            double lhs = fabs(fma(Ac, Bp, -(Ap * Bc)));
            if (!(lhs > CLOUDY_CF_TOL * fabs(Ac * Bp))) break;
            if (fabs(Bc) > 1e150) {
                Ap *= 1e-150;
                Bp *= 1e-150;
                Ac *= 1e-150;
                Bc *= 1e-150;
            }
        }
        double q = a * E * (Ac * recip_fast(Bc));
        q = q < 0.0 ? 0.0 : q;
        if (q_out) *q_out = q;
        return 1.0 - q;
    }
}

// fp32 variant of inc_gamma_p_from_E for the CLOUDY_F32_FAST plans: same two algorithms, single precision,
// convergence to ~1e-7 (a third of the terms), v_rcp_f32-class divisions.
__device__ __forceinline__ float inc_gamma_p_from_E_f32(float a, float z, float E) {
    if (z <= a + 1.0f) {
        if (z < 1e-3f * (a + 1.0f))  // two terms suffice (rel. error < 1e-9) and (a+n)/z would overflow single precision
            return E * (1.0f + (z / (a + 1.0f)) * (1.0f + z / (a + 2.0f)));
        const float invz = 1.0f / z;
        float q = a * invz, Nn = 1.0f, Dn = 1.0f;
#pragma unroll 1
        for (int it = 0; it < 50; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                q += invz;
                Nn = fmaf(Nn, q, 1.0f);
                Dn *= q;
            }
            if (!(Nn < 2.0e7f)) break;  // term/sum = 1/N' < 5e-8; D' <= N': no overflow before convergence
        }
        const float p = E * (Nn / Dn);
        return p > 1.0f ? 1.0f : p;
    } else {
        if (a * E < 1e-9f) return 1.0f;
        float b = z + 1.0f - a;
        float Ap = 0.0f, Bp = 1.0f, Ac = 1.0f, Bc = b;
        float an = a - 1.0f, c = a - 1.0f;
#pragma unroll 1
        for (int it = 0; it < 50; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                b += 2.0f;
                const float An = fmaf(b, Ac, an * Ap);
                const float Bn = fmaf(b, Bc, an * Bp);
                Ap = Ac;
                Bp = Bc;
                Ac = An;
                Bc = Bn;
                c -= 2.0f;
                an += c;
            }
            const float lhs = fabsf(fmaf(Ac, Bp, -(Ap * Bc)));
            if (!(lhs > 1e-7f * fabsf(Ac * Bp))) break;
            if (fabsf(Bc) > 1e15f) {
                Ap *= 1e-15f;
                Bp *= 1e-15f;
                Ac *= 1e-15f;
                Bc *= 1e-15f;
            }
        }
        float qv = a * E * (Ac / Bc);
        qv = qv < 0.0f ? 0.0f : qv;
        return 1.0f - qv;
    }
}

__device__ __forceinline__ double log_pos(double x);
__device__ __forceinline__ double lgamma_pos(double k);
__device__ __forceinline__ double exp_fin(double x);

// P(a, z), Q(a, z) for a standalone call (one exp; lz = ln z)
__device__ __forceinline__ double inc_gamma_p(double a, double z, double lz, double lgamma_a1, double *q_out) {
    if (!(z > 0.0)) {
        if (q_out) *q_out = 1.0;
        return 0.0;
    }
    double E = exp_fin(fma(a, lz, -z) - lgamma_a1);
    return inc_gamma_p_from_E(a, z, E, q_out);
}

// x with P(a, x) = p, Q(a, x) = q  (SpecialFunctions.gamma_inc_inv; ParticleDistributions.jl:760).
// Safeguarded Halley iteration on the smaller tail.
__device__ inline double inc_gamma_inv(double a, double p, double q, double x_start = 0.0) {
    if (!(p > 0.0)) return 0.0;
    if (!(q > 0.0)) return INFINITY;
    const double a1 = a - 1.0;
    const double lga1 = lgamma_pos(a + 1.0);  // log Gamma(a+1)
    const double lga = lga1 - log_pos(a);     // log Gamma(a)
    double x;
    if (x_start > 0.0) {  // the caller's start value (moving_threshold: a staged fit, good to ~1e-7)
        x = x_start;
    } else if (a > 1.0) {
        double pp = (p < 0.5) ? p : q;
        double t = sqrt(-2.0 * log(pp));
        double zz = (2.30753 + t * 0.27061) / (1.0 + t * (0.99229 + t * 0.04481)) - t;
        if (p < 0.5) zz = -zz;
        double w = 1.0 - 1.0 / (9.0 * a) - zz / (3.0 * sqrt(a));
        x = a * w * w * w;
        if (x < 1e-3) x = 1e-3;
    } else {
        double t = 1.0 - a * (0.253 + a * 0.12);
        // x ~ (p Gamma(a+1))^(1/a) for small a: below the smallest double the answer is 0 (callers clamp it from below;
        // without this exit a shape clamped to k = eps spends the full iteration budget bisecting towards 0 while the
        // other 63 parcels of its wave wait)
        if (p < t && log(p / t) < -744.0 * a) return 0.0;
        if (p < t)
            x = pow(p / t, 1.0 / a);
        else
            x = 1.0 - log(1.0 - (p - t) / (1.0 - t));
    }
    double lo = 0.0, hi = INFINITY, prev_step = INFINITY;
#pragma unroll 1
    for (int it = 0; it < 100; ++it) {
        if (!(x > 0.0)) x = (hi < INFINITY) ? 0.5 * (lo + hi) : 1e-300;
        double Q;
        const double lx = log_pos(x);
        double Pv = inc_gamma_p(a, x, lx, lga1, &Q);
        double err = (p <= 0.5) ? (Pv - p) : (q - Q);
        if (err > 0.0) {
            if (x < hi) hi = x;
        } else if (err < 0.0) {
            if (x > lo) lo = x;
        } else {
            return x;
        }
        double dens = exp_fin(fma(a1, lx, -x) - lga);
        double xn;
        if (dens > 0.0 && dens < INFINITY) {
            double u = err / dens;
            double corr = u / (1.0 - 0.5 * fmin(1.0, u * (a1 / x - 1.0)));
            xn = x - corr;
        } else {
            xn = -1.0;
        }
        if (!(xn > lo) || !(xn < hi)) xn = (hi < INFINITY) ? 0.5 * (lo + hi) : 2.0 * x;
        double step = fabs(xn - x);
        if (step <= 4.0 * kEps * fabs(xn)) return xn;
        // a start value from the staged fit is good to ~1e-7: one Halley step (third order) then leaves ~1e-18
        if (x_start > 0.0 && it == 0 && step <= 1e-6 * fabs(xn) && xn > lo && xn < hi) return xn;
        // rounding-level limit cycle: the step stopped shrinking and is already negligible
        if (step >= prev_step && step <= 1e-13 * fabs(xn)) return xn;
        prev_step = step;
        x = xn;
    }
    return x;
}

// e^x for finite or NaN x (never -Inf: the callers' arguments are differences of finite logarithms), < 1 ulp
// (1.7e-16 relative over [-700, 700]), 19 VALU instructions where the library routine has 31 (it selects 0 / Inf around
// its result for arguments outside [-745, 709.8]; here v_ldexp_f64 saturates the same way).  x = n ln 2 + r with a split
// ln 2 whose high part times n is exact, then a degree-11 polynomial fitted on [-ln2/2, ln2/2] (4.3e-18 in exact
// arithmetic).  Used per Simpson node and per quadrature point.
__device__ __forceinline__ double exp_fin(double x) {
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = fma(n, -0.69314718055989033, x);   // 0x3FE62E42FEFA3800: 21 trailing zero bits
    r = fma(n, -5.497923018708371e-14, r);
    double p = 0x1.adeb8db5d7212p-26;
    p = fma(p, r, 0x1.28afdbfa89bf0p-22);
    p = fma(p, r, 0x1.71dedfc117959p-19);
    p = fma(p, r, 0x1.a019970598987p-16);
    p = fma(p, r, 0x1.a01a014a32d85p-13);
    p = fma(p, r, 0x1.6c16c18581530p-10);
    p = fma(p, r, 0x1.1111111121b01p-7);
    p = fma(p, r, 0x1.55555555500b2p-5);
    p = fma(p, r, 0x1.5555555555513p-3);
    p = fma(p, r, 0x1.000000000000bp-1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// ln x for finite x > 0, ~1 ulp, ~32 VALU instructions (the library log is ~95: it carries a double-double quotient
// and the special-case selects): x = 2^e m with m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh(s) = 2 s (1 + z/3 + z^2/5 + ...),
// s = (m - 1)/(m + 1), z = s^2 <= 0.0295 (nine terms: truncation 2e-17); e ln 2 added with a split constant.
// Used where a kernel takes many logarithms of positive finite arguments (quad_kernels.hpp: one per quadrature point).
__device__ __forceinline__ double log_pos(double x) {
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);  // [1/2, 1)
    const bool low = m < 0.70710678118654752;
    m = low ? m + m : m;
    e -= low ? 1 : 0;
    const double f = m - 1.0;  // exact
    const double s = f * recip_fast(m + 1.0);
    const double z = s * s;
    double p = 1.0 / 19.0;
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    const double ed = (double)e;
    // ln 2 = 0x3FE62E42FEFA3800 (hi, 21 trailing zero bits: e * hi is exact) + 5.497923018708371e-14
    const double lo = fma(ed, 5.497923018708371e-14, (s + s) * (z * p));
    return fma(ed, 0.69314718055989033, (s + s) + lo);
}

// ln Gamma(k) for k > 0 without the library routine's range branches (~1100 instructions of code, lane-divergent):
// arguments below 10 are shifted up, ln Gamma(k) = ln Gamma(k + 10) - ln(k (k+1) ... (k+9)), and ln Gamma(z), z >= 10,
// is Stirling's series (z - 1/2) ln z - z + ln(2 pi)/2 + S(z) (stirling_tail: truncation 3e-17 at z = 10).  Absolute
// error <= 3 ulp of max(|result|, 20) against 40-digit mpmath over k in [1e-16, 60] (tools check; relative accuracy is
// lost near the zeros k = 1, 2, which the callers -- differences of log densities -- do not need).
__device__ __forceinline__ double stirling_tail(double z);
__device__ __forceinline__ double lgamma_pos(double k) {
    const bool shift = k < 10.0;
    double prod = 1.0;
#pragma unroll
    for (int i = 0; i < 10; ++i) prod *= shift ? k + double(i) : 1.0;
    const double z = shift ? k + 10.0 : k;
    const double st = fma(z - 0.5, log_pos(z), -z) + (0.91893853320467274 + stirling_tail(z));
    return shift ? st - log_pos(prod) : st;
}

// ln Gamma(k + v) - ln Gamma(k) for k > 0, v >= 0: the log of the fractional-moment factor Gamma(q + k) / Gamma(k)
// of moment(dist, q) (ParticleDistributions.jl:177-191) behind the sedimentation and condensation sources.
// Both arguments are shifted up by 10 when k < 10 -- Gamma(k+v)/Gamma(k) = Gamma(k+10+v)/Gamma(k+10) x
// prod_i (k+i)/(k+v+i) -- and the Stirling series is differenced analytically,
//   (z - 1/2) log1p(v/z) + v ln(z + v) - v + S(z + v) - S(z),   S(z) = sum_j B_2j / (2j (2j-1) z^(2j-1)),
// so no term is larger than O(v ln z): measured against 40-digit mpmath over k in [1e-16, 40] and
// v in [0.01, 3.2] the absolute error is <= 7.1e-15 (1.8e-15 for k > 1e-3), where the difference of two libm
// lgamma calls gives 3-4e-14 -- at about a quarter of the instructions.
__device__ __forceinline__ double stirling_tail(double z) {
    const double r = recip_fast(z), r2 = r * r;
    double p = -3617.0 / 122400.0;
    p = fma(p, r2, 1.0 / 156.0);
    p = fma(p, r2, -691.0 / 360360.0);
    p = fma(p, r2, 1.0 / 1188.0);
    p = fma(p, r2, -1.0 / 1680.0);
    p = fma(p, r2, 1.0 / 1260.0);
    p = fma(p, r2, -1.0 / 360.0);
    p = fma(p, r2, 1.0 / 12.0);
    return p * r;
}

__device__ __forceinline__ double log_gamma_ratio(double k, double v) {
    const bool shift = k < 10.0;
    const double kv = k + v;
    double num = 1.0, den = 1.0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        num *= shift ? k + double(i) : 1.0;
        den *= shift ? kv + double(i) : 1.0;
    }
    const double z = shift ? k + 10.0 : k, zv = z + v;
    // (log_pos / recip_fast: ~1 ulp each, a third of the library routines' instructions; log1p stays the library's:
    // ln(1 + v/z) through log_pos would lose the small-v digits)
    return fma(v, log_pos(zv), -v) + (z - 0.5) * log1p(v * recip_fast(z)) + (stirling_tail(zv) - stirling_tail(z)) +
           log_pos(num * recip_fast(den));
}

}  // namespace cloudy
