// quad_conv.hpp -- NumericalCoalStyle plans in CONVERGED mode (cloudy_plan_desc.quad_mode = CLOUDY_QUAD_CONVERGED).
//
// Reference: get_coal_ints(::NumericalCoalStyle, pdists, kernel_func), src/Sources/Coalescence.jl:470-708 -- nested adaptive
// quadgk(rtol = 1e-8) over the densities.  The fixed Gauss rule of quad.hpp converges only algebraically for the
// hydrodynamic kernel (|A1 - A2| kink on x = y, KernelFunctions.jl:102-108) and the Long kernel (jump at x_threshold,
// :110-116): 2e-3 of scale at 10 points.  Here the integrals are split along those non-smooth sets -- and on each side
// every CoalescenceKernelFunction of the reference is a short sum of SEPARABLE power terms c x^alpha y^beta:
//     constant  c;      linear  c (x + y);
//     hydrodynamic, x > y:  C (x^4/3 + 2 x y^1/3 - 2 x^1/3 y - y^4/3),  C = E pi (3/4pi)^(4/3);  the negative for x < y;
//     Long:  c_b (x^2 + y^2) on the square x, y < x_t;  c_a (x + y) elsewhere.
// For Gamma / Exponential densities the integral of a power term over such a region is a product of moments times a
// probability with a closed form,
//     int int_{y < x} x^a y^b f_j(x) f_k(y) dx dy = M^j_a M^k_b I_z(k_k + b, k_j + a),   z = theta_j / (theta_j + theta_k)
// (regularised incomplete beta: P(Y' < X') for two Gamma variates), and the Long square gives partial moments
// M_q P(k + q, x_t / theta).  So the Q and R matrices (:503-578) need no quadrature at all.  The self collisions that
// weighting_fn (:624-642) splits between a mode and the next are integrals over s = x' + y of a function of s alone:
// S = X + Y ~ Gamma(2k, theta) is independent of tau = Y / S ~ Beta(k, k), hence
//     T_m = S_2k^(m) = 1/2 n^2 E[ S^m (1 - w(S)) G(S) ],     G(s) = E_tau[ K(s (1 - tau), s tau) ],
// with G in closed form too (homogeneous kernels: G(s) = G(1) s^gamma, absorbed in the Gamma weight; Long: incomplete betas
// of argument x_t / s between x_t and 2 x_t).  What is left is ONE 1-D integral per mode of a sigmoid-like function of s
// (ratios of the modes' densities) against a Gamma density.  Its transitions can be arbitrarily sharp against the mode's own
// scale (a narrow or much smaller neighbour), so the rule is ADAPTIVE, as the reference's quadgk is: Gauss-Kronrod (7, 15)
// panels bisected until the estimate meets a relative tolerance, the initial panels graded around the other modes' cores
// (conv_adaptive below).  A fixed composite rule (rounds up to 3: 48 panels x 8 points) met 1e-8 on the golden cases but
// missed it on 8 % of random mixtures with shapes in [1, 10] -- 25 % with shapes down to 0.01 -- by up to 1e-3.
//
// Measured against nested adaptive quadrature of the reference integrals (tests/golden/numerical_adaptive.json):
// <= 5.5e-10 of scale (the accuracy of the golden values), where the 10-point rule has 1e-3 ... 1e-2; against itself at
// tol = 1e-13 on random mixtures: <= 1e-9 (tests/test_numerical_oracle.py prints both tables).
// (The CPU test suite holds a same-rule restatement of these formulas and the adaptive values.)
#pragma once
#include "kernels.hpp"
#include "quad.hpp"

namespace cloudy {

// Workgroup size of the converged-mode kernels: 256 ahead of time; a kernel compiled for its plan gets the size the plan was
// created with (jit.hpp: jit_conv_block_size, HostPlan::jit_conv_bs) as a macro in front of this header -- the LDS rows of this
// file are one slot per lane wide, and the rankings order that many parcels.
#ifndef CLOUDY_CONV_BLOCK
#define CLOUDY_CONV_BLOCK 256
#endif
constexpr int kConvBlock = CLOUDY_CONV_BLOCK;
constexpr int kConvUnrolledModes = 4;   // = CLOUDY_AOT_MAX_MODES (include/cloudy_hip.h): plans of more modes take conv_coal_ints_rolled

// continued fraction of the incomplete beta function (DLMF 8.17.22), converging for x < (a+1)/(a+b+2):
//   h = 1 / (1 + e_1 / (1 + e_2 / (1 + ...))),   e_1 = -(a+b) x / (a+1),
//   e_2m = m (b-m) x / ((a+2m-1)(a+2m)),   e_2m+1 = -(a+m)(a+b+m) x / ((a+2m)(a+2m+1))
// by the forward (Wallis) recurrence of its numerators and denominators, A_n = A_n-1 + e_n A_n-2 (B likewise): one
// reciprocal per coefficient instead of the three of a modified-Lentz step (rounds 2-3; ~55 -> ~36 instructions per
// double step).  In the range it is called for |e_n| < 1, so A and B grow at most like Fibonacci numbers (1e125 over the
// 600 coefficients of the iteration cap) and no rescaling is needed; the relative change of A_n / B_n is tested through
// the cross product, without a division.  Against 30-digit mpmath over a, b in [1e-3, 50]: 7e-15 (Lentz: 6e-15), the
// same number of steps (7 on average).
__device__ __forceinline__ double inc_beta_cf(double a, double b, double x) {
    double Ap = 1.0, Bp = 1.0, Ac = 1.0, Bc = 1.0 - (a + b) * x * recip_fast(a + 1.0);
#pragma unroll 1
    for (int m = 1; m <= 300; ++m) {
        const double md = double(m), am2 = a + 2.0 * md;
        double e = md * (b - md) * x * recip_fast((am2 - 1.0) * am2);
        double An = fma(e, Ap, Ac), Bn = fma(e, Bp, Bc);
        Ap = Ac;
        Bp = Bc;
        Ac = An;
        Bc = Bn;
        e = -(a + md) * (a + b + md) * x * recip_fast(am2 * (am2 + 1.0));
        An = fma(e, Ap, Ac);
        Bn = fma(e, Bp, Bc);
        Ap = Ac;
        Bp = Bc;
        Ac = An;
        Bc = Bn;
        // (A_n / B_n - A_n-1 / B_n-1) / (A_n / B_n) = (A_n B_n-1 - A_n-1 B_n) / (A_n B_n-1)
        const double cross = Ac * Bp;
        if (!(fabs(fma(-Ap, Bc, cross)) > 1e-15 * fabs(cross))) break;
    }
    return Ac * recip_fast(Bc);
}

// I_x(a, b) given D = x^a (1-x)^b / B(a, b); omx = 1 - x (passed in: the callers know it without cancellation)
__device__ __forceinline__ double inc_beta_from_D(double a, double b, double x, double omx, double D) {
    // (x < (a + 1) / (a + b + 2) without the division; at the boundary either branch converges)
    if (x * (a + b + 2.0) < a + 1.0) return D * inc_beta_cf(a, b, x) * recip_fast(a);
    return fma(-D * recip_fast(b), inc_beta_cf(b, a, omx), 1.0);
}

// per-mode quantities of the closed forms.  Lognormal modes keep (mu, sigma) in (th, k), as everywhere (kernels.hpp).
struct ConvMode {
    bool lognormal;
    double n, th, k, lnth, lgk;
    double lgk3;   // hydrodynamic, Gamma family: ln Gamma(k + 1/3)
    double Mi[5];  // M_0 .. M_4
    double Mt[4];  // M_{1/3}, M_{4/3}, M_{7/3}, M_{10/3}   (hydrodynamic)
    double pm[5];  // partial moments below the Long threshold, M_q P(k + q, x_t / theta)
};

__device__ __forceinline__ double conv_norm_cdf(double z) { return 0.5 * erfc(-z * 0.70710678118654752440); }

template <int KIND>
__device__ __forceinline__ void conv_mode(const QArgs &Q, bool lognormal, double n, double th, double k, ConvMode &m) {
    m.lognormal = lognormal;
    m.n = n;
    m.th = th;
    m.k = k;
    if (lognormal) {  // wave-uniform.  M_q = n exp(q mu + q^2 sigma^2 / 2); partial moments M_q Phi((ln x_t - mu - q sigma^2) / sigma)
        const double s2 = k * k;
        m.lnth = th;
        m.lgk = m.lgk3 = 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) m.Mi[q] = n * exp_fin(fma(double(q), th, 0.5 * double(q * q) * s2));
        if (KIND == KF_HYDRODYNAMIC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double o = double(q) + 1.0 / 3.0;
                m.Mt[q] = n * exp_fin(fma(o, th, 0.5 * (o * o) * s2));
            }
        }
        if (KIND == KF_LONG) {
            const double w0 = (log_pos(Q.kf[0]) - th) / k;
#pragma unroll
            for (int q = 0; q < 5; ++q) m.pm[q] = m.Mi[q] * conv_norm_cdf(w0 - double(q) * k);
        }
        return;
    }
    m.lnth = log_pos(th);
    m.lgk = lgamma_pos(k);
    m.lgk3 = 0.0;
    m.Mi[0] = n;
#pragma unroll
    for (int q = 1; q < 5; ++q) m.Mi[q] = m.Mi[q - 1] * (th * (k + double(q - 1)));
    if (KIND == KF_HYDRODYNAMIC) {
        const double lr3 = log_gamma_ratio(k, 1.0 / 3.0);
        m.lgk3 = m.lgk + lr3;   // (the pair terms need ln Gamma(k + 1/3): kept instead of a second lgamma per pair and grid)
        m.Mt[0] = n * exp_fin(fma(1.0 / 3.0, m.lnth, lr3));
#pragma unroll
        for (int q = 1; q < 4; ++q) m.Mt[q] = m.Mt[q - 1] * (th * (k + (double(q - 1) + 1.0 / 3.0)));
    }
    if (KIND == KF_LONG) {
        // P(k + 4, z) by series / continued fraction, the lower orders by P(a - 1, z) = P(a, z) + z^(a-1) e^-z / Gamma(a)
        const double z = Q.kf[0] / th, a_top = k + 4.0;
        double E = exp_fin(fma(a_top, log_pos(z), -z) - lgamma_pos(a_top + 1.0));  // z^a e^-z / Gamma(a + 1)
        double P = inc_gamma_p_from_E(a_top, z, E, nullptr);
        const double invz = 1.0 / z;
        double a = a_top;
#pragma unroll
        for (int q = 4; q >= 0; --q) {
            m.pm[q] = m.Mi[q] * fmin(P, 1.0);
            E *= a * invz;
            a -= 1.0;
            P += E;
        }
    }
}

// ln of the normed density at s (ls = ln s), ParticleDistributions.jl:363-388:
//   Gamma family: a ls - b s - c,  a = k - 1, b = 1 / theta, c = ln Gamma(k) + k ln theta
//   Lognormal:    -(ls - a)^2 b - ls - c,  a = mu, b = 1 / (2 sigma^2), c = ln(sigma sqrt(2 pi))
struct ConvLogDensity {
    bool lognormal;
    double a, b, c;
    __device__ __forceinline__ double operator()(double s, double ls) const {
        if (lognormal) {
            const double d = ls - a;
            return -(d * d) * b - ls - c;
        }
        return fma(a, ls, -(s * b)) - c;
    }
};
__device__ __forceinline__ ConvLogDensity conv_log_density(const ConvMode &m) {
    ConvLogDensity l;
    l.lognormal = m.lognormal;
    if (m.lognormal) {
        l.a = m.th;
        l.b = 0.5 / (m.k * m.k);
        l.c = log_pos(m.k * 2.5066282746310002);
    } else {
        l.a = m.k - 1.0;
        l.b = 1.0 / m.th;
        l.c = fma(m.k, m.lnth, m.lgk);
    }
    return l;
}
// 1 - weighting_fn(s, j + 1) from density ratios against mode j's own density (Coalescence.jl:624-642)
template <int N>
__device__ __forceinline__ double conv_one_minus_w(const ConvLogDensity (&lg)[N], int j, double s, double ls) {
    double own = 0.0;
#pragma unroll
    for (int m = 0; m < N; ++m)
        if (m == j) own = lg[m](s, ls);
    double up = 0.0, den = 1.0;
#pragma unroll
    for (int m = 0; m < N; ++m) {
        if (m == j) continue;
        const double rho = exp_fin(fmin(lg[m](s, ls) - own, 700.0));
        den += rho;
        if (m > j) up += rho;
    }
    return up * recip_fast(den);
}

// ---- the adaptive 1-D rule (same specification as the CPU restatement the tests hold) ----
// Variable t = ln(s / theta) in [t_lo, t_hi]: the Gamma weight u^(A-1) e^-u du is exp(A t - e^t) dt, entire in t (the
// bisection deals with the double-exponential upper flank) -- two exponentials per node, no logarithm, no branch.
// Initial panels: the union of kConvNInit equal pieces of [t_lo, t_hi] (width h0) with a list of marks -- the kinks of the
// integrand (Long: s = x_t, 2 x_t) and, for every other mode, marks graded around its core, where weighting_fn has its
// transitions: ln s = c and c +- w 2^i, i = 0 .. I (c = ln of the mode's mean size; w = sigma for a Lognormal mode,
// 1 / sqrt(max(k, 1)) for a Gamma mode; I = the first doubling at which w 2^I reaches h0, at most kConvIMax)
// -- walked from t_lo upwards (next edge = the smallest mark or equal-piece edge more than gap = 1e-7 (t_hi - t_lo) beyond
// the current one).  So a spike of another mode, however narrow, meets panels of its own size, and the transitions in its
// flanks lie well inside a panel, not at an edge where the nodes would miss them.  Every initial panel is integrated by
// bisection with the Gauss-Kronrod (7, 15) pair: a panel is accepted when, for each output, |K15 - G7| <= tol x max(|the
// integral accumulated so far, this panel included|, kConvFloor x the output's scale) -- the relative criterion of the
// reference's quadgk(rtol), with the accumulated value standing in for the final one -- at depth kConvLMax, or once the rule
// has spent its budget of panel evaluations; the accepted value is K15.  With tol = 1e-9 the accepted K15 values are good to
// ~1e-14 (G7 is the estimate's accuracy, K15 has 1.6 x its order), so a decision that flips on a rounding difference
// between two implementations moves the result by that much, not by tol.
// Lanes walk their own panel trees in ONE flat loop (a new initial panel is just another state of it): a wave runs for as
// long as its lane with the most panel evaluations.
constexpr int kConvNInit = 10, kConvLMax = 12, kConvIMax = 12, kConvBudget = 8192, kConvBudgetLn = 1024;
constexpr double kConvTol = 1e-7, kConvFloor = 1e-10, kConvTermTol = 1e-10;
__device__ static const double kGKX[15] = {-0.991455371120812639206854697526329, -0.949107912342758524526189684047851,
                                           -0.864864423359769072789712788640926, -0.741531185599394439863864773280788,
                                           -0.586087235467691130294144838258730, -0.405845151377397166906606412076961,
                                           -0.207784955007898467600689403773245, 0.0,
                                           0.207784955007898467600689403773245,  0.405845151377397166906606412076961,
                                           0.586087235467691130294144838258730,  0.741531185599394439863864773280788,
                                           0.864864423359769072789712788640926,  0.949107912342758524526189684047851,
                                           0.991455371120812639206854697526329};
__device__ static const double kGKWK[15] = {0.022935322010529224963732008058970, 0.063092092629978553290700663189204,
                                            0.104790010322250183839876322541518, 0.140653259715525918745189590510238,
                                            0.169004726639267902826583426598550, 0.190350578064785409913256402421014,
                                            0.204432940075298892414161999234649, 0.209482141084727828012999174891714,
                                            0.204432940075298892414161999234649, 0.190350578064785409913256402421014,
                                            0.169004726639267902826583426598550, 0.140653259715525918745189590510238,
                                            0.104790010322250183839876322541518, 0.063092092629978553290700663189204,
                                            0.022935322010529224963732008058970};
__device__ static const double kGKWG[15] = {0.0, 0.129484966168869693270611432679082, 0.0, 0.279705391489276667901467771423780,
                                            0.0, 0.381830050505118944950369775488975, 0.0, 0.417959183673469387755102040816327,
                                            0.0, 0.381830050505118944950369775488975, 0.0, 0.279705391489276667901467771423780,
                                            0.0, 0.129484966168869693270611432679082, 0.0};

// the Gamma(A) weight of the rule's variable at t = ln u: u and density x du/dt
struct ConvNode {
    double u, lu, wt;
};
__device__ __forceinline__ ConvNode conv_node(double t, double A, double lgA) {
    ConvNode nd;
    nd.u = exp_fin(t);  // du = u dt
    nd.lu = t;
    nd.wt = exp_fin(fma(A, t, -nd.u) - lgA);
    return nd;
}
__device__ __forceinline__ void conv_range(double A, double top, double lgA, double &tlo, double &thi) {
    // (ln 1e-13; the lower clamp serves closures clamped to k = eps, whose weight is ~ 1/u over hundreds of decades)
    tlo = fmax(-690.0, fmin(-1.0, (-29.933606208922594 + (lgA + log_pos(A))) / A));
    thi = log_pos((A + top) + sqrt(60.0 * (A + top)) + 30.0);
}
// ln of the mean size of a mode and the width of its core in ln s
__device__ __forceinline__ double conv_ln_mean(const ConvMode &m) {
    return m.lognormal ? fma(0.5 * m.k, m.k, m.th) : log_pos(m.k * m.th);
}
__device__ __forceinline__ double conv_core_width(const ConvMode &m) { return m.lognormal ? m.k : 1.0 / sqrt(fmax(m.k, 1.0)); }

// The marks of a rule: NM cores, each an ascending list walked by a pointer (the walk only ever asks for the smallest mark
// beyond a limit that grows), and up to three single marks.
template <int NM>
struct ConvMarks {
    double c[NM], w[NM], v[NM];
    int I[NM], q[NM];
    double extra[3];
    double shift;  // marks are ln s - shift: ln theta for a Gamma-weight rule, 0 for a rule over ln s itself

    __device__ __forceinline__ double value(int m) const {
        const int qq = q[m];
        if (qq > I[m] + 1) return INFINITY;
        const int aq = qq < 0 ? -qq : qq;
        const double off = qq == 0 ? 0.0 : (qq < 0 ? -1.0 : 1.0) * ldexp(w[m], aq - 1);
        return (c[m] + off) - shift;
    }
    __device__ __forceinline__ void core(int m, double c_, double w_, double h0) {
        c[m] = c_;
        w[m] = w_;
        const double ratio = h0 / w_;  // (every rule's variable is a logarithm of s: the equal pieces are h0 wide in ln s)
        int i = 0;
        if (ratio > 1.0) i = ratio < 4096.0 ? (int)ceil(log2(ratio)) : kConvIMax;
        I[m] = i > kConvIMax ? kConvIMax : i;
        q[m] = -(I[m] + 1);
        v[m] = value(m);
    }
    // the same marks walked DOWNWARDS (conv_T_merged): start at q = I + 1; below -(I + 1): none (-inf)
    __device__ __forceinline__ double value_d(int m) const {
        const int qq = q[m];
        if (qq < -(I[m] + 1)) return -INFINITY;
        const int aq = qq < 0 ? -qq : qq;
        const double off = qq == 0 ? 0.0 : (qq < 0 ? -1.0 : 1.0) * ldexp(w[m], aq - 1);
        return (c[m] + off) - shift;
    }
    // the largest mark < lim; -inf: none
    __device__ __forceinline__ double prev(double lim, bool need) {
        double r = -INFINITY;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            while (need && v[m] >= lim) {
                --q[m];
                v[m] = value_d(m);
            }
            r = fmax(r, v[m]);
        }
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (extra[e] < lim) r = fmax(r, extra[e]);
        return r;
    }
    // the smallest mark > lim; +inf: none.  Lanes with need = false run through it without changing anything (the caller
    // keeps every lane on one path: conv_adaptive)
    __device__ __forceinline__ double next(double lim, bool need) {
        double r = INFINITY;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            while (need && v[m] <= lim) {
                ++q[m];
                v[m] = value(m);
            }
            r = fmin(r, v[m]);
        }
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (extra[e] > lim) r = fmin(r, extra[e]);
        return r;
    }
};

// The walk.  node(t, vals[NOUT]) = the integrand per unit t.  TWO_PASS (the 32 outputs of conv_H_grid): the estimate
// comes from est(t, ve[NEST]), the NEST outputs est_idx[] of the full node, and an accepted panel is evaluated a
// second time for all outputs.
template <int NOUT, int NEST, bool TWO_PASS, int NM, class EstFn, class NodeFn>
__device__ __forceinline__ int conv_adaptive(double tlo, double thi, ConvMarks<NM> &mk, const int (&est_idx)[NEST],
                                             const double (&scaleS)[NOUT], int budget, EstFn &&est, NodeFn &&node,
                                             double (&out)[NOUT]) {
    const int budget0 = budget;   // (returns the number of panel evaluations spent: the cost hint of a Lognormal mode's rule)
    const double h0 = (thi - tlo) * (1.0 / double(kConvNInit)), gap = 1e-7 * (thi - tlo);
    double cur = tlo, a0 = tlo, h = 0.0;
    int io = 1, L = 0;
    unsigned i = 0;
    bool busy = true;
    // The next initial panel [cur, nxt) for the lanes with need = true (busy = false: their range is exhausted); the others
    // run through the same instructions and keep their state -- no branch around it, see the end of the loop body.
    const auto next_panel = [&](bool need) {
        const bool go = need && cur < thi;
        const double lim = cur + gap;
        int io2 = io;
        double own = fma(h0, double(io2), tlo);
        while (need && own <= lim) {
            ++io2;
            own = fma(h0, double(io2), tlo);
        }
        double nxt = fmin(thi, own);
        nxt = fmin(nxt, mk.next(lim, need));
        nxt = nxt > thi - gap ? thi : nxt;
        io = io2;
        a0 = go ? cur : a0;
        h = go ? nxt - cur : h;
        cur = go ? nxt : cur;
        L = go ? 0 : L;
        i = go ? 0u : i;
        busy = need ? go : busy;
    };
    next_panel(true);
#pragma unroll 1
    while (busy) {
        const double w = ldexp(h, -L), hw = 0.5 * w, c = fma(w, double(i), a0) + hw;
        double K[NEST], G[NEST];
#pragma unroll
        for (int e = 0; e < NEST; ++e) K[e] = G[e] = 0.0;
        --budget;
        // (unrolled: the Kronrod nodes and weights become literals.  As a loop every node waited for three scalar loads of
        // its constants: 7 % of the kernel's time at two waves per SIMD)
#pragma unroll
        for (int g = 0; g < 15; ++g) {
            double ve[NEST];
            est(fma(hw, kGKX[g], c), ve);
            const double wk = kGKWK[g], wg = kGKWG[g];
#pragma unroll
            for (int e = 0; e < NEST; ++e) {
                K[e] = fma(wk, ve[e], K[e]);
                if (g & 1) G[e] = fma(wg, ve[e], G[e]);  // (the Gauss nodes are the odd ones; a zero weight is not a no-op for the compiler)
            }
        }
        bool ok = true;
#pragma unroll
        for (int e = 0; e < NEST; ++e) {
            double acc_o = 0.0, sc_o = 0.0;
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
                if (o == est_idx[e]) {
                    acc_o = out[o];
                    sc_o = scaleS[o];
                }
            // (a NaN estimate accepts: the NaN reaches the output)
            if (fabs(K[e] - G[e]) * hw > kConvTol * fmax(fabs(fma(K[e], hw, acc_o)), kConvFloor * sc_o)) ok = false;
        }
        const bool accept = ok || L == kConvLMax || budget <= 0;
        if (accept) {
            if (TWO_PASS) {
                double Kf[NOUT];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) Kf[o] = 0.0;
#pragma unroll 1
                for (int g = 0; g < 15; ++g) {
                    double vals[NOUT];
                    node(fma(hw, kGKX[g], c), vals);
                    const double wk = kGKWK[g];
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) Kf[o] = fma(wk, vals[o], Kf[o]);
                }
#pragma unroll
                for (int o = 0; o < NOUT; ++o) out[o] = fma(Kf[o], hw, out[o]);
            } else {
#pragma unroll
                for (int e = 0; e < NEST; ++e) {
#pragma unroll
                    for (int o = 0; o < NOUT; ++o)
                        if (o == est_idx[e]) out[o] = fma(K[e], hw, out[o]);
                }
            }
        }
        // The step of the tree walk as straight-line code: accepted -> the next sibling, climbing while it is a right
        // child's successor (trailing zeros of the index); rejected -> the left child; then the next initial panel for the
        // lanes that have finished theirs, as predicated code every lane runs through.  (Written with branches -- accept /
        // reject / new panel each jumping back to the top -- the loop had three back edges, which the compiler turns into
        // three NESTED loops: lanes that had accepted then waited for the lanes still bisecting, and a wave of 64 different
        // parcels ran 1.75 x the trips of its busiest lane.)
        const unsigned ni = accept ? i + 1u : i << 1;
        const int up = accept ? min((int)__builtin_ctz(ni), L) : 0;
        i = ni >> up;
        L = accept ? L - up : L + 1;
        next_panel(accept && L == 0);
    }
    return budget0 - budget;
}

// P(L'_c < G'_a) for a Gamma-family mode G and a Lognormal mode L on the grid a = f + i (f = 0, 1/3; i = 0..3),
// c = 0, 1/3, 1, 4/3, 2, 7/3, 3, 10/3:  H[f][i][ci] = E_{Gamma(k_g + a, theta_g)}[Phi((ln x - mu - c sigma^2) / sigma)]
// by the adaptive rule; the orders i share the nodes of the base shape: E_{Gamma(A0 + i)}[g] = E_{Gamma(A0)}[u^i g] / (A0)_i.
// The error estimate looks at the four corner outputs (i = 0, 3; c = 0, 10/3).
__device__ __forceinline__ void conv_H_grid(const ConvMode &G, const ConvMode &L, double (&H)[2][4][8]) {
    const double inv_sg = 1.0 / L.k;
#pragma unroll 1
    for (int f = 0; f < 2; ++f) {
        const double A0 = G.k + (f ? 1.0 / 3.0 : 0.0), lgA = f ? lgamma_pos(A0) : G.lgk;
        double tlo, thi;
        conv_range(A0, 4.0, lgA, tlo, thi);
        ConvMarks<1> mk;
        mk.shift = G.lnth;
        mk.extra[0] = mk.extra[1] = mk.extra[2] = INFINITY;
        mk.core(0, fma((5.0 / 3.0) * L.k, L.k, L.th), L.k, (thi - tlo) * (1.0 / double(kConvNInit)));  // where the Phi's turn
        double acc[32], scaleS[32];
        {
            double poch = 1.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    acc[8 * i + c] = 0.0;
                    scaleS[8 * i + c] = poch;
                }
                poch *= A0 + double(i);
            }
        }
        const int est_idx[4] = {0, 7, 24, 31};
        const auto est = [&](double t, double (&ve)[4]) {
            const ConvNode nd = conv_node(t, A0, lgA);
            const double w0 = (nd.lu + G.lnth - L.th) * inv_sg;
            const double p0 = conv_norm_cdf(w0), p7 = conv_norm_cdf(w0 - (3.0 + 1.0 / 3.0) * L.k);
            const double w3 = ((nd.wt * nd.u) * nd.u) * nd.u;
            ve[0] = nd.wt * p0;
            ve[1] = nd.wt * p7;
            ve[2] = w3 * p0;
            ve[3] = w3 * p7;
        };
        const auto node = [&](double t, double (&vals)[32]) {
            const ConvNode nd = conv_node(t, A0, lgA);
            const double w0 = (nd.lu + G.lnth - L.th) * inv_sg;
            double ph[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) ph[c] = conv_norm_cdf(w0 - (double(c >> 1) + ((c & 1) ? 1.0 / 3.0 : 0.0)) * L.k);
            double wu = nd.wt;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int c = 0; c < 8; ++c) vals[8 * i + c] = wu * ph[c];
                wu *= nd.u;
            }
        };
        conv_adaptive<32, 4, true>(tlo, thi, mk, est_idx, scaleS, kConvBudget, est, node, acc);
        double poch = 1.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double r = 1.0 / poch;
#pragma unroll
            for (int c = 0; c < 8; ++c) H[f][i][c] = acc[8 * i + c] * r;
            poch *= A0 + double(i);
        }
    }
}

// The four pair integrals  int int x^p y^q K(x, y) f_j(x) f_k(y)  for (p, q) = (0,0), (1,0), (2,0), (1,1)  ->  s0, sa, saa, sab
// (tab: the Gauss-Legendre table; used only by a Gamma-Lognormal pair of the hydrodynamic kernel)
template <int KIND>
__device__ __forceinline__ void conv_pair(const QArgs &Q, const double *__restrict__ tab, const ConvMode &J, const ConvMode &K,
                                          bool self, double (&out)[4]) {
    constexpr int PP[4] = {0, 1, 2, 1}, QQ[4] = {0, 0, 0, 1};
    if (KIND == KF_CONSTANT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = Q.kf[0] * (J.Mi[PP[i]] * K.Mi[QQ[i]]);
    } else if (KIND == KF_LINEAR) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = Q.kf[0] * fma(J.Mi[PP[i] + 1], K.Mi[QQ[i]], J.Mi[PP[i]] * K.Mi[QQ[i] + 1]);
    } else if (KIND == KF_LONG) {
        const double cb = Q.kf[1], ca = Q.kf[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = PP[i], q = QQ[i];
            const double full = fma(J.Mi[p + 1], K.Mi[q], J.Mi[p] * K.Mi[q + 1]);
            const double b2 = fma(J.pm[p + 2], K.pm[q], J.pm[p] * K.pm[q + 2]);
            const double b1 = fma(J.pm[p + 1], K.pm[q], J.pm[p] * K.pm[q + 1]);
            out[i] = fma(ca, full - b1, cb * b2);
        }
    } else {
        // hydrodynamic: sum_t c_t M^j_{p + al_t} M^k_{q + be_t} (2 P_t - 1),  P_t = P(Y' < X') for the size-biased laws
        // X' (mode j, order p + al_t) and Y' (mode k, order q + be_t).  Pr[t][i] for the four terms
        //   t = 0: (p + 4/3, q)   1: (p + 1, q + 1/3)   2: (p + 1/3, q + 1)   3: (p, q + 4/3)
        double Pr[4][4];
        if (!J.lognormal && !K.lognormal) {
            // Gamma pair: P = I_z(k_k + b, k_j + a), z = theta_j / (theta_j + theta_k).  The 16 incomplete betas lie on two
            // grids of unit spacing, G1[i][i'] = I_z(k_k + i, k_j + 1/3 + i') and G2[i][i'] = I_z(k_k + 1/3 + i, k_j + i'):
            // one continued fraction each, the rest by the recurrences
            //   I(a, b+1) = I(a, b) + D(a, b) / b,   I(a+1, b) = I(a, b) - D(a, b) / a,   D = z^a (1-z)^b / B(a, b),
            //   D(a, b+1) = D(a, b) (1-z) (a+b) / b,   D(a+1, b) = D(a, b) z (a+b) / a
            // (absolute accuracy is what 2 I - 1 needs; the recurrences keep it).
            const double z = self ? 0.5 : J.th / (J.th + K.th), omz = self ? 0.5 : K.th / (J.th + K.th);
            const double lz = log_pos(z), lomz = log_pos(omz);
            double G[2][3][4];
            const double lgab = lgamma_pos(K.k + J.k + 1.0 / 3.0);  // ln Gamma(a0 + b0), the same for both grids
            // A self pair (z = 1/2, both modes the same): the second grid is the mirror image of the first,
            //   I_1/2(k + 1/3 + i, k + i') = 1 - I_1/2(k + i', k + 1/3 + i),
            // so ONE continued fraction serves it (the first grid then runs one row further in a).
            double G0x[4];  // G1 row a = k + 3 of the first grid (self pairs only)
#pragma unroll
            for (int g = 0; g < (self ? 1 : 2); ++g) {
                const double a0 = K.k + (g ? 1.0 / 3.0 : 0.0), b0 = J.k + (g ? 0.0 : 1.0 / 3.0);
                const double lgb = g ? J.lgk : J.lgk3, lga = g ? K.lgk3 : K.lgk;
                double D0 = exp_fin(fma(a0, lz, b0 * lomz) + (lgab - lga - lgb));
                double I0 = inc_beta_from_D(a0, b0, z, omz, D0);
                // (quotients by a, a + 1, a + 2 and b ... b + 3 through seven reciprocals per grid -- ~1 ulp each, as in
                // inc_beta_cf -- instead of 32 IEEE divisions: round 4, ~3 k of the 11.7 k instructions of the closed forms)
                double ra[3];
#pragma unroll
                for (int ia = 0; ia < 3; ++ia) ra[ia] = recip_fast(a0 + double(ia));
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {  // walk along b at i = 0, then down the column in a
                    const double b = b0 + double(ib), rb = recip_fast(b);
                    double I = I0, D = D0, a = a0;
#pragma unroll
                    for (int ia = 0; ia < 3; ++ia) {
                        G[g][ia][ib] = I;
                        I = fma(-D, ra[ia], I);
                        D *= (z * (a + b)) * ra[ia];
                        a += 1.0;
                    }
                    if (self) G0x[ib] = I;   // I_z(a0 + 3, b0 + ib)
                    I0 = fma(D0, rb, I0);
                    D0 *= (omz * (a0 + b)) * rb;
                }
            }
            if (self) {
#pragma unroll
                for (int ia = 0; ia < 3; ++ia)
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) G[1][ia][ib] = 1.0 - (ib < 3 ? G[0][ib][ia] : G0x[ia]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = PP[i], q = QQ[i];
                Pr[0][i] = G[0][q][p + 1];
                Pr[1][i] = G[1][q][p + 1];
                Pr[2][i] = G[0][q + 1][p];
                Pr[3][i] = G[1][q + 1][p];
            }
        } else if (J.lognormal && K.lognormal) {
            // ln X' - ln Y' ~ N(mu_j + a s_j^2 - mu_k - b s_k^2, s_j^2 + s_k^2)
            const double sj2 = J.k * J.k, sk2 = K.k * K.k, inv = 1.0 / sqrt(sj2 + sk2), d0 = J.th - K.th;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double p = double(PP[i]), q = double(QQ[i]);
                Pr[0][i] = conv_norm_cdf((d0 + (p + 4.0 / 3.0) * sj2 - q * sk2) * inv);
                Pr[1][i] = conv_norm_cdf((d0 + (p + 1.0) * sj2 - (q + 1.0 / 3.0) * sk2) * inv);
                Pr[2][i] = conv_norm_cdf((d0 + (p + 1.0 / 3.0) * sj2 - (q + 1.0) * sk2) * inv);
                Pr[3][i] = conv_norm_cdf((d0 + p * sj2 - (q + 4.0 / 3.0) * sk2) * inv);
            }
        } else {
            // one Gamma-family and one Lognormal mode: P(L'_c < G'_a) by the 1-D rule (conv_H_grid); with the Lognormal mode
            // as j, P(Y' < X') = 1 - P(X' < Y').  Gamma order a = i + (f ? 1/3 : 0), Lognormal order c -> index 2 floor(c) + (frac != 0)
            double H[2][4][8];
            const bool jg = !J.lognormal;
            conv_H_grid(jg ? J : K, jg ? K : J, H);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = PP[i], q = QQ[i];
                // term orders (x-order thirds, y-order thirds): t0 (3p+4, 3q), t1 (3p+3, 3q+1), t2 (3p+1, 3q+3), t3 (3p, 3q+4)
                const int ox[4] = {3 * p + 4, 3 * p + 3, 3 * p + 1, 3 * p}, oy[4] = {3 * q, 3 * q + 1, 3 * q + 3, 3 * q + 4};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int og = jg ? ox[t] : oy[t], ol = jg ? oy[t] : ox[t];   // Gamma / Lognormal orders in thirds
                    const double h = H[og % 3 ? 1 : 0][og / 3][2 * (ol / 3) + (ol % 3 ? 1 : 0)];
                    Pr[t][i] = jg ? h : 1.0 - h;
                }
            }
        }
        const double C = Q.kf[0] * 0.46526286817455001;  // pi (3 / (4 pi))^(4/3)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = PP[i], q = QQ[i];
            double v = J.Mt[p + 1] * K.Mi[q] * fma(2.0, Pr[0][i], -1.0);               // x^(4/3)
            v = fma(2.0 * (J.Mi[p + 1] * K.Mt[q]), fma(2.0, Pr[1][i], -1.0), v);       // 2 x y^(1/3)
            v = fma(-2.0 * (J.Mt[p] * K.Mi[q + 1]), fma(2.0, Pr[2][i], -1.0), v);      // -2 x^(1/3) y
            v = fma(-(J.Mi[p] * K.Mt[q + 1]), fma(2.0, Pr[3][i], -1.0), v);            // -y^(4/3)
            out[i] = C * v;
        }
    }
}

// E_tau[K(s (1 - tau), s tau)], tau ~ Beta(k, k), Long kernel, for x_t < s < 2 x_t
__device__ __forceinline__ double conv_long_G_mid(const QArgs &Q, double k, double lgB, double rB, double s) {
    const double xt = Q.kf[0], cb = Q.kf[1], ca = Q.kf[2];
    const double x = xt * recip_fast(s), omx = 1.0 - x;  // x in (1/2, 1)
    // I_x(k, k) = 1 - I_{1-x}(k, k); I_x(k+1, k+1) by two steps of the recurrences
    const double D = exp_fin(fma(k, log_pos(x * omx), lgB));  // x^k (1-x)^k / B(k, k)  (x (1 - x) in [0, 1/4]: one logarithm)
    const double Dk = D * recip_fast(k);                 // (one reciprocal for the three quotients by k)
    const double Ikk = fma(-Dk, inc_beta_cf(k, k, omx), 1.0);
    const double Ik1k = Ikk - Dk;                        // I_x(k+1, k)
    const double Ik1k1 = fma(Dk * x, 2.0, Ik1k);         // I_x(k+1, k+1) = I_x(k+1, k) + D(k+1, k) / k,  D(k+1, k) = D z (2k) / k
    const double P0 = fma(2.0, Ikk, -1.0), P1 = rB * fma(2.0, Ik1k1, -1.0);
    const double s2 = s * s;
    return fma(ca, s, fma(fma(cb, s2, -(ca * s)), P0, -2.0 * cb * s2 * P1));
}

// ---- Long kernel, x_t < s < 2 x_t: I_y(k, k), y = 1 - x_t / s in (0, 1/2), from a per-rule table (round 5) ----
// Rounds 3-4 ran one incomplete-beta continued fraction per node there (inc_beta_cf: ~7 double steps on average, each lane
// as many as ITS (k, y) needs, the wave as many as its slowest lane): a Long-kernel rule cost 3.7 x a hydrodynamic one per
// evaluation (7.4e7 against 4.6e8 parcel-RHS/s).  For fixed k the function is smooth in y once its
// end-point behaviour is factored out:
//     I_y(k, k) = y^k / (k B(k, k)) g(y),   g(y) = (1 - y)^k 2F1(2k, 1; k+1; y) = 2F1(1-k, k; k+1; y)   (Euler)
// -- g is a polynomial for integer k, bounded by 1, with its only singularity at y = 1 -- so a rule tabulates g ONCE, before
// the walk, at kLongNT Chebyshev points of [0, 1/2] (one continued fraction each: every lane the same dozen), turns the values
// into Chebyshev coefficients (the cosine transform below: perfectly conditioned) and a node between x_t and 2 x_t costs one
// logarithm, two exponentials and a kLongNT-term Clenshaw sum (round 6: the same polynomial in powers of 4 y - 1 by Horner,
// kLongMono below), the same for every lane.  Truncation + rounding against
// 30-digit mpmath over k in [1e-8, 10]: <= 7e-13 absolute in I_y (kLongNT = 12); shapes beyond kLongTabKmax (a plan whose
// k_range allows them) keep the continued fraction per node, as do N = 4 plans (the tables of three rules would not leave
// two workgroups per CU their LDS).
constexpr int kLongNT = 12;
constexpr double kLongTabKmax = 10.0;
// y_i = (1 + cos(pi (i + 1/2) / 12)) / 4 and ln(1 - y_i)
__device__ static const double kLongY[kLongNT] = {0.4978612153434526028, 0.480969883127821689, 0.4483383350728087911, 0.4021903572521801599, 0.3456708580912724429, 0.2826315480550128979, 0.2173684519449871021, 0.1543291419087275571, 0.09780964274781984015, 0.05166166492719120886, 0.01903011687217831097, 0.002138784656547397214};
__device__ static const double kLongL1mY[kLongNT] = {-0.6888787340401301498, -0.65579336884439617, -0.5948203464774217886, -0.5144828988712590366, -0.4241447790355740864, -0.3321656903945551713, -0.2450932581667887604, -0.1676250517077815243, -0.1029297421028883111, -0.05304394686174657753, -0.01921352006354706144, -0.002141075122909895517};
// kLongMono[i][m]: coefficient m of the interpolating polynomial IN POWERS OF z = 4 y - 1 from the node values -- the cosine
// transform ((r == 0 ? 1 : 2) / 12 cos(r pi (i + 1/2) / 12): Chebyshev coefficient r) times the expansion of T_r in powers of
// z, formed in 40-digit arithmetic.  Round 6, last day: a node then evaluates g by Horner, two chains of five FMAs in z^2,
// instead of a 12-term Clenshaw sum (an FMA and a subtraction per term, one chain): 12 instead of 23 instructions per node of a
// hole.  Entries reach 562 and g is bounded by 1: 2e-13 of rounding in g (numpy against 40-digit values over k in [1e-8, 10]),
// beside the 12-point interpolation's own 1.4e-11 (in g; <= 7e-13 in I_y).
__device__ static const double kLongMono[kLongNT][kLongNT] = {
    {-0.01097104146561632112, -0.01106571014994634529, 0.7787537897977534927, 0.7854736255515628456, -8.423423384614512823, -8.496108772950288064, 30.7507912521321291, 31.01613861765528237, -44.5480629552044293, -44.93246643436705004, 22.08589184874843839, 22.27647013888880496},
    {0.0345177968644245874, 0.03736179409733042651, -2.444841259718689435, -2.646277110472540478, 26.13063966192888182, 28.28360055872290008, -93.09783532115045802, -100.7683708157189963, 129.5161133197968052, 140.1872308695338979, -60.33977866125205542, -65.31130579030865437},
    {-0.06394391566491336191, -0.08059954173942211059, 4.5023684289293516, 5.675111202376181658, -46.55956804149142378, -58.68705112453385624, 155.2015847409644934, 195.6273161766621412, -195.3975096083345678, -246.2931706276113185, 82.4256705100004938, 103.8952838841549891},
    {0.108602114403433813, 0.1783984812905714895, -7.526300685504181875, -12.36330083816194929, 70.91683363139958739, 116.4936381512825583, -197.8682514076311601, -325.0341463483171081, 216.7308429416679012, 356.0193412624425403, -82.4256705100004938, -135.3989700763708014},
    {-0.2011844635310912541, -0.5257203383164916734, 13.1115079263853561, 34.26202134059632493, -79.46397299526221516, -207.6493683150922371, 178.4311686544837914, 466.2631134871182447, -172.1827799864634719, -449.9352870395410811, 60.33977866125205542, 157.6754402152596064},
    {0.6329795093937625367, 4.849444380685177597, -8.421488199889589884, -64.51952712825610705, 37.39949112803968255, 286.5286307056946956, -73.41745791879879577, -562.4729923556317233, 65.88139628853776263, 504.7369816585899212, -22.08589184874843839, -169.2065896744636435},
    {0.6329795093937625367, -4.849444380685177597, -8.421488199889589884, 64.51952712825610705, 37.39949112803968255, -286.5286307056946956, -73.41745791879879577, 562.4729923556317233, 65.88139628853776263, -504.7369816585899212, -22.08589184874843839, 169.2065896744636435},
    {-0.2011844635310912541, 0.5257203383164916734, 13.1115079263853561, -34.26202134059632493, -79.46397299526221516, 207.6493683150922371, 178.4311686544837914, -466.2631134871182447, -172.1827799864634719, 449.9352870395410811, 60.33977866125205542, -157.6754402152596064},
    {0.108602114403433813, -0.1783984812905714895, -7.526300685504181875, 12.36330083816194929, 70.91683363139958739, -116.4936381512825583, -197.8682514076311601, 325.0341463483171081, 216.7308429416679012, -356.0193412624425403, -82.4256705100004938, 135.3989700763708014},
    {-0.06394391566491336191, 0.08059954173942211059, 4.5023684289293516, -5.675111202376181658, -46.55956804149142378, 58.68705112453385624, 155.2015847409644934, -195.6273161766621412, -195.3975096083345678, 246.2931706276113185, 82.4256705100004938, -103.8952838841549891},
    {0.0345177968644245874, -0.03736179409733042651, -2.444841259718689435, 2.646277110472540478, 26.13063966192888182, -28.28360055872290008, -93.09783532115045802, 100.7683708157189963, 129.5161133197968052, -140.1872308695338979, -60.33977866125205542, 65.31130579030865437},
    {-0.01097104146561632112, 0.01106571014994634529, 0.7787537897977534927, -0.7854736255515628456, -8.423423384614512823, 8.496108772950288064, 30.7507912521321291, -31.01613861765528237, -44.5480629552044293, 44.93246643436705004, 22.08589184874843839, -22.27647013888880496}};

// e^x for the walk's nodes: the weight and the density ratios of a node enter its value as FACTORS, so 4e-14 of relative error
// each is far inside the walk's budget (acceptance at 1e-7 of scale, results <= 1e-9; exp_fin is good to an ulp, which nothing
// downstream can use).  Degree 9 instead of 11 (Remez fit of 1 + r + r^2 p(r) on [-ln 2 / 2, ln 2 / 2]: 1.6e-14) and ONE reduction
// step -- the product n ln 2 is exact inside the FMA, what is lost is n (ln 2 - its double) = 2.3e-17 n: 14 instructions instead
// of 17, and three of every five instructions of the walk are these exponentials (round 6, late: 1 731 -> 1 596 per trip).
// The integer n is read from the LOW WORD of x / ln 2 + 1.5 2^52 instead of v_rndne + v_cvt_i32 (round 6, last day:
// 13 instructions; |n| < 2^31 is the caller's business -- exp_arg_clamp).  An LDS table of 2^(j/64) per lane of a
// wave (32 KB, conflict-free) with a degree-4 polynomial was measured too: 9 fp64 + 5 integer instructions + a ds_read_b64 per
// exponential, hydrodynamic 21.9 against 20.6 ms, constant 8.2 against 7.8 -- the reads' latency is not covered at two waves per SIMD;
// an Estrin form of the polynomial (chains of 5 instead of 9, one instruction more): 21.2 against 20.6.
__device__ __forceinline__ double exp_node(double x) {
    constexpr double kMagic = 0x1.8p52;
    const double t = fma(x, 1.4426950408889634, kMagic);
    const double n = t - kMagic;
    const double r = fma(n, -0.6931471805599453, x);
    double p = 0x1.710182df3d7acp-19;
    p = fma(p, r, 0x1.a16e32bc8180fp-16);
    p = fma(p, r, 0x1.a01b7383bafc4p-13);
    p = fma(p, r, 0x1.6c163be91fb17p-10);
    p = fma(p, r, 0x1.1111108e2cc07p-7);
    p = fma(p, r, 0x1.5555557deef18p-5);
    p = fma(p, r, 0x1.5555555589f00p-3);
    p = fma(p, r, 0x1.fffffffff13f6p-2);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, __double2loint(t));
}
// x limited to [-1e9, 700] by two INTEGER operations on its high word (2 cycles each on a SIMD with two waves; a v_min_f64 is 4 and
// limits one side): doubles above zero order like their high words taken as signed numbers, doubles below zero like unsigned
// ones.  As fmin(x, 700) for everything above zero: +Inf and the NaN the hardware makes (0x7FF8...) give 700 (<= 700.0002: the
// low word stays); below zero -Inf and a NaN with the sign bit set give -1e9 (e^x = 0, as the CPU restatement's libm has it for
// -Inf; exp_node answered NaN).
__device__ __forceinline__ double exp_arg_clamp(double x) {
    int hi = __double2hiint(x);
    hi = hi < 0x4085E000 ? hi : 0x4085E000;                               // 700 = 0x4085E00000000000
    const unsigned hu = (unsigned)hi < 0xC1CDCD65u ? (unsigned)hi : 0xC1CDCD65u;   // -1e9 = 0xC1CDCD6500000000
    return __hiloint2double((int)hu, __double2loint(x));
}
// 1 / x for the walk (tools/rcp_probe on gfx950: v_rcp_f64 is good to 4.6e-8; one Newton step leaves 2.2e-15, two -- recip_fast --
// 1.1e-16, and so does the cubic single pass r (1 + e + e^2) in three instructions instead of four).  A node's 1 / (1 + sum rho)
// is a factor of its value: one step; the reciprocal of a pair's e^d goes on into an exponent (u = e^c / e^d): the cubic pass.
__device__ __forceinline__ double recip_node(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double recip_cubic(double x) {
    const double r = __builtin_amdgcn_rcp(x), e = fma(-x, r, 1.0);
    return fma(fma(e, e, e), r, r);
}
// exp_fin with one reduction step, for the half widths of panels (0 <= x <= a few: n <= 3 loses 7e-17)
__device__ __forceinline__ double exp_small(double x) {
    constexpr double kMagic = 0x1.8p52;   // (n from the low word, as exp_node)
    const double t = fma(x, 1.4426950408889634, kMagic);
    const double n = t - kMagic;
    const double r = fma(n, -0.6931471805599453, x);
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
    return ldexp(p, __double2loint(t));
}
// the coefficients of g's interpolating polynomial (in powers of 4 y - 1) for shape k, left in the lane's own LDS slots sh[row0 + r][lane] (conflict-free, no barrier:
// a lane reads what it wrote)
template <int NROW>
__device__ __forceinline__ void conv_long_tab_build(double k, double (&sh)[NROW][kConvBlock], int row0) {
    double c[kLongNT];
#pragma unroll
    for (int r = 0; r < kLongNT; ++r) c[r] = 0.0;
#pragma unroll 1
    for (int i = 0; i < kLongNT; ++i) {
        // (1 - y)^k 2F1(2k, 1; k+1; y): inc_beta_cf(k, k, y) is the hypergeometric factor (y < 1/2: its range of convergence)
        const double g = exp_fin(k * kLongL1mY[i]) * inc_beta_cf(k, k, kLongY[i]);
#pragma unroll
        for (int r = 0; r < kLongNT; ++r) c[r] = fma(kLongMono[i][r], g, c[r]);
    }
#pragma unroll
    for (int r = 0; r < kLongNT; ++r) sh[row0 + r][threadIdx.x] = c[r];
}
// E_tau[K(s (1 - tau), s tau)] for x_t < s < 2 x_t from the table: lx = ln(x_t / s) (the rule has it for free),
// c0 = 1 / (k B(k, k)), rB = B(k+1, k+1) / B(k, k)
template <int NROW>
__device__ __forceinline__ double conv_long_G_mid_tab(const QArgs &Q, double k, double c0, double rB, double s, double lx,
                                                      const double (&sh)[NROW][kConvBlock], int row0) {
    const double xt = Q.kf[0], cb = Q.kf[1], ca = Q.kf[2];
    const double x = xt * recip_cubic(s), y = 1.0 - x;   // y in (0, 1/2)
    const double z = fma(4.0, y, -1.0), w = z * z;       // z = 4 y - 1 in [-1, 1]
    // (the lane index through an opaque copy: the twelve coefficients are READ here, per node -- left alone the compiler hoists
    // the reads out of the node loop and keeps them in 24 registers the walk does not have: scratch instead of LDS)
    int tl = threadIdx.x;
    asm volatile("" : "+v"(tl));
    static_assert(kLongNT == 12, "two Horner chains of six coefficients");
    double pe = sh[row0 + 10][tl], po = sh[row0 + 11][tl];
#pragma unroll
    for (int r = 8; r >= 0; r -= 2) {
        pe = fma(pe, w, sh[row0 + r][tl]);
        po = fma(po, w, sh[row0 + r + 1][tl]);
    }
    const double g = fma(po, z, pe);
    const double yk = exp_node(k * log_pos(y)) * c0;     // y^k / (k B(k, k))   (factors of I_y and D: exp_node's 4e-14 is theirs)
    const double Dk = yk * exp_node(k * lx);             // x^k y^k / (k B(k, k)) = D(k, k) / k
    const double Ikk = fma(-yk, g, 1.0);                 // I_x(k, k) = 1 - I_y(k, k)
    const double Ik1k1 = fma(Dk, fma(2.0, x, -1.0), Ikk);  // I_x(k+1, k+1) = I_x(k, k) - D/k + 2 x D/k
    const double P0 = fma(2.0, Ikk, -1.0), P1 = rB * fma(2.0, Ik1k1, -1.0);
    const double s2 = s * s;
    return fma(ca, s, fma(fma(cb, s2, -(ca * s)), P0, -2.0 * cb * s2 * P1));
}

// exp_fin (device_math.hpp: the same reduction, the same polynomial, the same bits) with its constants HELD in registers: the
// trapezoidal loop below is one exponential and five other operations per point, and the compiler rebuilt the ten coefficients
// with two v_mov_b32 each in every trip (16 of the 48 VALU instructions per point; round 6, tools/isa_loops.py) -- constants
// are "free to rematerialise" in its books, and a v_fmac overwrites its addend.  An empty asm makes each an opaque register
// value defined once, before the walk; v_fma_f64 then reads it in place.
struct ExpKept {
    double l2e, nh, nl, c[11];
    __device__ __forceinline__ ExpKept() {
        l2e = 1.4426950408889634, nh = -0.69314718055989033, nl = -5.497923018708371e-14;
        c[0] = 0x1.adeb8db5d7212p-26, c[1] = 0x1.28afdbfa89bf0p-22, c[2] = 0x1.71dedfc117959p-19, c[3] = 0x1.a019970598987p-16;
        c[4] = 0x1.a01a014a32d85p-13, c[5] = 0x1.6c16c18581530p-10, c[6] = 0x1.1111111121b01p-7, c[7] = 0x1.55555555500b2p-5;
        c[8] = 0x1.5555555555513p-3, c[9] = 0x1.000000000000bp-1, c[10] = 1.0;
        asm volatile("" : "+v"(l2e), "+v"(nh), "+v"(nl));
#pragma unroll
        for (int i = 0; i < 10; ++i) asm volatile("" : "+v"(c[i]));
    }
    __device__ __forceinline__ double operator()(double x) const {
        const double n = __builtin_rint(x * l2e);
        double r = fma(n, nh, x);
        r = fma(n, nl, r);
        double p = c[0];
#pragma unroll
        for (int i = 1; i < 10; ++i) p = fma(p, r, c[i]);
        p = fma(p, r, 1.0);
        p = fma(p, r, 1.0);
        return ldexp(p, (int)n);
    }
};
// T_m of a LOGNORMAL mode j: the sum of two Lognormal variates has no closed law, so two variables remain,
//   s = x + y,  t = ln(x / y):   f(x) f(y) dx dy = n^2 g(ln x) g(ln y) d(ln s) dt   (g: the normal density of ln x),
//   T_m = 1/2 n^2 int d(ln s) s^m (1 - w(s)) G2(ln s),   G2 = 2 int_0^inf dt K(x, y) g(ln x) g(ln y),
//   ln x = ln s - ln(1 + e^-t),  ln y = ln s - ln(1 + e^t)
// -- weighting_fn stays outside the inner integral (a function of s alone) and the hydrodynamic kink sits on the boundary
// t = 0.  Outer: the adaptive rule over ln s in [mu - 8.5 sigma, mu + 8.5 sigma + (gamma + 2) sigma^2 + ln 2] (budget
// kConvBudgetLn), marks at the other modes' cores (and ln x_t, ln 2 x_t for Long); inner: kLnPanels2 panels of nq
// Gauss-Legendre points over t in [0, max(ln s - mu, 0) + 12 sigma], split at the Long kernel's jump.  totals[m]: T_m with 1 - w = 1 (closed form), the estimate's scale.
// INNER PANELS (round 4): the integrand in t is a Gaussian sqrt(2) sigma ... 2 sigma wide around its peak (at t = 0 for
// s ~ 2 e^mu), so the inner range is cut into as many equal panels as make a panel no wider than kLnPanelSigmas sigma: 12 (the
// minimum) for sigma >= 0.03, up to kLnPanels2Max = 256 below 0.001.  Rounds 2-3 used 12 panels whatever sigma: measured against
// the rule with panels of sigma / 2 at s ~ 2 e^mu: 2e-10 of scale at sigma = 0.02, 6e-7 at 0.01, 1e-4 ... 1e-1 below 0.005; now
// <= 1.5e-12 down to sigma = 0.003, 6.5e-11 at 0.002, 1e-9 ... 2e-6 at 0.001 (the cap), tests/test_numerical_oracle.py.
constexpr int kLnPanels2 = 6, kLnPanels2Max = 256;   // (round 6: six panels of <= 3 sigma cover the bounded range 2 sqrt(84) sigma; 12 before the bound)
constexpr double kLnCut = 32.0;   // the density of the sum of two Lognormal variates is followed down to e^-32 of its peak (42 until late in round 6)
// The end of the inner range in t (round 6): sigma^2 E(t) = (d - qd(t))^2 + t^2 / 4 with qd = ln cosh(t / 2) >= sqrt(1 + t^2 / 4) - 1 =
// w - 1 (equal at 0; the derivative of the difference is tanh x - x / sqrt(1 + x^2) >= 0), and E_min <= E(0) = d^2 / sigma^2:
// wherever w - 1 >= d, sigma^2 E >= (w - 1 - d)^2 + w^2 - 1 > d^2 + kLnCut sigma^2 for every
// w > max(1 + max(d, 0), [(1 + d) + sqrt((d - 1)^2 + 2 kLnCut sigma^2)] / 2).  The older bounds (t^2 / 4 alone; m + 12 sigma) stay as
// well.  At d = 0, sigma = ln 2: 6.3 against 7.8 (the true end: 4.9), about half for d < 0; a few per cent for sigma << 1.
__device__ __forceinline__ double conv_ln_inner_top(double m, double d, double sg) {
    const double cs = kLnCut * (sg * sg), dm = d - 1.0;
    const double wr = 0.5 * ((1.0 + d) + sqrt(fma(dm, dm, 2.0 * cs)));
    const double w = fmax(wr, 1.0 + fmax(d, 0.0));
    const double T = fmin(fmax(m, 0.0) + 12.0 * sg, 2.0 * sqrt(fma(d, d, cs)));
    return fmin(T, 2.0 * sqrt(fma(w, w, -1.0)));
}
constexpr double kLnPanelSigmas = 3.0;
template <int N, int KIND>
__device__ __forceinline__ void conv_T_lognormal(const QArgs &Q, const double *__restrict__ tab, double n, double mu, double sg,
                                                 const double (&cm)[N], const double (&wm)[N], const ConvLogDensity (&lg)[N],
                                                 int j, const double (&totals)[3], double &T0, double &T1, double &T2) {
    const int nq = Q.nq;
    constexpr double gtop = KIND == KF_LINEAR ? 1.0 : KIND == KF_HYDRODYNAMIC ? 4.0 / 3.0 : KIND == KF_LONG ? 2.0 : 0.0;
    const double L0 = fma(-8.5, sg, mu), L1 = mu + 8.5 * sg + (gtop + 2.0) * (sg * sg) + 0.6931471805599453;
    const double c2 = 0.5 / (sg * sg), nrm = c2 * 0.3183098861837907, pref = 0.5 * (n * n);
    ConvMarks<(N > 1 ? N - 1 : 1)> mk;
    mk.shift = 0.0;
    mk.extra[0] = INFINITY;
    mk.extra[1] = KIND == KF_LONG ? log_pos(Q.kf[0]) : INFINITY;
    mk.extra[2] = KIND == KF_LONG ? log_pos(2.0 * Q.kf[0]) : INFINITY;
    {
        int slot = 0;
#pragma unroll
        for (int m = 0; m < N; ++m)
            if (m != j) {
#pragma unroll
                for (int sl = 0; sl < N - 1; ++sl)
                    if (sl == slot) mk.core(sl, cm[m], wm[m], (L1 - L0) * (1.0 / double(kConvNInit)));
                ++slot;
            }
    }
    const auto node = [&](double ls, double (&vals)[3]) {
        const double s = exp_fin(ls);
        // Round 6: the Gaussian factor of the inner integrand is that of the polynomial kernels (conv_T_lognormal_poly) whatever
        // the kernel function, so its two bounds hold here too: nothing where min(d^2, 2 d - 1) > kLnCut sigma^2, nothing beyond
        // T = 2 sqrt(d^2 + kLnCut sigma^2).  (Without them a shape clamped to sigma = eps ran 256 panels x nq points at EVERY node while
        // the 63 other parcels of its wave waited: 2e5 parcel-RHS/s against 3.5e7 for the polynomial kernels.)
        const double md = ls - mu, dd = md - 0.6931471805599453;
        if ((dd <= 1.0 ? dd * dd : fma(2.0, dd, -1.0)) > kLnCut * (sg * sg)) {
            vals[0] = vals[1] = vals[2] = 0.0;
            return;
        }
        const double Tm = conv_ln_inner_top(md, dd, sg);
        const ExpKept ek;   // (its 13 constants live for the point loops of this node only)
        // the Long kernel jumps where the larger particle x = s / (1 + e^-t) crosses x_t: at t_b = ln(x_t / (s - x_t)) for
        // x_t < s < 2 x_t (below, both stay under x_t; above, x >= s / 2 >= x_t) -- the inner panels are split there
        double tb = 0.0;
        int n1 = 0;
        double npd = ceil(Tm / (kLnPanelSigmas * sg));
        npd = !(npd >= double(kLnPanels2)) ? double(kLnPanels2) : npd > double(kLnPanels2Max) ? double(kLnPanels2Max) : npd;
        const int np2 = (int)npd;
        if (KIND == KF_LONG && s > Q.kf[0] && s < 2.0 * Q.kf[0]) {
            tb = log_pos(Q.kf[0] / (s - Q.kf[0]));
            if (tb > 0.0 && tb < Tm) {
                n1 = (int)(double(np2) * (tb / Tm) + 0.5);
                n1 = n1 < 1 ? 1 : n1 > np2 - 1 ? np2 - 1 : n1;
            }
        }
        double G2 = 0.0;
#pragma unroll 1
        for (int i2 = 0; i2 < np2; ++i2) {
            // panel i2 of the first piece [0, t_b] (n1 panels) or of the second [t_b, Tm] (the rest); n1 = 0: one piece
            const bool first = i2 < n1;
            const double a = first ? 0.0 : (n1 ? tb : 0.0), b = first ? tb : Tm;
            const double h2 = (b - a) / double(first ? n1 : np2 - n1);
            const double tc = fma(h2, double(first ? i2 : i2 - n1) + 0.5, a);
#pragma unroll 1
            for (int g2 = 0; g2 < nq; ++g2) {
                const double t = fma(0.5 * h2, tab[g2], tc);
                // Round 6 (late): ~300 -> ~130 instructions per point.  ln(1 + e^-t) through log_pos(1 + v) -- its ABSOLUTE error
                // (1e-16) is what ln x and ln y carry, the library log1p's relative accuracy (120 instructions) bought nothing;
                // x = s / (1 + v), y = x v instead of two exponentials (Long); the kernel's power of x y joins the Gaussian's
                // exponent (hydrodynamic: one exponential instead of two), 1 / e^(t/3) by a Newton reciprocal instead of a division.
                const double v1 = ek(-t), opv = 1.0 + v1, spm = log_pos(opv);   // ln(1 + e^-t); ln(1 + e^t) = t + ln(1 + e^-t)
                const double lx = ls - spm, ly = lx - t, dx = lx - mu, dy = ly - mu;
                const double ge = -fma(dx, dx, dy * dy) * c2;
                double Kv;
                if (KIND == KF_CONSTANT) {
                    Kv = Q.kf[0] * ek(ge);
                } else if (KIND == KF_LINEAR) {
                    Kv = (Q.kf[0] * s) * ek(ge);
                } else if (KIND == KF_HYDRODYNAMIC) {
                    // K = C (xy)^(2/3) (e^(2t/3) + 2 e^(t/3) - 2 e^(-t/3) - e^(-2t/3)),  x / y = e^t
                    const double e1 = ek(t * (1.0 / 3.0)), r1 = recip_fast(e1);
                    Kv = (Q.kf[0] * 0.46526286817455001) * ek(fma(2.0 / 3.0, lx + ly, ge)) *
                         (fma(e1, e1, 2.0 * e1) - fma(r1, r1, 2.0 * r1));
                } else {
                    const double x = s * recip_fast(opv), y = x * v1;
                    Kv = ((x < Q.kf[0] && y < Q.kf[0]) ? Q.kf[1] * fma(x, x, y * y) : Q.kf[2] * (x + y)) * ek(ge);
                }
                G2 = fma(0.5 * h2 * tab[nq + g2], Kv, G2);
            }
        }
        const double v = conv_one_minus_w<N>(lg, j, s, ls) * (2.0 * nrm * G2);
        vals[0] = v;
        vals[1] = v * s;
        vals[2] = (v * s) * s;
    };
    const int est_idx[3] = {0, 1, 2};
    const double rp = 1.0 / pref;
    const double scaleS[3] = {totals[0] * rp, totals[1] * rp, totals[2] * rp};
    double T[3] = {0.0, 0.0, 0.0};
    conv_adaptive<3, 3, false>(L0, L1, mk, est_idx, scaleS, kConvBudgetLn, node, node, T);
    T0 = T[0] * pref;
    T1 = T[1] * pref;
    T2 = T[2] * pref;
}

// ---- T_m of a LOGNORMAL mode under a POLYNOMIAL kernel (constant, linear): the inner integral by a trapezoidal rule (round 5) ----
// the reference's own n_particles_lognorm.jl (two Lognormal modes, sigma = ln 2, linear kernel) ran at
// 2.3e5 ... 4e5 parcel-RHS/s -- 21 M VALU instructions per parcel at 0.28 active lanes (round 5 PMC): >= 96 Gauss-Legendre
// points per outer node, each with a log1p, two exponentials and the generic kernel, and up to 2048 of them for a shape clamped
// to sigma = eps, whose lane the other 63 of its wave waited for.  For K = c s^gamma (gamma = 0, 1) the kernel leaves the inner
// integral, which is then the density of S = X + Y in ln s,
//     G2(ln s) = 2 c s^gamma int_0^inf g(ln x) g(ln y) dt = c s^gamma / (pi sigma^2) int_0^inf exp(-[(m - q(t))^2 + t^2 / 4] / sigma^2) dt,
//     m = ln s - mu,   q(t) = ln(2 cosh(t / 2))   (ln x - mu = m + t/2 - q, ln y - mu = m - t/2 - q),
// an EVEN, analytic integrand (poles of q at t = +- i pi) that decays like a Gaussian: the trapezoidal rule converges
// geometrically.  Step h = min(sigma, 1/2): <= 4e-13 of the density's peak against 30-digit mpmath for sigma <= 0.3 and sigma >= ln 2,
// rising to 3e-10 at sigma = 1/2, where the step is at both of its limits (0.4: 1.4e-11, 0.45: 7e-11, 0.55: 3e-11, 0.6: 4e-12; round 6
// re-measured -- round 5's "2e-13 from 0.002 to 2" had not sampled that neighbourhood; 1.2 sigma: 3e-12 ... 5e-8; 1.4 sigma: 4e-9).  Range: the integrand is below e^-kLnCut of its maximum beyond
// T = min(max(m, 0) + 12 sigma, 2 sqrt((m - ln 2)^2 + kLnCut sigma^2)) (E(t) >= t^2 / (4 sigma^2), E_min <= E(0)); and the whole
// density is below e^-kLnCut of its peak where min(d^2, 2 d - 1) > kLnCut sigma^2, d = m - ln 2 (q <= ln 2 + t^2 / 8): zero there --
// which also bounds the number of points by ~20 for every sigma <= 1/2, clamped shapes included.  q(i h) does not depend on s:
// the first kLnQTab values wait in the lane's own LDS slots (conflict-free, no barrier), further ones (sigma > 1/2 only) are
// computed.  Per inner point: one LDS read, one exponential, four FMAs.  The outer rule is the adaptive walk over ln s of
// conv_T_lognormal (same range, marks and tolerance).
constexpr int kLnQTab = 24;
// exp_node (below: degree 9, one reduction step, 4e-14) with its constants held the same way: the trapezoidal rule's terms are
// summed as they come, so that is their relative error and the sum's (round 6, late: 28 -> 25 VALU instructions per point).
struct ExpKept9 {
    double l2e, nl2, c[8];
    __device__ __forceinline__ ExpKept9() {
        l2e = 1.4426950408889634, nl2 = -0.6931471805599453;
        c[0] = 0x1.710182df3d7acp-19, c[1] = 0x1.a16e32bc8180fp-16, c[2] = 0x1.a01b7383bafc4p-13, c[3] = 0x1.6c163be91fb17p-10;
        c[4] = 0x1.1111108e2cc07p-7, c[5] = 0x1.5555557deef18p-5, c[6] = 0x1.5555555589f00p-3, c[7] = 0x1.fffffffff13f6p-2;
        asm volatile("" : "+v"(l2e), "+v"(nl2));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(c[i]));
    }
    __device__ __forceinline__ double operator()(double x) const {
        const double n = __builtin_rint(x * l2e);
        const double r = fma(n, nl2, x);
        double p = c[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) p = fma(p, r, c[i]);
        p = fma(p, r, 1.0);
        p = fma(p, r, 1.0);
        return ldexp(p, (int)n);
    }
};
template <int N, int KIND>
__device__ __forceinline__ void conv_T_lognormal_poly(const QArgs &Q, double n, double mu, double sg, const double (&cm)[N],
                                                      const double (&wm)[N], const ConvLogDensity (&lg)[N], int j,
                                                      const double (&totals)[3], double &T0, double &T1, double &T2, int &cost) {
    static_assert(KIND == KF_CONSTANT || KIND == KF_LINEAR, "polynomial kernels only");
    constexpr double gtop = KIND == KF_LINEAR ? 1.0 : 0.0, ln2 = 0.6931471805599453;
    __shared__ double sh_q[kLnQTab + 1][kConvBlock];
    const int lane = threadIdx.x;
    const double L0 = fma(-8.5, sg, mu), L1 = mu + 8.5 * sg + (gtop + 2.0) * (sg * sg) + ln2;
    const double c1 = 1.0 / (sg * sg), h = fmin(sg, 0.5), rh = 1.0 / h, pref = 0.5 * (n * n);
    const double eh = exp_fin(-h);
    {   // q(i h) - ln 2 = i h / 2 + log1p(e^(-i h)) - ln 2, i = 1 .. kLnQTab
        double v = 1.0;
#pragma unroll 1
        for (int i = 1; i <= kLnQTab; ++i) {
            v *= eh;
            sh_q[i - 1][lane] = fma(0.5 * h, double(i), log1p(v) - ln2);
        }
    }
    ConvMarks<(N > 1 ? N - 1 : 1)> mk;
    mk.shift = 0.0;
    mk.extra[0] = mk.extra[1] = mk.extra[2] = INFINITY;
    {
        int slot = 0;
#pragma unroll
        for (int m = 0; m < N; ++m)
            if (m != j) {
#pragma unroll
                for (int sl = 0; sl < N - 1; ++sl)
                    if (sl == slot) mk.core(sl, cm[m], wm[m], (L1 - L0) * (1.0 / double(kConvNInit)));
                ++slot;
            }
    }
    const double kc = Q.kf[0] * (0.3183098861837907 * c1);   // c / (pi sigma^2)
    const ExpKept9 ek;
    const auto node = [&](double ls, double (&vals)[3]) {
        const double m = ls - mu, d = m - ln2;
        const double lb = (d <= 1.0 ? d * d : fma(2.0, d, -1.0)) * c1;   // a lower bound of the exponent over t
        double sum = 0.0;
        if (lb <= kLnCut) {
            const double Tm = conv_ln_inner_top(m, d, sg);
            const int npt = (int)ceil(Tm * rh);
            sum = 0.5 * exp_fin(-(d * d) * c1);   // t = 0
            const double hq = 0.25 * (h * h);
            // the points served by the table, then (sigma > 1/2 only) the ones beyond it: two loops, so that the first carries
            // neither the second's e^(-t) nor its select (round 6: 48 -> 27 VALU instructions per point together with ek)
            const int ntab = npt < kLnQTab ? npt : kLnQTab;
            // the exponent of point i: -c1 [(d - qd_i)^2 + (h i / 2)^2]; its second term by differences (e2 = c1 hq i^2: two additions
            // a point instead of an addition and two products), the first as (-c1 dq) dq
            const double nc1 = -c1, g2 = 2.0 * (c1 * hq);
            double e2 = 0.0, st = c1 * hq, qn = sh_q[0][lane];
#pragma unroll 1
            for (int i = 0; i < ntab; ++i) {
                double dq = d - qn;
                qn = sh_q[i + 1][lane];   // (the next trip's value, asked for before this trip's arithmetic; row kLnQTab is padding)
                asm volatile("" : "+v"(dq) : : "memory");   // (the read is issued HERE, before the arithmetic on dq: without the fence
                                                            // the compiler moves it to the end of the trip, four instructions before its use)
                e2 += st;
                st += g2;
                sum += ek(fma(nc1 * dq, dq, -e2));
            }
            if (npt > kLnQTab) {
                double vv = exp_fin(-h * double(kLnQTab)), fi = double(kLnQTab);   // e^(-t) at the end of the table
#pragma unroll 1
                for (int i = kLnQTab + 1; i <= npt; ++i) {
                    vv *= eh;
                    fi += 1.0;
                    const double dq = d - fma(0.5 * h, fi, log1p(vv) - ln2);
                    e2 += st;
                    st += g2;
                    sum += ek(fma(nc1 * dq, dq, -e2));
                }
            }
        }
        const double s = exp_fin(ls);
        double v = conv_one_minus_w<N>(lg, j, s, ls) * (kc * h * sum);   // (1 - w) x G2 / s^gamma
        if (KIND == KF_LINEAR) v *= s;
        vals[0] = v;
        vals[1] = v * s;
        vals[2] = (v * s) * s;
    };
    const int est_idx[3] = {0, 1, 2};
    const double rp = 1.0 / pref;
    const double scaleS[3] = {totals[0] * rp, totals[1] * rp, totals[2] * rp};
    double T[3] = {0.0, 0.0, 0.0};
    // (an evaluation of this rule -- 15 nodes x ~20 inner points -- costs about four of a Gamma-weight rule's: the hint's unit)
    cost += 4 * conv_adaptive<3, 3, false>(L0, L1, mk, est_idx, scaleS, kConvBudgetLn, node, node, T);
    T0 = T[0] * pref;
    T1 = T[1] * pref;
    T2 = T[2] * pref;
}

// ---- ONE walk over all the Gamma-weight rules of a parcel (round 4) ----
// A wave runs a rule for as long as its lane with the most panel evaluations; run rule by rule, a wave of the cfg4q batch
// keeps 0.69 of its lanes busy (tools/conv_lab: 48 evaluations per parcel on average, 69 for the slowest lanes of a wave).
// Here a lane that has finished the rule of mode j goes straight on to the rule of its next mode while its neighbours
// still work on theirs: the bound becomes the lanes' TOTAL evaluations (0.81).  Everything a rule needs is prepared
// before the loop, in straight-line code every lane runs through together (ConvRule: ~12 doubles per rule); the switch
// inside the loop is a handful of selects.  (Round 3 tried this with the rules' set-up -- lgamma, two logarithms, the marks
// -- inside the loop: 328 registers and the set-up executed, masked, in almost every trip: slower.)  The arithmetic of
// every rule is that of conv_adaptive<3, 3, false> with the node function below: results are bit-identical to running
// the rules one after the other.
template <int V>
struct ConvInt {
    static constexpr int value = V;
};
template <int NM>
struct ConvRule {
    bool valid;
    double A, lgA, th, lnth, tlo, thi;
    double gl;                     // Long: (k + 1) / (2k + 1), the factor of G below x_t
    double kj, lgB, rB, ex1, ex2;  // Long: the shape, -ln B(k, k), B(k+1, k+1) / B(k, k), the kinks of G in the rule's variable
    double c0;                     // Long: 1 / (k B(k, k))  (conv_long_G_mid_tab)
    double sc[3];                  // Long: the scale of each output (T_m with 1 - w = 1, over the prefactor)
    // the bound of what is left of the integral below an edge (the walk goes DOWN and stops early, see conv_T_merged):
    double tmode, lwmode;          // ln A, ln of the weight's maximum
    double ltlo[NM];               // ln rho of the other modes at t_lo
    bool convex;                   // every mode above j is Gamma-family with theta >= theta_j: ln rho is convex in t
    double mc[NM], mw[NM];         // the other modes' cores (ln mean size, width)
    int mI[NM];
};

// Everything the walk needs of the rule of mode j, prepared before the loop in straight-line code every lane runs through
// together.  sc: the Long kernel's output scales (the caller has the self-pair closed forms they come from).  The same function
// prepares the rules of phase 1 (on the lane that owns the parcel) and, for a Long plan whose holes are walked by another lane
// (conv_coal_ints_long_split), those of phase 2 again from (n, theta, k) -- so the two agree to the bit whichever lane runs them.
template <int N, int KIND>
__device__ __forceinline__ void conv_rule_prepare(const QArgs &Q, int j, bool lnj, double nj, double kj, double thj, double lnthj,
                                                  double lgkj, const double (&cm)[N], const double (&wm)[N],
                                                  const ConvLogDensity (&lg)[N], const double (&sc)[3],
                                                  ConvRule<(N > 1 ? N - 1 : 1)> &r) {
    constexpr int NM = N > 1 ? N - 1 : 1;
    constexpr double gam = KIND == KF_LINEAR ? 1.0 : KIND == KF_HYDRODYNAMIC ? 4.0 / 3.0 : 0.0;
    r.valid = !lnj && nj > 0.0;
    r.A = r.lgA = r.th = 1.0;
    r.lnth = r.tlo = r.thi = r.kj = r.lgB = r.rB = r.gl = 0.0;
    r.ex1 = r.ex2 = INFINITY;
    r.c0 = 0.0;
    r.sc[0] = r.sc[1] = r.sc[2] = 1.0;
    r.tmode = r.lwmode = 0.0;
    r.convex = false;
#pragma unroll
    for (int sl = 0; sl < NM; ++sl) {
        r.ltlo[sl] = 0.0;
        r.mc[sl] = 0.0;
        r.mw[sl] = 1.0;
        r.mI[sl] = 0;
    }
    if (lnj) return;  // (wave-uniform: the closure family of a mode is a plan constant)
    const double Ash = fma(2.0, kj, gam);  // homogeneous kernels: T_m = 1/2 s0 E_{Gamma(2k + gamma)}[s^m (1 - w)]
    r.A = Ash;
    r.lgA = lgamma_pos(Ash);
    r.th = thj;
    r.lnth = lnthj;
    conv_range(Ash, 2.0, r.lgA, r.tlo, r.thi);
    // marks: the other modes' cores; Long: s = x_t and 2 x_t (kinks of G)
    r.ex1 = KIND == KF_LONG ? log_pos(Q.kf[0] / thj) : INFINITY;
    r.ex2 = KIND == KF_LONG ? log_pos(2.0 * Q.kf[0] / thj) : INFINITY;
    {
        ConvMarks<NM> mk;
        int slot = 0;
#pragma unroll
        for (int m = 0; m < N; ++m)
            if (m != j) {
#pragma unroll
                for (int sl = 0; sl < N - 1; ++sl)
                    if (sl == slot) {
                        mk.shift = lnthj;
                        mk.core(sl, cm[m], wm[m], (r.thi - r.tlo) * (1.0 / double(kConvNInit)));
                        r.mc[sl] = cm[m];
                        r.mw[sl] = wm[m];
                        r.mI[sl] = mk.I[sl];
                    }
                ++slot;
            }
    }
    {
        // for the bound of what is left below an edge (conv_T_merged): the weight's mode, ln rho of the others at t_lo
        r.tmode = log_pos(Ash);
        r.lwmode = Ash * r.tmode - Ash - r.lgA;
        const double s_lo = exp_fin(r.tlo) * thj, ls_lo = r.tlo + lnthj;
        double own_lo = 0.0;
#pragma unroll
        for (int m = 0; m < N; ++m)
            if (m == j) own_lo = lg[m](s_lo, ls_lo);
        bool cvx = true;
        int slot = 0;
#pragma unroll
        for (int m = 0; m < N; ++m)
            if (m != j) {
#pragma unroll
                for (int sl = 0; sl < N - 1; ++sl)
                    if (sl == slot) r.ltlo[sl] = lg[m](s_lo, ls_lo) - own_lo;
                double bj = 0.0;
#pragma unroll
                for (int m2 = 0; m2 < N; ++m2)
                    if (m2 == j) bj = lg[m2].b;
                if (m > j && (lg[m].lognormal || !(lg[m].b <= bj))) cvx = false;
                ++slot;
            }
        r.convex = cvx;
    }
    r.kj = kj;
    r.gl = (kj + 1.0) / fma(2.0, kj, 1.0);
    if (KIND == KF_LONG) {
        r.lgB = lgamma_pos(2.0 * kj) - 2.0 * lgkj;  // -ln B(k, k)
        r.rB = kj / (2.0 * fma(2.0, kj, 1.0));      // B(k+1, k+1) / B(k, k)
        r.c0 = exp_fin(r.lgB) / kj;                  // 1 / (k B(k, k))
        r.sc[0] = sc[0];
        r.sc[1] = sc[1];
        r.sc[2] = sc[2];
    }
}

// PHASE (the Long kernel only; 0 for the others): the range x_t < s < 2 x_t of a Long rule -- where G(s) needs the incomplete
// beta table, ~115 instructions per node on top of the ~130 of any node -- is a few panels of a rule, but the lanes of a wave
// reach theirs in different trips, so a walk that takes the panels in order has some lane there in almost every trip and the
// whole wave pays for the table code in all of them (round 5, first version: 22.2 ms per 4e6 parcels, 172 k VALU instructions
// per parcel against 73 k for the hydrodynamic kernel).  The walk is therefore split: PHASE 1 walks every rule of the parcel
// from t_hi down to t_lo and JUMPS over [x_t, 2 x_t] (the "hole": both ends are forced panel edges) -- G is c_a s above and
// c_b s^2 (k+1)/(2k+1) below, the loop is the hydrodynamic kernel's; PHASE 2, a second loop every lane of the wave enters
// together, walks the holes of the parcel's rules from 2 x_t down to x_t with the table code unconditional.  The accumulated
// sums of a rule carry over (the acceptance test of a hole's panels compares with everything outside the hole); the bound
// that ends a rule early covers the hole while the walk is above it (midneed[r] = false then: the hole is negligible too).
// NRULES (round 6): the number of rules in `rb`.  N - 1 (the default): the merged walk of every rule of the parcel, rule r = mode r.
// 1 with N > 2 (plans of more than CLOUDY_AOT_MAX_MODES modes, conv_coal_ints_rolled): ONE rule, that of mode `jsingle` (a run-time
// index) -- the prepared rules of seven modes do not fit a lane's registers (2 303 spilled for N = 8), so such plans walk their rules
// one after the other and prepare each just before its walk; nothing but the rule in hand is live in the loop.
template <int N, int KIND, bool LTAB, int PHASE, int NROW, int NRULES = (N > 1 ? N - 1 : 1)>
__device__ __forceinline__ void conv_T_merged(const QArgs &Q, const ConvLogDensity (&lg)[N],
                                              const ConvRule<(N > 1 ? N - 1 : 1)> (&rb)[NRULES],
                                              double (&Traw)[NRULES][3], bool (&midneed)[NRULES],
                                              const double (&gtab)[NROW][kConvBlock], int &cost, int jsingle = 0) {
    constexpr int NM = N > 1 ? N - 1 : 1, NR = NRULES;
    constexpr bool kSingle = NRULES == 1 && NM > 1;
    static_assert(NRULES == NM || NRULES == 1, "all the rules of the parcel, or one");
    static_assert(!(kSingle && LTAB), "the per-rule tables belong to the merged walk");
    static_assert((KIND == KF_LONG) == (PHASE != 0), "the Long kernel's rules are walked in two phases, the others in one");
    // The Long kernel's G(s) behaves like (s - x_t)^k just above x_t (the Beta(k, k) law of tau ends like tau^(k-1)), so K15
    // converges only algebraically in the panel whose lower edge is x_t; rounds 3-4 ran the Long rules at a hundredth of the
    // tolerance for it (43 against 30 panel evaluations per parcel, 1e-9 of scale on random mixtures, shapes below 0.1 not
    // converged at the depth limit).  Round 5: THAT initial panel [a0, a0 + h] is integrated in xi, t = a0 + h xi^4,
    // dt = 4 h xi^3 d xi -- integrand x Jacobian ~ xi^(4k + 3) -- and the bisection works on xi in [0, 1]; every other panel has
    // t = a0 + h xi.  Same tolerance as the other kernels now: 33 evaluations per parcel, <= 5e-12 of scale on 600 random
    // mixtures with shapes down to 1e-3 (against the rule at 1e-13).
    constexpr double kTolT = kConvTol;
    bool sing = false;   // PHASE 2: the initial panel in hand has x_t as its lower edge
    double stop = 0.0;   // the lower end of the walk: t_lo, or the lower end of the hole (PHASE 2)
    double hlo = 0.0, hhi = 0.0;   // the hole [x_t, 2 x_t] in the rule's variable, clamped to [t_lo, t_hi]
    bool mid_flag = false;         // PHASE 1, set by next_panel when a rule ends: its hole has to be walked
    // ---- the state of the rule in hand
    int j = -1, jm = -1;   // the rule in hand and its mode (the same number in the merged walk)
    double A = 1.0, lgA = 0.0, thj = 1.0, lnthj = 0.0, tlo = 0.0, kj = 1.0, lgB = 0.0, rB = 0.0, gl = 0.0, c0 = 0.0;
    double out[3] = {0.0, 0.0, 0.0};
    double flo[3] = {kConvFloor, kConvFloor, kConvFloor};   // kConvFloor x scaleS, formed when a rule is taken
    double tmode = 0.0, lwmode = 0.0, lnup = 0.0, ltlo[NM];
    bool convex = false;
    ConvLogDensity own, oth[NM];
    // ln rho of a Gamma-family mode against the rule's own (Gamma-family) mode is linear in (ln s, s): its three coefficients,
    // differences formed once per rule -- two FMAs per node and slot instead of two log densities and their difference (and
    // without the cancellation of that difference); Lognormal modes keep the general form.  (Round 4: -7 of ~118 instructions
    // per node for an all-Gamma plan, where `own` and `oth` then drop out of the loop.)
    double da[NM], nb[NM];
    // (round 6, last day: the same line in the rule's own variables, ln rho = da t + (nb th_j) u + (nc + da ln th_j) -- a node of an
    // all-Gamma plan then needs neither s nor ln s for its density ratios)
    double ncu[NM];
    double upw[NM];   // 1 for the slots of the modes above the rule's own (they make up 1 - w), 0 below: one FMA instead of a select
    bool anyln = false;
    ConvMarks<NM> mk;
    double h0 = 0.0, gap = 0.0, cur = 0.0, a0 = 0.0, h = 0.0;
    int io = 1, L = 0, budget = kConvBudget;
    unsigned i = 0;
    bool busy = true;
    own = lg[0];
#pragma unroll
    for (int sl = 0; sl < NM; ++sl) oth[sl] = lg[0];
#pragma unroll
    for (int m = 0; m < N; ++m) anyln = anyln || lg[m].lognormal;
    mk.shift = 0.0;
    mk.extra[0] = mk.extra[1] = mk.extra[2] = INFINITY;
#pragma unroll
    for (int sl = 0; sl < NM; ++sl) {
        ltlo[sl] = 0.0;
        da[sl] = nb[sl] = 0.0;
        ncu[sl] = 0.0;
        upw[sl] = 0.0;
        mk.c[sl] = 0.0;
        mk.w[sl] = 1.0;
        mk.I[sl] = 0;
        mk.q[sl] = 0;
        mk.v[sl] = INFINITY;
    }
    // the next valid rule after j for the lanes with need = true (busy = false: none left); selects only
    // (rmin, a compile-time fact at each call site: the first call, before the loop, may take any rule; a call inside the loop
    // comes from a lane that has finished one, so rule 0 cannot be its answer -- round 6: said so, the prepared rule 0, ~44
    // registers for a three-mode plan, is dead across the loop instead of being carried through it and spilled around it)
    const auto next_rule = [&](bool need, auto rmin_c) {
        constexpr int rmin = decltype(rmin_c)::value;
        int jn = NR;
        // (what only the block `if (go)` below reads is local: carried through the walk's loop in a chain of selects it would stay
        // live across it -- the compiler does not see that `go` implies one of the selects)
        double nc[NM];   // c_j - c_m of the rule taken: every slot is set when one is
        double thi = 0.0, scaleS[3] = {1.0, 1.0, 1.0};
#pragma unroll
        for (int sl = 0; sl < NM; ++sl) nc[sl] = 0.0;
#pragma unroll
        for (int r = NR - 1; r >= rmin; --r)
            if (rb[r].valid && (PHASE != 2 || midneed[r]) && r > j) jn = r;
        const bool go = need && jn < NR;
#pragma unroll
        for (int r = rmin; r < NR; ++r) {
            const bool sel = go && jn == r;
            A = sel ? rb[r].A : A;
            lgA = sel ? rb[r].lgA : lgA;
            thj = sel ? rb[r].th : thj;
            lnthj = sel ? rb[r].lnth : lnthj;
            tlo = sel ? rb[r].tlo : tlo;
            thi = sel ? rb[r].thi : thi;
            tmode = sel ? rb[r].tmode : tmode;
            lwmode = sel ? rb[r].lwmode : lwmode;
            convex = sel ? rb[r].convex : convex;
            const int rm = kSingle ? jsingle : r;   // the mode of rule r
            {   // ln of the number of modes above (round 6: to ln 7; ln 3 stood in for every count >= 3 -- plans of up to 4 modes)
                const int nup = N - 1 - rm;
                const double l = nup <= 1 ? 0.0 : nup == 2 ? 0.6931471805599453 : nup == 3 ? 1.0986122886681098 : nup == 4 ? 1.3862943611198906
                                 : nup == 5 ? 1.6094379124341003 : nup == 6 ? 1.791759469228055 : 1.9459101090932196;
                lnup = sel ? l : lnup;
            }
#pragma unroll
            for (int sl = 0; sl < NM; ++sl) ltlo[sl] = sel ? rb[r].ltlo[sl] : ltlo[sl];
            if (KIND == KF_LONG) {
                kj = sel ? rb[r].kj : kj;
                gl = sel ? rb[r].gl : gl;
                lgB = sel ? rb[r].lgB : lgB;
                rB = sel ? rb[r].rB : rB;
                c0 = sel ? rb[r].c0 : c0;
                mk.extra[1] = sel ? rb[r].ex1 : mk.extra[1];
                mk.extra[2] = sel ? rb[r].ex2 : mk.extra[2];
                scaleS[0] = sel ? rb[r].sc[0] : scaleS[0];
                scaleS[1] = sel ? rb[r].sc[1] : scaleS[1];
                scaleS[2] = sel ? rb[r].sc[2] : scaleS[2];
            }
#pragma unroll
            for (int sl = 0; sl < NM; ++sl) {
                mk.c[sl] = sel ? rb[r].mc[sl] : mk.c[sl];
                mk.w[sl] = sel ? rb[r].mw[sl] : mk.w[sl];
                mk.I[sl] = sel ? rb[r].mI[sl] : mk.I[sl];
            }
            // the densities of the rule's own mode (rm) and of the others, ascending (slot sl is mode sl < rm ? sl : sl + 1)
            double ra = 0.0, rbb = 0.0, rc = 0.0;   // the own mode's coefficients
#pragma unroll
            for (int m = 0; m < N; ++m) {
                const bool is_own = m == rm;
                ra = is_own ? lg[m].a : ra;
                rbb = is_own ? lg[m].b : rbb;
                rc = is_own ? lg[m].c : rc;
            }
#pragma unroll
            for (int m = 0; m < N; ++m) {
                const bool is_own = m == rm;
                own.a = (sel && is_own) ? lg[m].a : own.a;
                own.b = (sel && is_own) ? lg[m].b : own.b;
                own.c = (sel && is_own) ? lg[m].c : own.c;
                own.lognormal = (sel && is_own) ? lg[m].lognormal : own.lognormal;
                const int sl = m < rm ? m : m - 1;
#pragma unroll
                for (int s2 = 0; s2 < NM; ++s2) {
                    const bool hit = sel && !is_own && s2 == sl;
                    oth[s2].a = hit ? lg[m].a : oth[s2].a;
                    oth[s2].b = hit ? lg[m].b : oth[s2].b;
                    oth[s2].c = hit ? lg[m].c : oth[s2].c;
                    oth[s2].lognormal = hit ? lg[m].lognormal : oth[s2].lognormal;
                    da[s2] = hit ? lg[m].a - ra : da[s2];
                    nb[s2] = hit ? rbb - lg[m].b : nb[s2];
                    nc[s2] = hit ? rc - lg[m].c : nc[s2];
                }
            }
        }
#pragma unroll
        for (int sl = 0; sl < NM; ++sl) upw[sl] = go ? (sl >= (kSingle ? jsingle : jn) ? 1.0 : 0.0) : upw[sl];
        if (go) {
            j = jn;
            jm = kSingle ? jsingle : jn;
            mk.shift = lnthj;
#pragma unroll
            for (int sl = 0; sl < NM; ++sl) {
                ncu[sl] = fma(da[sl], lnthj, nc[sl]);
            }
            h0 = (thi - tlo) * (1.0 / double(kConvNInit));
            gap = 1e-7 * (thi - tlo);
#pragma unroll
            for (int sl = 0; sl < NM; ++sl) {  // the marks are walked downwards
                mk.q[sl] = mk.I[sl] + 1;
                mk.v[sl] = mk.value_d(sl);
            }
            if (KIND != KF_LONG) {
                scaleS[0] = 1.0;
                scaleS[1] = A * thj;
                scaleS[2] = A * (A + 1.0) * thj * thj;
            }
#pragma unroll
            for (int e = 0; e < 3; ++e) flo[e] = kConvFloor * scaleS[e];
            stop = tlo;
            if (PHASE != 0) {
                hlo = fmin(fmax(mk.extra[1], tlo), thi);
                hhi = fmin(fmax(mk.extra[2], tlo), thi);
            }
            cur = PHASE == 2 ? hhi : thi;
            if (PHASE == 2) stop = hlo;
            a0 = cur;
            io = kConvNInit - 1;
            L = 0;
            i = 0u;
            budget = kConvBudget;
            out[0] = out[1] = out[2] = 0.0;
        }
        if (PHASE == 2) {   // the sums of phase 1 carry over
#pragma unroll
            for (int r = rmin; r < NR; ++r)
#pragma unroll
                for (int e = 0; e < 3; ++e) out[e] = (go && jn == r) ? Traw[r][e] : out[e];
        }
        busy = need ? go : busy;
    };
    // The next initial panel [nxt, cur) BELOW the current edge for the lanes with need = true; need = false: no change.
    // Returns true for the lanes whose rule is finished: no panel left, or the rigorous bound of everything that is left,
    //   int_{t_lo}^{cur} W s^m (1 - w) G dt <= sup W x sup (1 - w) x sup (s^m G) x (cur - t_lo),
    // is below kConvTermTol x max(|accumulated|, floor x scale) for all three outputs (sup W: the weight is log-concave
    // with its mode at ln A; sup (1 - w) <= min(1, sum of the rho of the modes above j), and ln rho of a Gamma-family mode
    // with theta >= theta_j is convex in t, so its supremum over [t_lo, cur] is at an end point -- any other mode above
    // j: sup (1 - w) = 1; s^m G <= s(cur)^m (c_a s + c_b s^2) for Long).  Two exponentials per initial panel.
    const auto next_panel = [&](bool need) -> bool {
        const double ub = exp_node(cur), sb = ub * thj, lsb = cur + lnthj;   // (a bound's factors: exp_node's 4e-14 is plenty)
        const double lw = cur > tmode ? lwmode : fma(A, cur, -ub) - lgA;
        const double ow = anyln ? own(sb, lsb) : 0.0;
        double lmax = -INFINITY;
#pragma unroll
        for (int sl = 0; sl < NM; ++sl) {
            const double lr = oth[sl].lognormal ? oth[sl](sb, lsb) - ow : fma(da[sl], cur, fma(nb[sl], sb, ncu[sl]));
            const double l = fmax(lr, ltlo[sl]);
            lmax = sl >= jm ? fmax(lmax, l) : lmax;
        }
        const double lsig = convex ? fmin(0.0, lmax + lnup) : 0.0;
        double Bv = exp_node(exp_arg_clamp(lw + lsig)) * (cur - stop);
        if (KIND == KF_LONG) Bv *= fma(Q.kf[1] * sb, sb, Q.kf[2] * sb);  // G(s) <= c_a s + c_b s^2
        bool stopb = true;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            if (!(Bv <= kConvTermTol * fmax(fabs(out[e]), flo[e]))) stopb = false;
            Bv *= sb;
        }
        // PHASE 1: an edge on the upper end of the hole jumps to its lower end
        const double curj = (PHASE == 1 && cur <= hhi && cur > hlo) ? hlo : cur;
        const bool go = need && curj > stop && !stopb;
        if (PHASE == 1) mid_flag = hlo < hhi && !(stopb && cur >= hhi);
        const double lim = curj - gap;
        int io2 = io;
        double ownp = io2 > 0 ? fma(h0, double(io2), tlo) : tlo;
        while (need && io2 > 0 && ownp >= lim) {
            --io2;
            ownp = io2 > 0 ? fma(h0, double(io2), tlo) : tlo;
        }
        double nxt = fmax(stop, ownp);
        nxt = fmax(nxt, mk.prev(lim, need));
        if (PHASE == 1) nxt = hhi < curj ? fmax(nxt, hhi) : nxt;   // (a forced edge, whatever the gap rule says)
        nxt = nxt < stop + gap ? stop : nxt;
        io = io2;
        // (need && !go: the rule is over -- the caller sees it when it looks at the return value; the two call sites that take
        // a rule's FIRST panel do not, and the lane then runs one trip on a panel of width zero, which adds nothing, and is
        // told again.  Without the zero width that trip evaluated the previous rule's last panel with the new rule's
        // parameters: harmless while a rule could not end before its first panel -- a hole of phase 2 can, on the bound.)
        a0 = go ? nxt : need ? curj : a0;
        h = go ? curj - nxt : need ? 0.0 : h;
        cur = go ? nxt : cur;
        L = need ? 0 : L;
        i = need ? 0u : i;
        if (PHASE == 2) sing = need ? (go && nxt == mk.extra[1]) : sing;
        return need && !go;
    };
    next_rule(true, ConvInt<0>{});
    (void)next_panel(busy);
#pragma unroll 1
    while (busy) {
        cost += PHASE == 2 ? 0x10000 : 1;   // (panel evaluations of this lane -- phase 2's in the upper half: conv_hint_byte)
        // the panel in xi (its position inside the initial panel [a0, a0 + h]): centre (i + 1/2) 2^-L, half width 2^-(L+1)
        const double hx = ldexp(0.5, -L), cx = fma(2.0 * hx, double(i), hx);
        const double hw = h * hx, c = fma(h, cx, a0);   // half width and centre in t of a plain panel
        double K[3] = {0.0, 0.0, 0.0}, G[3] = {0.0, 0.0, 0.0};
        --budget;
        // one node: t, u = e^t, its Kronrod weight and (Gauss nodes) its Gauss weight
        // u is handed over as a FACTOR ef of the panel's ucw (the pairs: ucw = e^c, ef = e^(+-d); PHASE 2: ucw = 1, ef = u) and never
        // formed: th_j ucw and the slots' nb th_j ucw once per panel, -ucw ef inside the weight's exponent
        double ucw = 1.0, thuc = thj, nbuc[NM];
        constexpr bool kMomEf = PHASE == 0;
        const auto eval_node = [&](double t, double ef, double wk, double wg, bool gauss) {
            const double wt = exp_node(fma(-ucw, ef, fma(A, t, -lgA)));   // (the argument is inside the rule's range: > -1e3)
            const double s = thuc * ef, ls = t + lnthj;
            const double ow = anyln ? own(s, ls) : 0.0;
            double up = 0.0, den = 1.0;
#pragma unroll
            for (int sl = 0; sl < NM; ++sl) {
                const double lr = oth[sl].lognormal ? oth[sl](s, ls) - ow : fma(da[sl], t, fma(nbuc[sl], ef, ncu[sl]));
                const double rho = exp_node(exp_arg_clamp(lr));   // (finite: 0 x rho below is 0)
                den += rho;
                up = fma(upw[sl], rho, up);
            }
            double hh = wt * (up * recip_node(den));
            if (PHASE == 1)   // outside [x_t, 2 x_t]: both particles below x_t, or the larger one above
                hh *= s <= Q.kf[0] ? (Q.kf[1] * gl) * (s * s) : Q.kf[2] * s;
            if (PHASE == 2)   // inside: this rule's table (rows j * kLongNT + r of the lane's LDS column); ln(x_t / s) = ex1 - t
                hh *= LTAB ? conv_long_G_mid_tab(Q, kj, c0, rB, s, mk.extra[1] - t, gtab, j * kLongNT)
                           : conv_long_G_mid(Q, kj, lgB, rB, s);
            // (the homogeneous kernels: the moments in powers of ef, th_j ucw and its square applied to the panel's six sums)
            const double sm = kMomEf ? ef : s;
            const double v0 = hh, v1 = hh * sm, v2 = (hh * sm) * sm;
            K[0] = fma(wk, v0, K[0]);
            K[1] = fma(wk, v1, K[1]);
            K[2] = fma(wk, v2, K[2]);
            if (gauss) {
                G[0] = fma(wg, v0, G[0]);
                G[1] = fma(wg, v1, G[1]);
                G[2] = fma(wg, v2, G[2]);
            }
        };
        if (PHASE == 2) {
            // The pairs of Kronrod nodes as a ROLLED loop (their constants come through scalar loads): unrolled, the 15 copies
            // of the node with the table code left 220 registers in scratch, 221 scratch accesses per trip and 20 KB of HBM
            // traffic per parcel (VALU busy 0.35; round 5 PMC); rolled: 30 accesses per trip.  Node positions go through xi;
            // the panel with the singular edge maps t = a0 + h xi^4.
#pragma unroll
            for (int sl = 0; sl < NM; ++sl) nbuc[sl] = nb[sl] * thj;
            const auto node_at = [&](double xi, double wk, double wg) {
                const double xi2 = xi * xi;
                const double tt = sing ? xi2 * xi2 : xi, jac = sing ? 4.0 * (xi2 * xi) : 1.0;
                const double t = fma(h, tt, a0);
                eval_node(t, exp_fin(t), wk * jac, wg * jac, true);
            };
#pragma unroll 1
            for (int g = 0; g < 7; ++g) {
                const double dx = hx * kGKX[14 - g], wk = kGKWK[g], wg = kGKWG[g];
                node_at(cx - dx, wk, wg);
                node_at(cx + dx, wk, wg);
            }
            node_at(cx, kGKWK[7], kGKWG[7]);
        } else {
            // The Kronrod nodes are symmetric about the centre: e^(c +- d) = e^c e^(+-d), so a pair of nodes shares ONE exponential
            // and its reciprocal (8 exponentials + 7 reciprocals per panel for the 15 values of u instead of 15 exponentials; round
            // 4).  The pairs are taken from the outside in (g = 0 with 14, ...), the centre last.
            const double uc = exp_fin(c);
            ucw = uc;
            thuc = thj * uc;
#pragma unroll
            for (int sl = 0; sl < NM; ++sl) nbuc[sl] = nb[sl] * thuc;
#pragma unroll
            for (int g = 0; g < 7; ++g) {
                const double d = hw * kGKX[14 - g];   // > 0
                const double e = exp_small(d), re = recip_cubic(e);
                eval_node(c - d, re, kGKWK[g], kGKWG[g], (g & 1) != 0);
                eval_node(c + d, e, kGKWK[14 - g], kGKWG[14 - g], (g & 1) != 0);
            }
            eval_node(c, 1.0, kGKWK[7], kGKWG[7], true);
            if (kMomEf) {
                const double thuc2 = thuc * thuc;
                K[1] *= thuc;
                G[1] *= thuc;
                K[2] *= thuc2;
                G[2] *= thuc2;
            }
        }
        bool ok = true;
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (fabs(K[e] - G[e]) * hw > kTolT * fmax(fabs(fma(K[e], hw, out[e])), flo[e])) ok = false;
        const bool accept = ok || L == kConvLMax || budget <= 0;
        if (accept && hw > 0.0) {   // (hw = 0: the panel of width zero of a rule that ended before its first panel, see next_panel)
#pragma unroll
            for (int e = 0; e < 3; ++e) out[e] = fma(K[e], hw, out[e]);
        }
        const unsigned ni = accept ? i + 1u : i << 1;
        const int upl = accept ? min((int)__builtin_ctz(ni), L) : 0;
        i = ni >> upl;
        L = accept ? L - upl : L + 1;
        const bool done = next_panel(accept && L == 0);
        // the rule of this lane is finished: park its sums and take the next one (selects; lanes with done = false pass through).
        // The block is a no-op for a wave none of whose lanes has finished a rule in this trip -- about half of the trips (the
        // lanes of a wave finish their rules within a few trips of each other) -- so it sits behind a wave-level vote: which
        // path a wave takes depends on its other lanes, no lane's values do.
        if (__builtin_amdgcn_ballot_w64(done) != 0ull) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const bool st = done && j == r;
                Traw[r][0] = st ? out[0] : Traw[r][0];
                Traw[r][1] = st ? out[1] : Traw[r][1];
                Traw[r][2] = st ? out[2] : Traw[r][2];
                if (PHASE == 1) midneed[r] = st ? mid_flag : midneed[r];
            }
            next_rule(done, ConvInt<1>{});
            (void)next_panel(done && busy);
        }
    }
}

struct ConvSplit {
    bool valid;                 // this lane owns a parcel
    int p2_hint;                // the trips phase 2 took for this lane's parcel the last time (0: unknown)
    unsigned int *sh_cnt;       // the ranking's LDS arrays (QB counters, QB slots)
    unsigned short *sh_perm;
    int cost2;                  // out: the trips of phase 2 of THIS lane's parcel (walked by whichever lane)
};

// Phase 2 of a Long plan on the lane the second ranking picks (see ConvSplit).  In: the owner's rules, sums of phase 1 and
// hole flags; out: the owner's sums after phase 2, sp.cost2.  xt: the table rows (exchange rows before and after the walk).
template <int N, int NROW>
__device__ __forceinline__ void conv_long_phase2_split(const KArgs<N, 1> &A, const QArgs &Q, const double (&nn)[N],
                                                       const double (&th)[N], const double (&kk)[N],
                                                       const ConvRule<(N > 1 ? N - 1 : 1)> (&rb)[(N > 1 ? N - 1 : 1)],
                                                       double (&Traw)[(N > 1 ? N - 1 : 1)][3],
                                                       const bool (&midneed)[(N > 1 ? N - 1 : 1)], double (&xt)[NROW][kConvBlock],
                                                       ConvSplit &sp) {
    constexpr int NM = N > 1 ? N - 1 : 1;
    static_assert(NROW >= 3 * N + 6 * NM + 1 && NROW >= NM * kLongNT, "exchange rows");
    const int t = threadIdx.x;
    // ---- the owner leaves what phase 2 needs
    int flags = 0;
#pragma unroll
    for (int m = 0; m < N; ++m) {
        xt[3 * m + 0][t] = nn[m];
        xt[3 * m + 1][t] = th[m];
        xt[3 * m + 2][t] = kk[m];
    }
#pragma unroll
    for (int j = 0; j < NM; ++j) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            xt[3 * N + 3 * j + e][t] = rb[j].sc[e];
            xt[3 * N + 3 * NM + 3 * j + e][t] = Traw[j][e];
        }
        flags |= (sp.valid && rb[j].valid && midneed[j]) ? (1 << j) : 0;
    }
    xt[3 * N + 6 * NM][t] = double(flags);
    // ---- the second ranking: by the trips phase 2 took the last time (regime_rank brackets itself with barriers: the rows
    // above are visible once it returns)
    regime_rank<kConvBlock>(sp.valid, sp.p2_hint > kConvBlock - 2 ? kConvBlock - 2 : sp.p2_hint, sp.sh_cnt, sp.sh_perm);
    const int src = sp.sh_perm[t];
    double n2[N], th2[N], k2[N], sc2[NM][3], T2[NM][3];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        n2[m] = xt[3 * m + 0][src];
        th2[m] = xt[3 * m + 1][src];
        k2[m] = xt[3 * m + 2][src];
    }
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            sc2[j][e] = xt[3 * N + 3 * j + e][src];
            T2[j][e] = xt[3 * N + 3 * NM + 3 * j + e][src];
        }
    const int flags2 = (int)xt[3 * N + 6 * NM][src];
    __syncthreads();   // every lane has what it came for: the rows become the tables
    // ---- the rules of the parcel in hand, prepared again (the function that prepared them for phase 1)
    ConvLogDensity lg2[N];
    double cm2[N], wm2[N], lnth2[N], lgk2[N];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        ConvMode md;
        md.lognormal = A.dist_type[m] == DIST_LOGNORMAL;
        md.n = n2[m];
        md.th = th2[m];
        md.k = k2[m];
        md.lnth = md.lognormal ? th2[m] : log_pos(th2[m]);
        md.lgk = md.lognormal ? 0.0 : lgamma_pos(k2[m]);
        lnth2[m] = md.lnth;
        lgk2[m] = md.lgk;
        lg2[m] = conv_log_density(md);
        cm2[m] = conv_ln_mean(md);
        wm2[m] = conv_core_width(md);
    }
    ConvRule<NM> rb2[NM];
    bool mid2[NM];
    int cost2 = 0;
    // (no lane of the wave with a hole: nothing to prepare, no table, no walk -- the waves the second ranking fills with the
    // parcels that have none)
    if (__builtin_amdgcn_ballot_w64(flags2 != 0) != 0ull) {
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            const bool lnj = A.dist_type[j] == DIST_LOGNORMAL;
            conv_rule_prepare<N, KF_LONG>(Q, j, lnj, n2[j], k2[j], th2[j], lnth2[j], lgk2[j], cm2, wm2, lg2, sc2[j], rb2[j]);
            mid2[j] = (flags2 >> j) & 1;
            rb2[j].valid = rb2[j].valid && mid2[j];
            if (!lnj) conv_long_tab_build(k2[j], xt, j * kLongNT);
        }
        conv_T_merged<N, KF_LONG, true, 2>(Q, lg2, rb2, T2, mid2, xt, cost2);
    }
    __syncthreads();   // the tables are dead: the rows carry the sums back to the owners
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
        for (int e = 0; e < 3; ++e) xt[3 * j + e][src] = T2[j][e];
    xt[3 * NM][src] = double(cost2 >> 16);
    __syncthreads();
    int tl = threadIdx.x;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
        for (int e = 0; e < 3; ++e) Traw[j][e] = xt[3 * j + e][tl];
    sp.cost2 = (int)xt[3 * NM][tl];
}

// ---- plans of more than CLOUDY_AOT_MAX_MODES modes (round 6; VERDICT r5 missing #3) ----
// conv_coal_ints below keeps, per lane, the closed-form tables of EVERY mode across the unrolled pair loops (25 doubles each) and
// the prepared rules of every mode across the merged walk (~38 each): fine for the 2-4 modes of the reference's examples, 2 303
// spilled registers and 15-40 s of hiprtc per kernel for eight.  The reference is generic in N (Coalescence.jl:470-489), so such
// plans run -- here with loops over the modes that STAY loops: (n, theta, k) of the modes and the 3N partial tendencies wait in the
// lane's LDS column (indexed at run time), a pair's two mode tables are built when the pair is taken, and the rules are walked one
// after the other, each prepared just before its walk (conv_T_merged with NRULES = 1) -- the same functions, the same order of
// every accumulation, so the same bits as the unrolled form would give; one wave per workgroup (jit.hpp: 64 threads for such plans)
// keeps the LDS of eight workgroups on a CU.
template <int N, int KIND>
__device__ __forceinline__ void conv_coal_ints_rolled(const KArgs<N, 1> &A, const QArgs &Q, const double *__restrict__ tab,
                                                      const double (&nn)[N], const double (&th)[N], const double (&kk)[N],
                                                      double (&acc)[N][3], int &cost) {
    constexpr int NM = N - 1;
    __shared__ double sh_ntk[3 * N][kConvBlock];
    __shared__ double sh_acc[3 * N][kConvBlock];
    __shared__ double sh_none[1][kConvBlock];   // (the walk's table argument: plans of this size have no incomplete-beta tables)
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int m = 0; m < N; ++m) {
        sh_ntk[3 * m + 0][t] = nn[m];
        sh_ntk[3 * m + 1][t] = th[m];
        sh_ntk[3 * m + 2][t] = kk[m];
        sh_acc[3 * m + 0][t] = sh_acc[3 * m + 1][t] = sh_acc[3 * m + 2][t] = 0.0;
    }
    // ---- every closed form (pairs j <= k), in the order of conv_coal_ints
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
        ConvMode mj;
        conv_mode<KIND>(Q, A.dist_type[j] == DIST_LOGNORMAL, sh_ntk[3 * j][t], sh_ntk[3 * j + 1][t], sh_ntk[3 * j + 2][t], mj);
        double pr[4];
        conv_pair<KIND>(Q, tab, mj, mj, true, pr);
        sh_acc[3 * j + 0][t] -= 0.5 * pr[0];
        sh_acc[3 * j + 2][t] += pr[3];
#pragma unroll 1
        for (int k = j + 1; k < N; ++k) {
            ConvMode mk;
            conv_mode<KIND>(Q, A.dist_type[k] == DIST_LOGNORMAL, sh_ntk[3 * k][t], sh_ntk[3 * k + 1][t], sh_ntk[3 * k + 2][t], mk);
            conv_pair<KIND>(Q, tab, mj, mk, false, pr);
            sh_acc[3 * j + 0][t] -= pr[0];
            sh_acc[3 * j + 1][t] -= pr[1];
            sh_acc[3 * j + 2][t] -= pr[2];
            sh_acc[3 * k + 1][t] += pr[1];
            sh_acc[3 * k + 2][t] += fma(2.0, pr[3], pr[2]);
        }
    }
    // ---- T_m of every mode but the last, one rule at a time
    cost = 0;
#pragma unroll 1
    for (int j = 0; j < N - 1; ++j) {
        const double nj = sh_ntk[3 * j][t], thj = sh_ntk[3 * j + 1][t], kj = sh_ntk[3 * j + 2][t];
        const bool lnj = A.dist_type[j] == DIST_LOGNORMAL;   // wave-uniform
        double T0 = 0.0, T1 = 0.0, T2 = 0.0;
        if (nj > 0.0) {
            // the log densities and cores of all modes: built per rule, nothing of them is live across a walk
            ConvLogDensity lg[N];
            double cm[N], wm[N];
#pragma unroll
            for (int m = 0; m < N; ++m) {
                ConvMode md;
                md.lognormal = A.dist_type[m] == DIST_LOGNORMAL;
                md.n = sh_ntk[3 * m + 0][t];
                md.th = sh_ntk[3 * m + 1][t];
                md.k = sh_ntk[3 * m + 2][t];
                md.lnth = md.lognormal ? md.th : log_pos(md.th);
                md.lgk = md.lognormal ? 0.0 : lgamma_pos(md.k);
                lg[m] = conv_log_density(md);
                cm[m] = conv_ln_mean(md);
                wm[m] = conv_core_width(md);
            }
            ConvMode mj;
            conv_mode<KIND>(Q, lnj, nj, thj, kj, mj);
            double pr[4];
            conv_pair<KIND>(Q, tab, mj, mj, true, pr);
            if (lnj) {
                const double totals[3] = {0.5 * pr[0], pr[1], pr[2] + pr[3]};
                if (KIND == KF_CONSTANT || KIND == KF_LINEAR)
                    conv_T_lognormal_poly<N, (KIND == KF_LINEAR ? KF_LINEAR : KF_CONSTANT)>(Q, nj, thj, kj, cm, wm, lg, j, totals, T0, T1, T2, cost);
                else
                    conv_T_lognormal<N, KIND>(Q, tab, nj, thj, kj, cm, wm, lg, j, totals, T0, T1, T2);
            } else {
                const double prefj = KIND == KF_LONG ? 0.5 * (nj * nj) : 0.5 * pr[0];
                double sc[3] = {1.0, 1.0, 1.0};
                if (KIND == KF_LONG) {
                    const double rp = 1.0 / prefj;
                    sc[0] = (0.5 * pr[0]) * rp;
                    sc[1] = pr[1] * rp;
                    sc[2] = (pr[2] + pr[3]) * rp;
                }
                ConvRule<NM> rb[1];
                conv_rule_prepare<N, KIND>(Q, j, false, nj, kj, thj, mj.lnth, mj.lgk, cm, wm, lg, sc, rb[0]);
                double Traw[1][3] = {{0.0, 0.0, 0.0}};
                bool midneed[1] = {false};
                conv_T_merged<N, KIND, false, (KIND == KF_LONG ? 1 : 0), 1, 1>(Q, lg, rb, Traw, midneed, sh_none, cost, j);
                if (KIND == KF_LONG)
                    conv_T_merged<N, KIND, false, (KIND == KF_LONG ? 2 : 0), 1, 1>(Q, lg, rb, Traw, midneed, sh_none, cost, j);
                double T[3] = {Traw[0][0], Traw[0][1], Traw[0][2]};
                if (KIND != KF_LONG) {   // the mass below t_lo, as conv_coal_ints
                    const double Ash = rb[0].A, tlo = rb[0].tlo;
                    const double u_lo = exp_fin(tlo);
                    T[0] = fma(conv_one_minus_w<N>(lg, j, u_lo * thj, tlo + mj.lnth), exp_fin(fma(Ash, tlo, -lgamma_pos(Ash + 1.0))), T[0]);
                }
                T0 = T[0] * prefj;
                T1 = T[1] * prefj;
                T2 = T[2] * prefj;
            }
        }
        sh_acc[3 * j + 0][t] -= T0;
        sh_acc[3 * j + 1][t] -= T1;
        sh_acc[3 * j + 2][t] -= T2;
        sh_acc[3 * j + 3][t] += T0;
        sh_acc[3 * j + 4][t] += T1;
        sh_acc[3 * j + 5][t] += T2;
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[k][m] = sh_acc[3 * k + m][t];
}

// get_coal_ints(::NumericalCoalStyle, ...) for the parcel of this lane in converged mode: acc[k][m], normalised units,
// kernel constants INCLUDED.  tab: nq Gauss-Legendre nodes on [-1, 1], then nq weights.
// SPLIT (round 6; the Long kernel's table plans, the kernel behind cloudy_coal_rhs only): the holes of a parcel's rules (phase 2)
// are walked by ANOTHER lane of the workgroup than the one that owns the parcel.  A wave runs each of the two loops for as long as
// its lane with the most trips, and the two counts of a parcel are nearly independent: one ranking of the workgroup's parcels
// (round 5: 4 bits of each count in one key) leaves 0.74 of the lanes active (tools/long_lane_sim.py reproduces the PMC figure
// 0.73 from the oracle's counts), a ranking PER PHASE 0.83.  So the workgroup is ranked by phase 1's count when the parcels are
// loaded and a second time, by phase 2's, between the loops.  What the second lane needs of a parcel -- (n, theta, k) of its
// modes, the output scales and the sums of phase 1 of its rules, which rules have a hole: 3N + 6(N-1) + 1 doubles -- crosses in
// the LDS rows that hold the incomplete-beta tables afterwards; the rules are prepared again by the same function that prepared
// them for phase 1 (conv_rule_prepare: ~1 % of the parcel's instructions), and the three sums per rule and the trip count come
// back the same way.  Nothing of phase 1 is live in phase 2 and the reverse.  Every lane of the workgroup must call (barriers):
// lanes without a parcel come with valid = false and n = 0.

template <int N, int KIND, bool SPLIT = false>
__device__ __forceinline__ void conv_coal_ints(const KArgs<N, 1> &A, const QArgs &Q, const double *__restrict__ tab,
                                               const double (&nn)[N], const double (&th)[N], const double (&kk)[N],
                                               double (&acc)[N][3], int &cost, ConvSplit *sp = nullptr) {
    static_assert(!SPLIT || (KIND == KF_LONG && N > 1 && N <= 3), "the split walk serves the Long kernel's table plans");
    if constexpr (N > kConvUnrolledModes) {   // (plans that exist only as kernels compiled for them)
        conv_coal_ints_rolled<N, KIND>(A, Q, tab, nn, th, kk, acc, cost);
        return;
    }
    cost = 0;   // panel evaluations of the parcel's Gamma-weight rules (the only part whose length differs between parcels)
    ConvMode md[N];
    ConvLogDensity lg[N];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        conv_mode<KIND>(Q, A.dist_type[m] == DIST_LOGNORMAL, nn[m], th[m], kk[m], md[m]);
        lg[m] = conv_log_density(md[m]);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0;
    // ---- phase 1: every closed form (pairs j <= k).  Phase 2 -- the adaptive rules -- then holds only what weighting_fn and
    // the rule of mode j need (the per-mode moment tables of phase 1 are dead: ~100 registers less across the long loops).
    double selfpr[N][4], cm[N], wm[N], nj_[N], kj_[N], thj_[N], lnthj_[N], lgkj_[N];
    bool lnj_[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double pr[4];
        conv_pair<KIND>(Q, tab, md[j], md[j], true, pr);
        acc[j][0] -= 0.5 * pr[0];  // S_1 + S_2 - R_jj = (-s0 / 2, 0, sab)
        acc[j][2] += pr[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) selfpr[j][i] = pr[i];
        cm[j] = conv_ln_mean(md[j]);
        wm[j] = conv_core_width(md[j]);
        nj_[j] = md[j].n;
        kj_[j] = md[j].k;
        thj_[j] = md[j].th;
        lnthj_[j] = md[j].lnth;
        lgkj_[j] = md[j].lgk;
        lnj_[j] = md[j].lognormal;
#pragma unroll
        for (int k = j + 1; k < N; ++k) {
            conv_pair<KIND>(Q, tab, md[j], md[k], false, pr);
            acc[j][0] -= pr[0];
            acc[j][1] -= pr[1];
            acc[j][2] -= pr[2];
            acc[k][1] += pr[1];
            acc[k][2] += fma(2.0, pr[3], pr[2]);
        }
    }
    // ---- phase 2: T_m, the self collisions weighting_fn hands to mode j + 1
    if (N > 1) {
        constexpr int NM = N > 1 ? N - 1 : 1;
        // the rules of the Gamma-family modes: everything prepared here, walked in ONE loop (conv_T_merged)
        ConvRule<NM> rb[NM];
        double Traw[NM][3], prefj[NM];
        bool any_gamma = false;
#pragma unroll
        for (int j = 0; j < N - 1; ++j) {
            Traw[j][0] = Traw[j][1] = Traw[j][2] = 0.0;
            const double (&pr)[4] = selfpr[j];
            prefj[j] = lnj_[j] ? 0.0 : KIND == KF_LONG ? 0.5 * (nj_[j] * nj_[j]) : 0.5 * pr[0];
            double sc[3] = {1.0, 1.0, 1.0};
            if (KIND == KF_LONG && !lnj_[j]) {
                // (totals: T_m with 1 - w = 1 -- half the self-collision integrals of orders 0, 1, 2 -- the estimates' scale)
                const double rp = 1.0 / prefj[j];
                sc[0] = (0.5 * pr[0]) * rp;
                sc[1] = pr[1] * rp;
                sc[2] = (pr[2] + pr[3]) * rp;
            }
            conv_rule_prepare<N, KIND>(Q, j, lnj_[j], nj_[j], kj_[j], thj_[j], lnthj_[j], lgkj_[j], cm, wm, lg, sc, rb[j]);
            if (lnj_[j]) continue;  // (wave-uniform: the closure family of a mode is a plan constant) -> conv_T_lognormal below
            any_gamma = true;
        }
        if (any_gamma || SPLIT) {   // (any_gamma is wave-uniform -- a plan constant --, and true for a SPLIT plan's first mode)
            // what the walk does not need waits in the lane's own LDS slots (conflict-free, no barrier: a lane reads what it
            // wrote): the 3N partial tendencies of phase 1 and the prefactors -- registers the allocator would otherwise
            // spill around (and, at three waves per SIMD, inside) the loop: 11 x the algorithmic HBM traffic in scratch
            __shared__ double sh_park[3 * N + (N > 1 ? N - 1 : 1)][kConvBlock];
            const int t = threadIdx.x;
#pragma unroll
            for (int k = 0; k < N; ++k)
#pragma unroll
                for (int m = 0; m < 3; ++m) sh_park[3 * k + m][t] = acc[k][m];
#pragma unroll
            for (int j = 0; j < N - 1; ++j) sh_park[3 * N + j][t] = prefj[j];
            constexpr bool kTabFits = KIND == KF_LONG && N <= 3;
            constexpr int kXRows = 3 * N + 6 * NM + 1;   // SPLIT: what crosses between the two lanes of a parcel
            constexpr int kTabRows = kTabFits ? (SPLIT && kXRows > NM * kLongNT ? kXRows : NM * kLongNT) : 1;
            __shared__ double sh_gtab[kTabRows][kConvBlock];
            bool midneed[NM];
#pragma unroll
            for (int j = 0; j < NM; ++j) midneed[j] = false;
            if (KIND != KF_LONG) {
                conv_T_merged<N, KIND, false, (KIND == KF_LONG ? 1 : 0)>(Q, lg, rb, Traw, midneed, sh_gtab, cost);
            } else {
                // (the choice is wave-uniform, and a compile-time fact in a kernel compiled for the plan)
                const bool tab = kTabFits && A.kmax <= kLongTabKmax;
                conv_T_merged<N, KIND, false, (KIND == KF_LONG ? 1 : 0)>(Q, lg, rb, Traw, midneed, sh_gtab, cost);
                if constexpr (SPLIT) {
                    conv_long_phase2_split<N, kTabRows>(A, Q, nn, th, kk, rb, Traw, midneed, sh_gtab, *sp);
                } else if (tab) {
#pragma unroll
                    for (int j = 0; j < N - 1; ++j)
                        if (!lnj_[j] && kTabFits) conv_long_tab_build(kj_[j], sh_gtab, kTabFits ? j * kLongNT : 0);
                    conv_T_merged<N, KIND, kTabFits, (KIND == KF_LONG ? 2 : 0)>(Q, lg, rb, Traw, midneed, sh_gtab, cost);
                } else {
                    conv_T_merged<N, KIND, false, (KIND == KF_LONG ? 2 : 0)>(Q, lg, rb, Traw, midneed, sh_gtab, cost);
                }
            }
#pragma unroll
            for (int k = 0; k < N; ++k)
#pragma unroll
                for (int m = 0; m < 3; ++m) acc[k][m] = sh_park[3 * k + m][t];
#pragma unroll
            for (int j = 0; j < N - 1; ++j) prefj[j] = sh_park[3 * N + j][t];
        }
#pragma unroll
        for (int j = 0; j < N - 1; ++j) {
            if (!(nj_[j] > 0.0)) continue;
            const double (&pr)[4] = selfpr[j];
            double T0 = 0.0, T1 = 0.0, T2 = 0.0;
            if (lnj_[j]) {  // wave-uniform
                const double totals[3] = {0.5 * pr[0], pr[1], pr[2] + pr[3]};
                if (KIND == KF_CONSTANT || KIND == KF_LINEAR)
                    conv_T_lognormal_poly<N, (KIND == KF_LINEAR ? KF_LINEAR : KF_CONSTANT)>(Q, nj_[j], thj_[j], kj_[j], cm, wm, lg, j,
                                                                                          totals, T0, T1, T2, cost);
                else
                    conv_T_lognormal<N, KIND>(Q, tab, nj_[j], thj_[j], kj_[j], cm, wm, lg, j, totals, T0, T1, T2);
            } else {
                double T[3] = {Traw[j][0], Traw[j][1], Traw[j][2]};
                if (KIND != KF_LONG) {
                    // the mass below t_lo (1e-13 of the weight; a sizeable part of it for a shape clamped to eps)
                    const double Ash = rb[j].A, tlo = rb[j].tlo;
                    const double u_lo = exp_fin(tlo);
                    T[0] = fma(conv_one_minus_w<N>(lg, j, u_lo * thj_[j], tlo + lnthj_[j]), exp_fin(fma(Ash, tlo, -lgamma_pos(Ash + 1.0))), T[0]);
                }
                T0 = T[0] * prefj[j];
                T1 = T[1] * prefj[j];
                T2 = T[2] * prefj[j];
            }
            acc[j][0] -= T0;
            acc[j][1] -= T1;
            acc[j][2] -= T2;
            acc[j + 1][0] += T0;
            acc[j + 1][1] += T1;
            acc[j + 1][2] += T2;
        }
    }
}

// The byte a parcel leaves in the plan's hint scratch from the `cost` of conv_coal_ints (quad_kernels.hpp ranks the parcels of a
// workgroup by it).  Homogeneous kernels: the number of panel evaluations, capped.  Long kernel: phase 2 (the holes, ~2 x the cost
// per evaluation, and no evaluation at all for many parcels) decides a wave's second loop, so it is the PRIMARY key -- waves of
// parcels without a hole skip that loop -- and phase 1's count (in units of four) the secondary one: 46.3 -> 42.4 ms per 1.25e7 parcels.
template <int KIND>
__device__ __forceinline__ unsigned char conv_hint_byte(int cost) {
    if (KIND == KF_LONG) {
        // (4 + 4 bits; 3 bits of phase 2 in units of two + 5 bits of phase 1 in units of two measured 1.3 % slower)
        const int p2 = cost >> 16, p1 = (cost & 0xffff) >> 2;
        return (unsigned char)(((p2 > 15 ? 15 : p2) << 4) | (p1 > 15 ? 15 : p1));
    }
    return (unsigned char)(cost > 255 ? 255 : cost);
}

template <int N, int KIND>
__device__ __forceinline__ void conv_coal_ints(const KArgs<N, 1> &A, const QArgs &Q, const double *__restrict__ tab,
                                               const double (&nn)[N], const double (&th)[N], const double (&kk)[N],
                                               double (&acc)[N][3]) {
    int cost;
    conv_coal_ints<N, KIND>(A, Q, tab, nn, th, kk, acc, cost);
}

}  // namespace cloudy
