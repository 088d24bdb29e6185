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
// of argument x_t / s between x_t and 2 x_t).  What is left is ONE 1-D integral per mode of a smooth sigmoid against a Gamma
// density: a composite Gauss-Legendre rule in z, s / theta = ln(1 + e^z) (logarithmic near 0, linear in the tail), with
// the panels split at the kinks of G.  Every lane does the same number of nodes; the incomplete-beta continued fractions
// are the only data-dependent trip counts.
//
// Measured against nested adaptive quadrature of the reference integrals (tests/golden/numerical_adaptive.json):
// <= 1e-11 of scale, where the 10-point rule has 1e-3 ... 1e-2 (tests/test_numerical_oracle.py prints the table).
// Same-rule CPU restatement: oracle/cloudy_oracle_quad.c (co_get_coal_ints_numerical_converged).
#pragma once
#include "kernels.hpp"
#include "quad.hpp"

namespace cloudy {

constexpr int kConvPanels = 48;  // panels of the 1-D rule (QArgs::nq Gauss-Legendre points each)

// continued fraction of the incomplete beta function (DLMF 8.17.22, modified Lentz), converging for x < (a+1)/(a+b+2)
__device__ __forceinline__ double inc_beta_cf(double a, double b, double x) {
    const double tiny = 1e-300;
    double c = 1.0, d = 1.0 - (a + b) * x / (a + 1.0);
    d = fabs(d) < tiny ? tiny : d;
    d = 1.0 / d;
    double h = d;
#pragma unroll 1
    for (int m = 1; m <= 300; ++m) {
        const double md = double(m), m2 = 2.0 * md;
        double aa = md * (b - md) * x / ((a + m2 - 1.0) * (a + m2));
        d = fma(aa, d, 1.0);
        d = fabs(d) < tiny ? tiny : d;
        c = 1.0 + aa / c;
        c = fabs(c) < tiny ? tiny : c;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + md) * (a + b + md) * x / ((a + m2) * (a + m2 + 1.0));
        d = fma(aa, d, 1.0);
        d = fabs(d) < tiny ? tiny : d;
        c = 1.0 + aa / c;
        c = fabs(c) < tiny ? tiny : c;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return h;
}

// I_x(a, b) given D = x^a (1-x)^b / B(a, b); omx = 1 - x (passed in: the callers know it without cancellation)
__device__ __forceinline__ double inc_beta_from_D(double a, double b, double x, double omx, double D) {
    if (x < (a + 1.0) / (a + b + 2.0)) return D * inc_beta_cf(a, b, x) / a;
    return 1.0 - D * inc_beta_cf(b, a, omx) / b;
}

// per-mode quantities of the closed forms
struct ConvMode {
    double n, th, k, lnth, lgk;
    double Mi[5];  // M_0 .. M_4
    double Mt[4];  // M_{1/3}, M_{4/3}, M_{7/3}, M_{10/3}   (hydrodynamic)
    double pm[5];  // partial moments below the Long threshold, M_q P(k + q, x_t / theta)
};

template <int KIND>
__device__ __forceinline__ void conv_mode(const QArgs &Q, double n, double th, double k, ConvMode &m) {
    m.n = n;
    m.th = th;
    m.k = k;
    m.lnth = log_pos(th);
    m.lgk = lgamma_pos(k);
    m.Mi[0] = n;
#pragma unroll
    for (int q = 1; q < 5; ++q) m.Mi[q] = m.Mi[q - 1] * (th * (k + double(q - 1)));
    if (KIND == KF_HYDRODYNAMIC) {
        m.Mt[0] = n * exp_fin(fma(1.0 / 3.0, m.lnth, log_gamma_ratio(k, 1.0 / 3.0)));
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

// The four pair integrals  int int x^p y^q K(x, y) f_j(x) f_k(y)  for (p, q) = (0,0), (1,0), (2,0), (1,1)  ->  s0, sa, saa, sab
template <int KIND>
__device__ __forceinline__ void conv_pair(const QArgs &Q, const ConvMode &J, const ConvMode &K, bool self, double (&out)[4]) {
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
        // hydrodynamic: sum_t c_t M^j_{p + al_t} M^k_{q + be_t} (2 I_z(k_k + q + be_t, k_j + p + al_t) - 1).  The 16
        // incomplete betas lie on two grids of unit spacing, G1[i][i'] = I_z(k_k + i, k_j + 1/3 + i') and
        // G2[i][i'] = I_z(k_k + 1/3 + i, k_j + i'): one continued fraction each, the rest by the recurrences
        //   I(a, b+1) = I(a, b) + D(a, b) / b,   I(a+1, b) = I(a, b) - D(a, b) / a,   D = z^a (1-z)^b / B(a, b),
        //   D(a, b+1) = D(a, b) (1-z) (a+b) / b,   D(a+1, b) = D(a, b) z (a+b) / a
        // (absolute accuracy is what 2 I - 1 needs; the recurrences keep it).
        const double z = self ? 0.5 : J.th / (J.th + K.th), omz = self ? 0.5 : K.th / (J.th + K.th);
        const double lz = log_pos(z), lomz = log_pos(omz);
        double G[2][3][4];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const double a0 = K.k + (g ? 1.0 / 3.0 : 0.0), b0 = J.k + (g ? 0.0 : 1.0 / 3.0);
            const double lgb = g ? J.lgk : lgamma_pos(b0), lga = g ? lgamma_pos(a0) : K.lgk;
            double D0 = exp_fin(fma(a0, lz, b0 * lomz) + (lgamma_pos(a0 + b0) - lga - lgb));
            double I0 = inc_beta_from_D(a0, b0, z, omz, D0);
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {  // walk along b at i = 0, then down the column in a
                const double b = b0 + double(ib);
                double I = I0, D = D0, a = a0;
#pragma unroll
                for (int ia = 0; ia < 3; ++ia) {
                    G[g][ia][ib] = I;
                    I -= D / a;
                    D *= z * (a + b) / a;
                    a += 1.0;
                }
                I0 += D0 / b;
                D0 *= omz * (a0 + b) / b;
            }
        }
        const double C = Q.kf[0] * 0.46526286817455001;  // pi (3 / (4 pi))^(4/3)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = PP[i], q = QQ[i];
            double v = J.Mt[p + 1] * K.Mi[q] * fma(2.0, G[0][q][p + 1], -1.0);            // x^(4/3)
            v = fma(2.0 * (J.Mi[p + 1] * K.Mt[q]), fma(2.0, G[1][q][p + 1], -1.0), v);    // 2 x y^(1/3)
            v = fma(-2.0 * (J.Mt[p] * K.Mi[q + 1]), fma(2.0, G[0][q + 1][p], -1.0), v);   // -2 x^(1/3) y
            v = fma(-(J.Mi[p] * K.Mt[q + 1]), fma(2.0, G[1][q + 1][p], -1.0), v);         // -y^(4/3)
            out[i] = C * v;
        }
    }
}

// ln of the normed density at s (ls = ln s) minus a constant per mode, Gamma family: (k - 1) ls - s / theta - c,
// c = ln Gamma(k) + k ln theta
struct ConvLogDensity {
    double a, b, c;
};

// E_tau[K(s (1 - tau), s tau)], tau ~ Beta(k, k), Long kernel, for x_t < s < 2 x_t
__device__ __forceinline__ double conv_long_G_mid(const QArgs &Q, double k, double lgB, double rB, double s) {
    const double xt = Q.kf[0], cb = Q.kf[1], ca = Q.kf[2];
    const double x = xt / s, omx = 1.0 - x;  // x in (1/2, 1)
    // I_x(k, k) = 1 - I_{1-x}(k, k); I_x(k+1, k+1) by two steps of the recurrences
    const double D = exp_fin(fma(k, log_pos(x) + log_pos(omx), lgB));  // x^k (1-x)^k / B(k, k)
    const double Ikk = 1.0 - D * inc_beta_cf(k, k, omx) / k;
    const double Ik1k = Ikk - D / k;                     // I_x(k+1, k)
    const double Dk1k = D * x * 2.0;                     // D(k+1, k) = D z (2k) / k
    const double Ik1k1 = Ik1k + Dk1k / k;                // I_x(k+1, k+1)
    const double P0 = fma(2.0, Ikk, -1.0), P1 = rB * fma(2.0, Ik1k1, -1.0);
    const double s2 = s * s;
    return fma(ca, s, fma(fma(cb, s2, -(ca * s)), P0, -2.0 * cb * s2 * P1));
}

// get_coal_ints(::NumericalCoalStyle, ...) for the parcel of this lane in converged mode: acc[k][m], normalised units,
// kernel constants INCLUDED.  tab: nq Gauss-Legendre nodes on [-1, 1], then nq weights.
template <int N, int KIND>
__device__ __forceinline__ void conv_coal_ints(const KArgs<N, 1> &A, const QArgs &Q, const double *__restrict__ tab,
                                               const double (&nn)[N], const double (&th)[N], const double (&kk)[N],
                                               double (&acc)[N][3]) {
    const int nq = Q.nq;
    ConvMode md[N];
#pragma unroll
    for (int m = 0; m < N; ++m) conv_mode<KIND>(Q, nn[m], th[m], kk[m], md[m]);
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0;
    ConvLogDensity lg[N];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        lg[m].a = md[m].k - 1.0;
        lg[m].b = 1.0 / md[m].th;
        lg[m].c = fma(md[m].k, md[m].lnth, md[m].lgk);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double pr[4];
        conv_pair<KIND>(Q, md[j], md[j], true, pr);
        acc[j][0] -= 0.5 * pr[0];  // S_1 + S_2 - R_jj = (-s0 / 2, 0, sab)
        acc[j][2] += pr[3];
        if (j < N - 1 && md[j].n > 0.0) {
            // ---- T_m: the self collisions weighting_fn hands to mode j + 1
            constexpr double gam = KIND == KF_LINEAR ? 1.0 : KIND == KF_HYDRODYNAMIC ? 4.0 / 3.0 : 0.0;
            const double Ash = fma(2.0, md[j].k, gam);  // homogeneous kernels: T_m = 1/2 s0 E_{Gamma(2k + gamma)}[s^m (1 - w)]
            const double pref = KIND == KF_LONG ? 0.5 * (md[j].n * md[j].n) : 0.5 * pr[0];
            const double lgA = lgamma_pos(Ash), Am1 = Ash - 1.0;
            const double zlo = fmin(-1.0, (-29.933606208922594 + (lgA + log_pos(Ash))) / Ash);  // ln 1e-13
            const double zhi = (Ash + 2.0) + sqrt(60.0 * (Ash + 2.0)) + 30.0;
            // segments: Long splits at s = x_t and 2 x_t (kinks of G); panels in proportion to the lengths
            double e1 = zlo, e2 = zlo;
            if (KIND == KF_LONG) {
                const double u1 = Q.kf[0] / md[j].th, u2 = 2.0 * u1;
                const double z1 = u1 > 30.0 ? u1 : log(expm1(u1)), z2 = u2 > 30.0 ? u2 : log(expm1(u2));
                e1 = fmin(fmax(z1, zlo), zhi);
                e2 = fmin(fmax(z2, e1), zhi);
            }
            const double total = zhi - zlo;
            double lgB = 0.0, rB = 0.0;
            if (KIND == KF_LONG) {
                lgB = lgamma_pos(2.0 * md[j].k) - 2.0 * md[j].lgk;                      // -ln B(k, k)
                rB = md[j].k / (2.0 * fma(2.0, md[j].k, 1.0));                          // B(k+1, k+1) / B(k, k)
            }
            ConvLogDensity dl[N];
#pragma unroll
            for (int m = 0; m < N; ++m) {
                dl[m].a = lg[m].a - lg[j].a;
                dl[m].b = lg[m].b - lg[j].b;
                dl[m].c = lg[m].c - lg[j].c;
            }
            double T0 = 0.0, T1 = 0.0, T2 = 0.0;
#pragma unroll 1
            for (int sg = 0; sg < (KIND == KF_LONG ? 3 : 1); ++sg) {
                const double a = KIND == KF_LONG ? (sg == 0 ? zlo : sg == 1 ? e1 : e2) : zlo;
                const double b = KIND == KF_LONG ? (sg == 0 ? e1 : sg == 1 ? e2 : zhi) : zhi;
                const double len = b - a;
                if (!(len > 0.0)) continue;
                int np = KIND == KF_LONG ? (int)ceil(double(kConvPanels) * (len / total) - 1e-9) : kConvPanels;
                np = np < 1 ? 1 : np;
                const double h = len / double(np);
#pragma unroll 1
                for (int ip = 0; ip < np; ++ip) {
                    const double zc = fma(h, double(ip) + 0.5, a);
#pragma unroll 1
                    for (int g = 0; g < nq; ++g) {
                        const double z = fma(0.5 * h, tab[g], zc);
                        const double ez = exp_fin(-fabs(z));                  // softplus and logistic from one exponential
                        const double u = fmax(z, 0.0) + log1p(ez);
                        const double sig = (z >= 0.0 ? 1.0 : ez) / (1.0 + ez);
                        const double lu = log_pos(u);
                        const double wt = (0.5 * h * tab[nq + g]) * sig * exp_fin(fma(Am1, lu, -u) - lgA);
                        const double s = u * md[j].th, ls = lu + md[j].lnth;
                        double up = 0.0, den = 1.0;
#pragma unroll
                        for (int m = 0; m < N; ++m) {
                            if (m == j) continue;
                            const double rho = exp_fin(fmin(fma(dl[m].a, ls, fma(-dl[m].b, s, -dl[m].c)), 700.0));
                            den += rho;
                            if (m > j) up += rho;
                        }
                        double hh = wt * (up / den);
                        if (KIND == KF_LONG) {
                            const double xt = Q.kf[0];
                            double G;
                            if (sg == 0)
                                G = Q.kf[1] * (s * s) * ((md[j].k + 1.0) / fma(2.0, md[j].k, 1.0));
                            else if (sg == 2)
                                G = Q.kf[2] * s;
                            else
                                G = conv_long_G_mid(Q, md[j].k, lgB, rB, fmin(fmax(s, xt * (1.0 + 1e-15)), 2.0 * xt * (1.0 - 1e-15)));
                            hh *= G;
                        }
                        T0 += hh;
                        T1 = fma(hh, s, T1);
                        T2 = fma(hh * s, s, T2);
                    }
                }
            }
            T0 *= pref;
            T1 *= pref;
            T2 *= pref;
            acc[j][0] -= T0;
            acc[j][1] -= T1;
            acc[j][2] -= T2;
            acc[j + 1][0] += T0;
            acc[j + 1][1] += T1;
            acc[j + 1][2] += T2;
        }
#pragma unroll
        for (int k = j + 1; k < N; ++k) {
            conv_pair<KIND>(Q, md[j], md[k], false, pr);
            acc[j][0] -= pr[0];
            acc[j][1] -= pr[1];
            acc[j][2] -= pr[2];
            acc[k][1] += pr[1];
            acc[k][2] += fma(2.0, pr[3], pr[2]);
        }
    }
}

}  // namespace cloudy
