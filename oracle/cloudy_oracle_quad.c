/*
 * cloudy_oracle_quad.c -- CPU oracle of the fixed-rule NumericalCoalStyle operator.
 *
 * TEST INFRASTRUCTURE ONLY (see cloudy_oracle.h): the checker of the HIP "quadrature-kernel" plans
 * (CLOUDY_NUMERICAL_COAL in include/cloudy_hip.h).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link, load or call it.
 *
 * What it restates.  The reference evaluates the coalescence integrals of an arbitrary kernel function K(x, y)
 * with nested ADAPTIVE Gauss-Kronrod quadrature over the densities (src/Sources/Coalescence.jl:470-708,
 * quadgk(rtol = 1e-8, maxevals = 1000) two deep: ~1e5 density evaluations per right-hand side).  This file keeps
 * the structure of that code -- the Q, R, S matrices with their zero rules (:503-622), weighting_fn (:624-642),
 * the assembly (:478-487) -- and replaces every quadgk by ONE fixed rule:
 *
 *   * the inner variable of the Q and S integrals is substituted x' = x - y, which maps their triangular
 *     domain 0 < y < x onto (0, inf)^2:
 *         Q_jk^(m) = 1/2 int int (x'+y)^m K(x', y) [f_j(x') f_k(y) + f_k(x') f_j(y)] dx' dy     (:644-670)
 *         R_jk^(m) =     int int  x^m     K(x, y)   f_k(x) f_j(y) dx dy                          (:672-687)
 *         S_k^(m)  = 1/2 int int (x'+y)^m K(x', y)   f_k(x') f_k(y) {w, 1 - w}(x'+y) dx' dy     (:689-708)
 *   * every integral against a density, int g(x) f_i(x) dx, is an nq-point Gauss rule for that density:
 *       Gamma / Exponential (k = 1):  generalised Gauss-Laguerre with weight u^(k-1) e^-u in u = x / theta
 *                                     (nodes theta u_a(k), weights n W_a(k), sum_a W_a = 1),
 *       Lognormal:                    Gauss-Hermite in t = (ln x - mu) / (sqrt(2) sigma),
 *     and a double integral is the tensor product of the two rules.
 *
 * The rule integrates K(x, y) x polynomial exactly for polynomial K of degree <= 2 nq - 1 - m per variable, so for
 * the constant and linear kernels it reproduces the analytic moments to rounding; for the hydrodynamic and Long
 * kernels it is a discretisation (error ~1e-3 at nq = 10, reported by tests against the adaptive restatement in
 * oracle/numerical_adaptive.py) -- the GPU parity claim is against THIS rule, not against quadgk.
 *
 * Nodes: eigenvalues of the Jacobi matrix of the Laguerre / Hermite recurrence by Sturm bisection, polished by
 * Newton on the orthogonal polynomial; weights from the Christoffel formula.  (The HIP library computes the same
 * rule by a different route -- a Chebyshev table in k for the start values and a fixed number of Newton steps per
 * parcel -- both are checked against scipy.special.roots_genlaguerre / roots_hermite in tests/.)
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

#include "cloudy_oracle.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- Gauss rules -------------------------------------------------------------------------------------------- */

/* number of eigenvalues of the symmetric tridiagonal (d, e2 = squared off-diagonals) below x (Sturm count) */
static int co_sturm_count(int n, const double *d, const double *e2, double x) {
    int cnt = 0;
    double q = d[0] - x;
    if (q < 0) ++cnt;
    for (int i = 1; i < n; ++i) {
        if (q == 0.0) q = 1e-300;
        q = d[i] - x - e2[i - 1] / q;
        if (q < 0) ++cnt;
    }
    return cnt;
}

/* L_n^(alpha)(x) and L_{n-1}^(alpha)(x), alpha = k - 1, by the three-term recurrence
 *   (j+1) L_{j+1} = (2j + 1 + alpha - x) L_j - (j + alpha) L_{j-1}   written in k: (2j + k - x), (j - 1 + k) */
static void co_laguerre(int n, double k, double x, double *Ln, double *Lnm1) {
    double p0 = 1.0, p1 = k - x;
    if (n == 0) {
        *Ln = 1.0;
        *Lnm1 = 0.0;
        return;
    }
    for (int j = 1; j < n; ++j) {
        double p2 = ((2.0 * j + k - x) * p1 - (j - 1.0 + k) * p0) / (j + 1.0);
        p0 = p1;
        p1 = p2;
    }
    *Ln = p1;
    *Lnm1 = p0;
}

/* nq-point generalised Gauss-Laguerre rule for the NORMALISED weight u^(k-1) e^-u / Gamma(k):
 * sum_a W[a] g(u[a]) ~ E[g(U)], U ~ Gamma(shape k, scale 1); sum_a W[a] = 1.  Returns 0, -1 on bad arguments. */
int co_gauss_gamma_rule(int nq, double k, double *u, double *W) {
    if (nq < 1 || nq > CO_MAX_QUAD || !(k > 0.0)) return -1;
    double d[CO_MAX_QUAD], e2[CO_MAX_QUAD];
    for (int i = 0; i < nq; ++i) {
        d[i] = 2.0 * i + k;            /* 2i + 1 + alpha */
        e2[i] = (i + 1.0) * (i + k);   /* (i+1)(i+1+alpha) */
    }
    const double hi0 = 4.0 * nq + 2.0 * k + 8.0; /* > largest node (Gershgorin) */
    /* C = Gamma(nq + k) / (Gamma(k) nq!) = prod_j (k + j) / (j + 1) */
    double Cnk = 1.0;
    for (int j = 0; j < nq; ++j) Cnk *= (k + j) / (j + 1.0);
    for (int i = 0; i < nq; ++i) {
        double lo = 0.0, hi = hi0;
        for (int it = 0; it < 200; ++it) { /* eigenvalue i: count(x) <= i below, > i above */
            double mid = 0.5 * (lo + hi);
            if (co_sturm_count(nq, d, e2, mid) <= i)
                lo = mid;
            else
                hi = mid;
            if (hi - lo <= 1e-13 * hi || hi - lo < 1e-300) break;
        }
        double x = 0.5 * (lo + hi);
        for (int it = 0; it < 8; ++it) { /* Newton on L_nq: x L' = n L_n - (n - 1 + k) L_{n-1} */
            double Ln, Lm;
            co_laguerre(nq, k, x, &Ln, &Lm);
            double D = nq * Ln - (nq - 1.0 + k) * Lm;
            double xn = x - x * Ln / D;
            if (!(xn > 0.0)) break;
            if (fabs(xn - x) <= 4e-16 * xn) {
                x = xn;
                break;
            }
            x = xn;
        }
        double Ln, Lm;
        co_laguerre(nq, k, x, &Ln, &Lm);
        double D = nq * Ln - (nq - 1.0 + k) * Lm; /* x L_n'(x) */
        u[i] = x;
        W[i] = Cnk * x / (D * D); /* Gamma(n+k)/(Gamma(k) n!) / (x L_n'^2) */
    }
    return 0;
}

/* nq-point Gauss-Hermite rule for the normalised weight e^(-t^2) / sqrt(pi): sum_a W[a] g(t[a]), sum W = 1 */
int co_gauss_hermite_rule(int nq, double *t, double *W) {
    if (nq < 1 || nq > CO_MAX_QUAD) return -1;
    double d[CO_MAX_QUAD], e2[CO_MAX_QUAD];
    for (int i = 0; i < nq; ++i) {
        d[i] = 0.0;
        e2[i] = 0.5 * (i + 1.0);
    }
    const double bound = sqrt(2.0 * nq + 1.0) + 1.0;
    for (int i = 0; i < nq; ++i) {
        double lo = -bound, hi = bound;
        for (int it = 0; it < 200; ++it) {
            double mid = 0.5 * (lo + hi);
            if (co_sturm_count(nq, d, e2, mid) <= i)
                lo = mid;
            else
                hi = mid;
            if (hi - lo <= 1e-14) break;
        }
        double x = 0.5 * (lo + hi);
        double p1 = 0, p0 = 0;
        for (int it = 0; it < 8; ++it) { /* orthonormal Hermite recurrence, H' = sqrt(2n) h_{n-1} */
            p0 = 0.0;
            p1 = 0.7511255444649425; /* pi^(-1/4) */
            for (int j = 0; j < nq; ++j) {
                double p2 = x * sqrt(2.0 / (j + 1.0)) * p1 - sqrt((double)j / (j + 1.0)) * p0;
                p0 = p1;
                p1 = p2;
            }
            double dp = sqrt(2.0 * nq) * p0;
            double xn = x - p1 / dp;
            if (fabs(xn - x) <= 1e-16 * (1.0 + fabs(xn))) {
                x = xn;
                break;
            }
            x = xn;
        }
        p0 = 0.0;
        p1 = 0.7511255444649425;
        for (int j = 0; j < nq; ++j) {
            double p2 = x * sqrt(2.0 / (j + 1.0)) * p1 - sqrt((double)j / (j + 1.0)) * p0;
            p0 = p1;
            p1 = p2;
        }
        t[i] = x;
        W[i] = 1.0 / (nq * p0 * p0) / sqrt(M_PI); /* w_i = 1 / (n h_{n-1}(x_i)^2), normalised by sqrt(pi) */
    }
    return 0;
}

/* the rule of one distribution: sum_a w[a] g(x[a]) ~ int g(x) f(x) dx / n   (normed density) */
int co_dist_rule(const co_dist *dst, int nq, double *x, double *w) {
    switch (dst->type) {
    case CO_EXPONENTIAL:
    case CO_GAMMA: {
        const double k = dst->type == CO_GAMMA ? dst->k : 1.0;
        double u[CO_MAX_QUAD];
        if (co_gauss_gamma_rule(nq, k, u, w) < 0) return -1;
        for (int a = 0; a < nq; ++a) x[a] = dst->theta * u[a];
        return 0;
    }
    case CO_LOGNORMAL: {
        double t[CO_MAX_QUAD];
        if (co_gauss_hermite_rule(nq, t, w) < 0) return -1;
        for (int a = 0; a < nq; ++a) x[a] = exp(dst->theta + sqrt(2.0) * dst->k * t[a]);
        return 0;
    }
    default: return -1; /* Monodisperse: the reference has no normed_density_func method (MethodError) */
    }
}

/* ---- KernelFunctions.jl:94-154 as a tagged value -------------------------------------------------------------- */
double co_kernel_func_eval(const co_kernel_func *kf, double x, double y) {
    switch (kf->kind) {
    case CO_KF_CONSTANT: return co_constant_kernel(kf->p[0], x, y);
    case CO_KF_LINEAR: return co_linear_kernel(kf->p[0], x, y);
    case CO_KF_HYDRODYNAMIC: return co_hydrodynamic_kernel(kf->p[0], x, y);
    case CO_KF_LONG: return co_long_kernel(kf->p[0], kf->p[1], kf->p[2], x, y);
    default: return NAN;
    }
}

/* get_normalized_kernel_func, KernelFunctions.jl:124-154 */
int co_get_normalized_kernel_func(const co_kernel_func *kf, const double norms[2], co_kernel_func *out) {
    *out = *kf;
    switch (kf->kind) {
    case CO_KF_CONSTANT: out->p[0] = kf->p[0] * norms[0]; return 0;
    case CO_KF_LINEAR: out->p[0] = kf->p[0] * norms[0] * norms[1]; return 0;
    case CO_KF_HYDRODYNAMIC: out->p[0] = kf->p[0] * norms[0] * pow(norms[1], 4.0 / 3.0); return 0;
    case CO_KF_LONG:
        out->p[0] = kf->p[0] / norms[1];
        out->p[1] = kf->p[1] * norms[0] * (norms[1] * norms[1]);
        out->p[2] = kf->p[2] * norms[0] * norms[1];
        return 0;
    default: return -1;
    }
}

static double co_ipow(double x, int m) {
    double r = 1.0;
    for (int i = 0; i < m; ++i) r *= x;
    return r;
}

/* get_coal_ints(::NumericalCoalStyle, pdists, kernel_func), Coalescence.jl:470-489, with the fixed rule above in
 * place of every quadgk.  out: sum(NProgMoms) tendencies, mode-major; scale (optional): sum of |terms|;
 * noise (optional): the absolute rounding error this formulation itself carries in S_2 -- the reference forms
 * (1 - weighting_fn) * inner (:703-708), and where a mode barely overlaps the next one weighting_fn = 1 - 1e-9, so the
 * difference keeps 7 digits: eps * sum |inner| of the mode below.  (The HIP kernel evaluates 1 - w directly from density
 * ratios; tests allow |hip - oracle| <= tol * scale + 8 * noise.) */
int co_get_coal_ints_numerical_fixed(const co_dist *pdists, int N, const co_kernel_func *kf, int nq, double *out,
                                     double *scale, double *noise) {
    if (N < 1 || N > CO_MAX_MODES || nq < 1 || nq > CO_MAX_QUAD) return -1;
    int np[CO_MAX_MODES], np_max = 0;
    double x[CO_MAX_MODES][CO_MAX_QUAD], w[CO_MAX_MODES][CO_MAX_QUAD];
    for (int i = 0; i < N; ++i) {
        np[i] = co_nparams(pdists[i].type);
        if (np[i] > np_max) np_max = np[i];
        if (co_dist_rule(&pdists[i], nq, x[i], w[i]) < 0) return -1;
    }
    /* Q, R: [m][j][k] (0-based; the reference's element [j,k] of matrix m+1); S: [m][{0,1}][k] */
    double Q[3][CO_MAX_MODES][CO_MAX_MODES], R[3][CO_MAX_MODES][CO_MAX_MODES], S[3][2][CO_MAX_MODES];
    double Sabs[3][CO_MAX_MODES];
    memset(Sabs, 0, sizeof Sabs);
    memset(Q, 0, sizeof Q);
    memset(R, 0, sizeof R);
    memset(S, 0, sizeof S);
    for (int m = 0; m < np_max; ++m) {
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < N; ++j) {
                /* get_Q_coalescence_matrix :503-539: zero if k <= j or NProgMoms[k] <= moment_order */
                if (!(k <= j || np[k] <= m)) {
                    double t1 = 0.0, t2 = 0.0; /* the two terms of q_integrand_inner (:644-654) after x' = x - y */
                    for (int a = 0; a < nq; ++a)
                        for (int b = 0; b < nq; ++b) {
                            t1 += (pdists[j].n * w[j][a]) * (pdists[k].n * w[k][b]) * co_ipow(x[j][a] + x[k][b], m) *
                                  co_kernel_func_eval(kf, x[j][a], x[k][b]);
                            t2 += (pdists[k].n * w[k][a]) * (pdists[j].n * w[j][b]) * co_ipow(x[k][a] + x[j][b], m) *
                                  co_kernel_func_eval(kf, x[k][a], x[j][b]);
                        }
                    Q[m][j][k] = 0.5 * (t1 + t2);
                }
                /* get_R_coalescence_matrix :541-578: zero if NProgMoms[k] <= moment_order;
                 * r_integrand: x^m K(x, y) f_k(x) f_j(y) (:672-687) */
                if (!(np[k] <= m)) {
                    double r = 0.0;
                    for (int a = 0; a < nq; ++a)
                        for (int b = 0; b < nq; ++b)
                            r += (pdists[k].n * w[k][a]) * (pdists[j].n * w[j][b]) * co_ipow(x[k][a], m) *
                                 co_kernel_func_eval(kf, x[k][a], x[j][b]);
                    R[m][j][k] = r;
                }
            }
        /* get_S_coalescence_matrix :580-622 */
        for (int k = 0; k < N; ++k) {
            const int zero = (k < N - 1) ? (np[k] <= m && np[k + 1] <= m) : (np[k] <= m);
            if (zero) continue;
            double s1 = 0.0, s2 = 0.0, sabs = 0.0;
            for (int a = 0; a < nq; ++a)
                for (int b = 0; b < nq; ++b) {
                    const double xs = x[k][a] + x[k][b];
                    /* s_integrand_inner (:689-695): x^m 1/2 K(x - y, y) f_k(x - y) f_k(y) */
                    const double inner = co_ipow(xs, m) * 0.5 * co_kernel_func_eval(kf, x[k][a], x[k][b]) *
                                         (pdists[k].n * w[k][a]) * (pdists[k].n * w[k][b]);
                    const double wf = co_weighting_fn(xs, k + 1, pdists, N); /* :624-642 */
                    s1 += wf * inner;         /* s_integrand1 :697-701 */
                    s2 += (1.0 - wf) * inner; /* s_integrand2 :703-708 */
                    sabs += fabs(inner);
                }
            S[m][0][k] = s1;
            S[m][1][k] = s2;
            Sabs[m][k] = sabs;
        }
    }
    /* assembly :478-487 */
    int o = 0;
    for (int k = 0; k < N; ++k)
        for (int m = 0; m < np[k]; ++m) {
            double sq = 0.0, sr = 0.0, mag = 0.0;
            for (int j = 0; j < N; ++j) {
                sq += Q[m][j][k];
                sr += R[m][j][k];
                mag += fabs(Q[m][j][k]) + fabs(R[m][j][k]);
            }
            double v = sq - sr + S[m][0][k];
            mag += fabs(S[m][0][k]);
            if (k > 0) {
                v += S[m][1][k - 1];
                mag += fabs(S[m][1][k - 1]);
            }
            out[o] = v;
            if (scale) scale[o] = mag;
            if (noise) noise[o] = 2.220446049250313e-16 * (Sabs[m][k] + (k > 0 ? Sabs[m][k - 1] : 0.0));
            ++o;
        }
    return o;
}

/* rhs_coal!(NumericalCoalStyle(), dmom, mom, p, ...), box_model_helpers.jl:29-53 with p.kernel_func already
 * normalised (the examples pass get_normalized_kernel_func(kernel, norms), Numerical/n_particles_gamma.jl:35) */
int co_rhs_coal_numerical(const co_params *p, const co_kernel_func *kf_normalized, int nq, const double *mom,
                          double *dmom, double *scale, double *noise) {
    double mom_norms[CO_MAX_MODES * 3], mn[CO_MAX_MODES * 3], ci[CO_MAX_MODES * 3];
    co_dist pdists[CO_MAX_MODES];
    int nmom = co_get_moments_normalizing_factors(p->NProgMoms, p->N, p->norms, mom_norms);
    if (nmom < 0) return -1;
    for (int q = 0; q < nmom; ++q) mn[q] = mom[q] / mom_norms[q];
    int off = 0;
    for (int i = 0; i < p->N; ++i) {
        if (co_update_dist_from_moments(p->dist_type[i], mn + off, p->NProgMoms[i], p->k_range, &pdists[i]) < 0)
            return -1;
        off += p->NProgMoms[i];
    }
    if (co_get_coal_ints_numerical_fixed(pdists, p->N, kf_normalized, nq, ci, scale, noise) < 0) return -1;
    for (int q = 0; q < nmom; ++q) {
        dmom[q] = ci[q] * mom_norms[q];
        if (scale) scale[q] *= mom_norms[q];
        if (noise) noise[q] *= mom_norms[q];
    }
    return nmom;
}

int co_rhs_coal_numerical_batch(const co_params *p, const co_kernel_func *kf_normalized, int nq, long n_parcels, long ld,
                                const double *mom, double *dmom, double *scale, double *noise, int n_threads) {
    int nmom = 0;
    for (int i = 0; i < p->N; ++i) nmom += p->NProgMoms[i];
    int err = 0;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
    for (long i = 0; i < n_parcels; ++i) {
        double m[CO_MAX_MODES * 3], d[CO_MAX_MODES * 3], s[CO_MAX_MODES * 3], z[CO_MAX_MODES * 3];
        for (int q = 0; q < nmom; ++q) m[q] = mom[(size_t)q * ld + i];
        if (co_rhs_coal_numerical(p, kf_normalized, nq, m, d, scale ? s : NULL, noise ? z : NULL) < 0) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            err = 1;
        }
        for (int q = 0; q < nmom; ++q) {
            dmom[(size_t)q * ld + i] = d[q];
            if (scale) scale[(size_t)q * ld + i] = s[q];
            if (noise) noise[(size_t)q * ld + i] = z[q];
        }
    }
    (void)n_threads;
    return err ? -1 : nmom;
}

/* ================================================================================================================ */
/* CONVERGED mode (CLOUDY_QUAD_CONVERGED, include/cloudy_hip.h): SAME-RULE restatement of what the HIP kernels of        */
/* csrc/quad_conv.hpp evaluate -- the integrals of Coalescence.jl:503-708 split along the non-smooth sets of the kernel  */
/* function, so that the result converges to the reference's quadgk answer instead of stalling at 1e-3:                  */
/*                                                                                                                      */
/*   Every CoalescenceKernelFunction of the reference (KernelFunctions.jl:39-116) is, on each side of its non-smooth     */
/*   set, a short sum of SEPARABLE power terms c x^alpha y^beta:                                                         */
/*     constant  c;   linear  c (x + y);                                                                                 */
/*     hydrodynamic, x > y:  C (x^4/3 + 2 x y^1/3 - 2 x^1/3 y - y^4/3), C = E pi (3/4pi)^(4/3); the negative for x < y;  */
/*     Long:  c_b (x^2 + y^2) on the square x, y < x_t;  c_a (x + y) elsewhere.                                          */
/*   For Gamma / Exponential densities the integral of a power term over such a region is a product of moments times a   */
/*   probability with a closed form:                                                                                     */
/*     int int_{y < x} x^a y^b f_j(x) f_k(y) = M^j_a M^k_b I_z(k_k + b, k_j + a),  z = theta_j / (theta_j + theta_k)     */
/*     (regularised incomplete beta: P(Y' < X') for Gamma variates), and partial moments M_q P(k + q, x_t / theta) on    */
/*     the Long square.  So Q and R (:503-578) need NO quadrature at all.                                                 */
/*   The self collisions split by weighting_fn (:580-622, 624-642) are integrals over s = x' + y of a function of s       */
/*   alone: with S = X + Y ~ Gamma(2k, theta) independent of tau = Y / S ~ Beta(k, k),                                   */
/*     T_m = S_2k^(m) = 1/2 n^2 E[ S^m (1 - w(S)) G(S) ],   G(s) = E_tau[ K(s (1 - tau), s tau) ]                        */
/*   and G is a closed form too (homogeneous kernels: G = kbar s^gamma; Long: incomplete betas of argument x_t / s).     */
/*   What is left is ONE 1-D integral per mode against a Gamma density of a smooth sigmoid: a composite Gauss-Legendre   */
/*   rule in z, s / theta = ln(1 + e^z) (logarithmic near 0, linear in the tail), panels split at the kinks of G.         */
/* Lognormal modes: see below (LN).                                                                                     */
/* ================================================================================================================ */

/* ln Gamma for x > 0 */
static double co_lgam(double x) { return lgamma(x); }

/* regularised incomplete beta I_x(a, b), a, b > 0, 0 <= x <= 1: Lentz's continued fraction (DLMF 8.17.22), evaluated on
 * the side where it converges fast (x < (a + 1) / (a + b + 2), else through I_x(a, b) = 1 - I_{1-x}(b, a)) */
static double co_inc_beta_cf(double a, double b, double x) {
    const double tiny = 1e-300;
    double c = 1.0, d = 1.0 - (a + b) * x / (a + 1.0);
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 500; ++m) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((a + m2 - 1.0) * (a + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c;
        if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (a + b + m) * x / ((a + m2) * (a + m2 + 1.0));
        d = 1.0 + aa * d;
        if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c;
        if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return h;
}
/* y = 1 - x handed in by callers that know it without cancellation (x = t_j / (t_j + t_k), y = t_k / (t_j + t_k)) */
static double co_inc_beta_xy(double a, double b, double x, double y) {
    if (!(x > 0.0)) return 0.0;
    if (!(y > 0.0)) return 1.0;
    const double lbt = co_lgam(a + b) - co_lgam(a) - co_lgam(b) + a * log(x) + b * log(y);
    if (x < (a + 1.0) / (a + b + 2.0)) return exp(lbt) * co_inc_beta_cf(a, b, x) / a;
    return 1.0 - exp(lbt) * co_inc_beta_cf(b, a, y) / b;
}
double co_inc_beta(double a, double b, double x) { return co_inc_beta_xy(a, b, x, 1.0 - x); }

/* q-point Gauss-Legendre rule on [-1, 1] */
int co_gauss_legendre_rule(int q, double *x, double *w) {
    if (q < 1 || q > CO_MAX_QUAD) return -1;
    for (int i = 0; i < q; ++i) {
        double t = cos(M_PI * (i + 0.75) / (q + 0.5)), dp = 1.0;
        for (int it = 0; it < 100; ++it) {
            double p0 = 1.0, p1 = t;
            for (int j = 2; j <= q; ++j) {
                const double p2 = ((2.0 * j - 1.0) * t * p1 - (j - 1.0) * p0) / j;
                p0 = p1;
                p1 = p2;
            }
            if (q == 1) {
                p0 = 1.0;
                p1 = t;
            }
            dp = q * (t * p1 - p0) / (t * t - 1.0);
            const double dt = p1 / dp;
            t -= dt;
            if (fabs(dt) < 1e-16) break;
        }
        x[q - 1 - i] = t;
        w[q - 1 - i] = 2.0 / ((1.0 - t * t) * dp * dp);
    }
    return 0;
}

static double co_shape(const co_dist *d) { return d->type == CO_EXPONENTIAL ? 1.0 : d->k; }
/* moment of order q (real): Gamma family n theta^q Gamma(k + q) / Gamma(k); Lognormal n exp(q mu + q^2 sigma^2 / 2) */
static double co_gmom(const co_dist *d, double q) {
    if (d->type == CO_LOGNORMAL) return d->n * exp(q * d->theta + 0.5 * q * q * d->k * d->k);
    const double k = d->type == CO_EXPONENTIAL ? 1.0 : d->k;
    return d->n * exp(q * log(d->theta) + co_lgam(k + q) - co_lgam(k));
}
static double co_conv_norm_cdf(double z) { return 0.5 * erfc(-z * 0.70710678118654752440); }
/* partial moment below xt: Gamma family M_q P(k + q, xt / theta); Lognormal M_q Phi((ln xt - mu - q sigma^2) / sigma) */
static double co_pmom(const co_dist *d, double q, double xt) {
    if (d->type == CO_LOGNORMAL) return co_gmom(d, q) * co_conv_norm_cdf((log(xt) - d->theta - q * d->k * d->k) / d->k);
    const double k = d->type == CO_EXPONENTIAL ? 1.0 : d->k;
    return co_gmom(d, q) * co_gamma_inc_p(k + q, xt / d->theta);
}

static void co_conv_range_top(double A, double top, double *tlo, double *thi);
/* ---- the adaptive 1-D rule of converged mode (shared specification with csrc/quad_conv.hpp) ----
 * Variable t = ln(s / theta) in [t_lo, t_hi]: the Gamma weight u^(A-1) e^-u du is exp(A t - e^t) dt, entire in t (the
 * bisection deals with the double-exponential upper flank).  Initial panels: the union of CO_CONV_NINIT equal
 * pieces of [t_lo, t_hi] (width h0) with a list of marks -- the kinks of the integrand (Long: s = x_t, 2 x_t) and,
 * for every other mode, marks graded around its core, where weighting_fn has its transitions: ln s = c and c +- w 2^i,
 * i = 0 .. I (c = ln of the mode's mean size; w = sigma for a Lognormal mode, 1 / sqrt(max(k, 1)) for a Gamma mode; I = the
 * first doubling at which w 2^I reaches the equal pieces' own width h0, at most CO_CONV_IMAX) -- walked from t_lo upwards (next edge = the smallest mark or equal-piece edge more than `gap` = 1e-7
 * (t_hi - t_lo) beyond the current one).  So a spike of another mode, however narrow, meets panels of its own size, and
 * the transitions in its flanks lie well inside a panel, not at an edge where the nodes would miss them.  Every initial panel
 * is integrated by bisection with the Gauss-Kronrod (7, 15) pair: a panel is accepted when, for each output, |K15 - G7| <=
 * tol x max(|the integral accumulated so far, this panel included|, CO_CONV_FLOOR x the output's scale) -- the relative
 * criterion of the reference's quadgk(rtol), with the accumulated value standing in for the final one; the floor (1e-10 of
 * the value the integral would have with 1 - w = 1) only ends the refinement of integrals that are zero to that level --
 * at depth CO_CONV_LMAX, or once the rule has spent its budget of panel evaluations; the accepted value is K15.  With tol = 1e-9 the accepted K15 values are good to
 * ~1e-14 of scale (G7 is the estimate's accuracy, K15 has 1.6 x its order), so a decision that flips on a rounding
 * difference between two implementations moves the result by that much, not by tol. */
#define CO_CONV_NINIT 10
#define CO_CONV_LMAX 12
#define CO_CONV_IMAX 12
#define CO_CONV_FLOOR 1e-10
#define CO_CONV_BUDGET 8192   /* panel evaluations of one rule (Gamma weight) */
#define CO_CONV_BUDGET_LN 1024 /* ... of the outer rule of a Lognormal mode's T_m (96 q inner points per node) */
/* marks of one rule: (2 (CO_CONV_IMAX + 1) + 1) = 27 per other mode at most, + 2 kinks of the Long kernel's G; up to
 * CO_MAX_MODES - 1 = 7 other modes (round 5: plans of up to eight modes; 96 were enough for four -- a six-mode Long case of
 * tools/fuzz_parity.py --converged --big wrote past them) */
#define CO_CONV_MAX_MARKS ((CO_MAX_MODES - 1) * (2 * (CO_CONV_IMAX + 1) + 1) + 3)
static const double CO_GK_X[15] = {-0.991455371120812639206854697526329, -0.949107912342758524526189684047851,
                                   -0.864864423359769072789712788640926, -0.741531185599394439863864773280788,
                                   -0.586087235467691130294144838258730, -0.405845151377397166906606412076961,
                                   -0.207784955007898467600689403773245, 0.0,
                                   0.207784955007898467600689403773245,  0.405845151377397166906606412076961,
                                   0.586087235467691130294144838258730,  0.741531185599394439863864773280788,
                                   0.864864423359769072789712788640926,  0.949107912342758524526189684047851,
                                   0.991455371120812639206854697526329};
static const double CO_GK_WK[15] = {0.022935322010529224963732008058970, 0.063092092629978553290700663189204,
                                    0.104790010322250183839876322541518, 0.140653259715525918745189590510238,
                                    0.169004726639267902826583426598550, 0.190350578064785409913256402421014,
                                    0.204432940075298892414161999234649, 0.209482141084727828012999174891714,
                                    0.204432940075298892414161999234649, 0.190350578064785409913256402421014,
                                    0.169004726639267902826583426598550, 0.140653259715525918745189590510238,
                                    0.104790010322250183839876322541518, 0.063092092629978553290700663189204,
                                    0.022935322010529224963732008058970};
static const double CO_GK_WG[15] = {0.0, 0.129484966168869693270611432679082, 0.0, 0.279705391489276667901467771423780,
                                    0.0, 0.381830050505118944950369775488975, 0.0, 0.417959183673469387755102040816327,
                                    0.0, 0.381830050505118944950369775488975, 0.0, 0.279705391489276667901467771423780,
                                    0.0, 0.129484966168869693270611432679082, 0.0};
static int co_conv_walk_down_ = 1; /* the Gamma-weight T_m rules: 1 = descending walk with early termination (the kernels'), 0 = round 3's upward walk */
void co_conv_set_walk_down(int on) { co_conv_walk_down_ = on; }
static _Thread_local long co_conv_nodes_; /* integrand evaluations of the adaptive rules since the last reset */
static _Thread_local long co_conv_rule_nodes_[CO_MAX_MODES]; /* ... of the T_m rule of each mode in the last call */
void co_conv_rule_node_counts(long *out) {
    for (int i = 0; i < CO_MAX_MODES; ++i) out[i] = co_conv_rule_nodes_[i];
}
static _Thread_local long co_conv_phase_evals_[3]; /* panel evaluations of the descending walk by phase (0: homogeneous kernels) */
static _Thread_local long co_conv_rejects_;        /* ... of which were rejected (bisected): evaluations - rejects = accepted panels */
void co_conv_phase_evals(long *out, int reset) { /* tools/long_lane_sim.py: the per-parcel cost the device's hint bytes record */
    for (int i = 0; i < 3; ++i) {
        out[i] = co_conv_phase_evals_[i];
        if (reset) co_conv_phase_evals_[i] = 0;
    }
}
long co_conv_rejects(int reset) {
    const long n = co_conv_rejects_;
    if (reset) co_conv_rejects_ = 0;
    return n;
}
long co_conv_node_count(int reset) {
    const long n = co_conv_nodes_;
    if (reset) co_conv_nodes_ = 0;
    return n;
}
/* vals[0..nout) = the integrand at t (per unit t) */
typedef void (*co_vec_fn)(double t, double *vals, void *ctx);
#define CO_CONV_MAX_OUT 32
/* marks[0..nmarks): see above; the estimate looks at outputs est_idx[0..n_est) only (NULL: all),
 * scaleS[] = the scale of each output (see CO_CONV_FLOOR) */
static void co_conv_adaptive(double tlo, double thi, const double *marks, int nmarks, int nout, const int *est_idx,
                             int n_est, double tol, const double *scaleS, int budget, co_vec_fn f, void *ctx,
                             double *out) {
    double vals[CO_CONV_MAX_OUT], K[CO_CONV_MAX_OUT], G[CO_CONV_MAX_OUT];
    const double h0 = (thi - tlo) / CO_CONV_NINIT, gap = 1e-7 * (thi - tlo);
    double cur = tlo;
    int io = 1;
    while (cur < thi) {
        double nxt = thi, own = tlo + h0 * io;
        while (own <= cur + gap) {
            ++io;
            own = tlo + h0 * io;
        }
        if (own < nxt) nxt = own;
        for (int m = 0; m < nmarks; ++m)
            if (marks[m] > cur + gap && marks[m] < nxt) nxt = marks[m];
        if (nxt > thi - gap) nxt = thi;
        const double a0 = cur, h = nxt - cur;
        cur = nxt;
        int L = 0;
        unsigned i = 0;
        for (;;) {
            const double w = ldexp(h, -L), hw = 0.5 * w, c = (a0 + w * i) + hw;
            for (int o = 0; o < nout; ++o) K[o] = G[o] = 0.0;
            --budget;
            for (int g = 0; g < 15; ++g) {
                f(c + hw * CO_GK_X[g], vals, ctx);
                ++co_conv_nodes_;
                for (int o = 0; o < nout; ++o) {
                    K[o] += CO_GK_WK[g] * vals[o];
                    if (g & 1) G[o] += CO_GK_WG[g] * vals[o];
                }
            }
            int ok = 1;
            for (int e = 0; e < (est_idx ? n_est : nout); ++e) {
                const int o = est_idx ? est_idx[e] : e;
                /* (a NaN estimate accepts: the NaN reaches the output) */
                if (fabs(K[o] - G[o]) * hw > tol * fmax(fabs(out[o] + K[o] * hw), CO_CONV_FLOOR * scaleS[o])) ok = 0;
            }
            if (ok || L == CO_CONV_LMAX || budget <= 0) {
                for (int o = 0; o < nout; ++o) out[o] += K[o] * hw;
                ++i;
                while (L > 0 && !(i & 1u)) {
                    i >>= 1;
                    --L;
                }
                if (L == 0) break;
            } else {
                ++L;
                i <<= 1;
            }
        }
    }
}
/* ---- the Gamma-weight T_m rules walk their initial panels DOWNWARDS and stop early (round 4; csrc/quad_conv.hpp,
 * conv_T_merged).  The same initial edges -- equal pieces, graded marks, kinks -- taken from t_hi down to t_lo (next edge =
 * the largest mark or equal-piece edge more than `gap` below the current one; an edge within gap of t_lo is t_lo), the same
 * bisection and acceptance test.  1 - w -> 1 towards large sizes (the modes above j dominate there), so the integral is
 * established in the first panels and the accumulated value the acceptance test compares with is the final one almost from
 * the start: measured on the cfg4q batch, 7.6e-14 of scale against the rule at 1e-14 where the upward walk has 1e-10 at the
 * same tolerance -- which is what lets CO_CONV_TOL_DESC be 1e-7.  And before every initial panel [., b] a RIGOROUS bound of
 * everything that is left, int_{t_lo}^{b} W s^m (1 - w) G dt <= sup W x sup (1 - w) x sup(s^m G) x (b - t_lo), is compared
 * with CO_CONV_TERM_TOL x max(|accumulated|, floor x scale): below it for all outputs the rule ends.  sup W: the weight
 * exp(A t - e^t) / Gamma(A) is log-concave with its mode at ln A; sup (1 - w) <= min(1, sum_{m > j} rho_m) and ln rho_m of
 * a Gamma-family mode with theta_m >= theta_j is CONVEX in t (its e^t coefficient 1 - theta_j / theta_m is >= 0), so its
 * supremum over [t_lo, b] is at an end point; any other mode above j (a Lognormal one, or a smaller scale): sup (1 - w) = 1.
 * bound(b, B[3], ctx) returns the three bounds. */
#define CO_CONV_TERM_TOL 1e-10
typedef void (*co_bound_fn)(double b, double stop, double *B, void *ctx);
/* Round 5 -- an END-POINT SINGULARITY inside the range (the Long kernel: G(s) behaves like (s - x_t)^k just above x_t, k the
 * shape of the mode, so K15 converges only algebraically in the panel whose lower edge is x_t and rounds 3-4 ran the Long
 * rules at a hundredth of the tolerance, 43 against 30 panel evaluations per parcel): the initial panel [a0, a0 + h] whose
 * LOWER edge is `sing_edge` is integrated in xi, t = a0 + h xi^P, P = CO_CONV_SING_P = 4, dt = h P xi^(P-1) d xi -- the integrand
 * times the Jacobian behaves like xi^(P k + P - 1) there, and the bisection works on xi in [0, 1].  Every other panel has
 * t = a0 + h xi.  sing_edge = NaN: none. */
#define CO_CONV_SING_P 4
static double co_conv_long_tol_factor_ = 1.0;   /* (experiments; rounds 3-4: 0.01) */
void co_conv_set_long_tol_factor(double f) { co_conv_long_tol_factor_ = f; }
static int co_conv_sing_p_ = CO_CONV_SING_P;   /* (experiments: 1 switches the substitution off) */
void co_conv_set_sing_p(int p) { co_conv_sing_p_ = p; }
/* Round 5, the Long kernel's rules are walked in TWO PHASES (csrc/quad_conv.hpp, conv_T_merged): phase 1 goes from t_hi down
 * to t_lo and jumps over the "hole" [hlo, hhi] = [x_t, 2 x_t] (clamped to the range; both ends are forced panel edges); phase
 * 2 walks the hole from hhi down to hlo, the sums of phase 1 carried over (the acceptance test of the hole's panels compares
 * with everything outside it).  The bound that ends a rule covers [stop, b] -- stop = t_lo in phases 0 / 1, hlo in phase 2 --
 * so a phase-1 stop ABOVE the hole makes the hole negligible too: the return value says whether phase 2 is needed.
 * phase 0: the plain walk of the other kernel functions (hlo, hhi unused). */
static int co_conv_descending(double tlo, double thi, const double *marks, int nmarks, double tol, const double *scaleS,
                              int budget, co_vec_fn f, co_bound_fn bound, void *ctx, double sing_edge, int phase, double hlo,
                              double hhi, double *out) {
    double vals[3], K[3], G[3], B[3];
    const double h0 = (thi - tlo) / CO_CONV_NINIT, gap = 1e-7 * (thi - tlo);
    const double stop = phase == 2 ? hlo : tlo;
    double cur = phase == 2 ? hhi : thi;
    int io = CO_CONV_NINIT - 1, midneed = phase == 1 && hlo < hhi;
    for (;;) {
        bound(cur, stop, B, ctx);
        int stopb = 1;
        for (int o = 0; o < 3; ++o)
            if (!(B[o] <= CO_CONV_TERM_TOL * fmax(fabs(out[o]), CO_CONV_FLOOR * scaleS[o]))) stopb = 0;
        if (stopb && phase == 1 && cur >= hhi) midneed = 0;
        const double curj = (phase == 1 && cur <= hhi && cur > hlo) ? hlo : cur; /* an edge on the hole's upper end jumps */
        if (stopb || !(curj > stop)) break;
        const double lim = curj - gap;
        double own = io > 0 ? tlo + h0 * io : tlo;
        while (io > 0 && own >= lim) {
            --io;
            own = io > 0 ? tlo + h0 * io : tlo;
        }
        double nxt = fmax(stop, own);
        for (int m = 0; m < nmarks; ++m)
            if (marks[m] < lim && marks[m] > nxt) nxt = marks[m];
        if (phase == 1 && hhi < curj) nxt = fmax(nxt, hhi); /* a forced edge, whatever the gap rule says */
        if (nxt < stop + gap) nxt = stop;
        const double a0 = nxt, h = curj - nxt;
        const int sing = phase == 2 && co_conv_sing_p_ > 1 && a0 == sing_edge;
        cur = nxt;
        int L = 0;
        unsigned i = 0;
        for (;;) {
            /* the panel in xi: centre (i + 1/2) 2^-L, half width 2^-(L+1) */
            const double wx = ldexp(1.0, -L), hx = 0.5 * wx, cx = wx * i + hx;
            for (int o = 0; o < 3; ++o) K[o] = G[o] = 0.0;
            --budget;
            ++co_conv_phase_evals_[phase];
            for (int g = 0; g < 15; ++g) {
                const double xi = cx + hx * CO_GK_X[g];
                double t = a0 + h * xi, jac = 1.0;
                if (sing) {
                    const double xp1 = co_ipow(xi, co_conv_sing_p_ - 1);
                    t = a0 + h * (xp1 * xi);
                    jac = co_conv_sing_p_ * xp1;
                }
                f(t, vals, ctx);
                ++co_conv_nodes_;
                for (int o = 0; o < 3; ++o) {
                    const double v = vals[o] * jac;
                    K[o] += CO_GK_WK[g] * v;
                    if (g & 1) G[o] += CO_GK_WG[g] * v;
                }
            }
            const double hw = h * hx;
            int ok = 1;
            for (int o = 0; o < 3; ++o)
                if (fabs(K[o] - G[o]) * hw > tol * fmax(fabs(out[o] + K[o] * hw), CO_CONV_FLOOR * scaleS[o])) ok = 0;
            if (ok || L == CO_CONV_LMAX || budget <= 0) {
                for (int o = 0; o < 3; ++o) out[o] += K[o] * hw;
                ++i;
                while (L > 0 && !(i & 1u)) {
                    i >>= 1;
                    --L;
                }
                if (L == 0) break;
            } else {
                ++co_conv_rejects_;
                ++L;
                i <<= 1;
            }
        }
    }
    return midneed;
}
/* t of a size s in the variable of a rule with scale theta */
static double co_conv_t_of_s(double s, double theta) { return log(s / theta); }
/* ln of the mean size of a mode and the width of its core in ln s */
static double co_ln_mean(const co_dist *d) {
    if (d->type == CO_LOGNORMAL) return d->theta + 0.5 * d->k * d->k;
    return log(co_shape(d) * d->theta);
}
static double co_core_width(const co_dist *d) {
    if (d->type == CO_LOGNORMAL) return d->k;
    return 1.0 / sqrt(fmax(co_shape(d), 1.0));
}
/* the marks of a core (c, w) for a rule with equal pieces h0; theta > 0: a Gamma-weight rule with that scale (marks
 * converted to its t), else a rule over ln s itself */
static int co_core_marks(double c, double w, double h0, double theta, double *marks, int n) {
    const double ratio = h0 / w; /* (every rule's variable is a logarithm of s: the equal pieces are h0 wide in ln s) */
    int I = 0;
    if (ratio > 1.0) I = ratio < 4096.0 ? (int)ceil(log2(ratio)) : CO_CONV_IMAX;
    if (I > CO_CONV_IMAX) I = CO_CONV_IMAX;
    for (int q = -(I + 1); q <= I + 1 && n < CO_CONV_MAX_MARKS; ++q) {
        const double off = q == 0 ? 0.0 : (q < 0 ? -1.0 : 1.0) * ldexp(w, (q < 0 ? -q : q) - 1);
        const double ls = c + off;
        marks[n++] = theta > 0.0 ? ls - log(theta) : ls;
    }
    return n;
}
/* the Gamma(A) weight of the rule's variable at t: u, ln u and density x du/dt */
static double co_conv_weight(double t, double A, double lgA, double *u, double *lu) {
    *u = exp(t); /* du = u dt */
    *lu = t;
    return exp(A * t - *u - lgA);
}

/* P(L'_c < G'_a) for a Gamma-family mode G and a Lognormal mode L on the grid a = f + i (f = 0, 1/3; i = 0..3),
 * c = 0, 1/3, 1, 4/3, 2, 7/3, 3, 10/3:  H[f][i][ci] = E_{Gamma(k_g + a, theta_g)}[Phi((ln x - mu - c sigma^2) / sigma)] by
 * the adaptive rule; the orders i share the nodes of the base shape: E_{Gamma(A0 + i)}[g] = E_{Gamma(A0)}[u^i g] / (A0)_i.
 * The error estimate looks at the four corner outputs (i = 0, 3; c = 0, 10/3). */
typedef struct {
    double A0, lgA, lth, mu, sg;
} co_H_ctx;
static void co_H_node(double t, double *vals, void *v) {
    const co_H_ctx *h = (const co_H_ctx *)v;
    double u, lu;
    const double wt = co_conv_weight(t, h->A0, h->lgA, &u, &lu);
    const double w0 = (lu + h->lth - h->mu) / h->sg;
    double ph[8];
    for (int c = 0; c < 8; ++c) ph[c] = co_conv_norm_cdf(w0 - ((c >> 1) + ((c & 1) ? 1.0 / 3.0 : 0.0)) * h->sg);
    double wu = wt;
    for (int i = 0; i < 4; ++i) {
        for (int c = 0; c < 8; ++c) vals[8 * i + c] = wu * ph[c];
        wu *= u;
    }
}
typedef struct {
    const co_dist *g, *l; /* the pair this grid belongs to */
    double H[2][4][8];
} co_H_grid;
static void co_conv_H_grid(const co_dist *g, const co_dist *l, double tol, co_H_grid *out) {
    out->g = g;
    out->l = l;
    static const int est_idx[4] = {0, 7, 24, 31};
    for (int f = 0; f < 2; ++f) {
        const double A0 = co_shape(g) + (f ? 1.0 / 3.0 : 0.0);
        co_H_ctx h = {A0, co_lgam(A0), log(g->theta), l->theta, l->k};
        double tlo, thi, acc[32], tolS[32], marks[CO_CONV_MAX_MARKS];
        co_conv_range_top(A0, 4.0, &tlo, &thi);
        int nm = 0;
        nm = co_core_marks(l->theta + (5.0 / 3.0) * l->k * l->k, l->k, (thi - tlo) / CO_CONV_NINIT, g->theta, marks, nm); /* where the Phi's turn */
        double poch = 1.0;
        for (int i = 0; i < 4; ++i) {
            for (int c = 0; c < 8; ++c) {
                acc[8 * i + c] = 0.0;
                tolS[8 * i + c] = poch;
            }
            poch *= A0 + i;
        }
        co_conv_adaptive(tlo, thi, marks, nm, 32, est_idx, 4, tol, tolS, CO_CONV_BUDGET, co_H_node, &h, acc);
        poch = 1.0;
        for (int i = 0; i < 4; ++i) {
            for (int c = 0; c < 8; ++c) out->H[f][i][c] = acc[8 * i + c] / poch;
            poch *= A0 + i;
        }
    }
}
/* look-up on the grid: a = f + i, c = cf + ci */
static double co_conv_H(const co_H_grid *grid, double a, double c) {
    const int i = (int)floor(a + 1e-9), ci = (int)floor(c + 1e-9);
    const int f = (a - i) < 0.1 ? 0 : 1, cf = (c - ci) < 0.1 ? 0 : 1;
    return grid->H[f][i][2 * ci + cf];
}
/* P(Y' < X'), X' / Y' the size-biased laws of orders a / b of modes j / k */
static double co_conv_prob_less(const co_dist *dj, double a, const co_dist *dk, double b, const co_H_grid *hg) {
    const int lj = dj->type == CO_LOGNORMAL, lk = dk->type == CO_LOGNORMAL;
    if (!lj && !lk) {
        const double z = dj->theta / (dj->theta + dk->theta), omz = dk->theta / (dj->theta + dk->theta);
        return co_inc_beta_xy(co_shape(dk) + b, co_shape(dj) + a, z, omz);
    }
    if (lj && lk) /* ln X' - ln Y' ~ N(mu_j + a s_j^2 - mu_k - b s_k^2, s_j^2 + s_k^2) */
        return co_conv_norm_cdf((dj->theta + a * dj->k * dj->k - dk->theta - b * dk->k * dk->k) /
                                sqrt(dj->k * dj->k + dk->k * dk->k));
    if (!lj) return co_conv_H(hg, a, b);      /* j Gamma, k Lognormal: the grid of (G = j, L = k) */
    return 1.0 - co_conv_H(hg, b, a);         /* j Lognormal, k Gamma: 1 - P(X' < Y'), the grid of (G = k, L = j) */
}

/* int int x^p y^q K(x, y) f_j(x) f_k(y) dx dy over (0, inf)^2 (closed forms; a Gamma-Lognormal pair of the hydrodynamic
 * kernel needs the 1-D rule for P(Y' < X')) */
static double co_conv_pair(const co_kernel_func *kf, const co_dist *dj, const co_dist *dk, int p, int q, double *mag,
                           const co_H_grid *hg) {
    double dummy;
    if (!mag) mag = &dummy;
    switch (kf->kind) {
    case CO_KF_CONSTANT: return *mag = kf->p[0] * co_gmom(dj, p) * co_gmom(dk, q);
    case CO_KF_LINEAR:
        return *mag = kf->p[0] * (co_gmom(dj, p + 1.0) * co_gmom(dk, q) + co_gmom(dj, p) * co_gmom(dk, q + 1.0));
    case CO_KF_HYDRODYNAMIC: {
        static const double term[4][3] = {{4.0 / 3.0, 0.0, 1.0}, {1.0, 1.0 / 3.0, 2.0}, {1.0 / 3.0, 1.0, -2.0}, {0.0, 4.0 / 3.0, -1.0}};
        double tot = 0.0, m = 0.0;
        for (int t = 0; t < 4; ++t) {
            const double al = term[t][0], be = term[t][1];
            const double I = co_conv_prob_less(dj, p + al, dk, q + be, hg); /* P(Y' < X') */
            const double mm = term[t][2] * co_gmom(dj, p + al) * co_gmom(dk, q + be);
            tot += mm * (2.0 * I - 1.0);
            m += fabs(mm);
        }
        *mag = kf->p[0] * 0.46526286817455001 * m;
        return kf->p[0] * 0.46526286817455001 * tot; /* pi (3 / (4 pi))^(4/3) */
    }
    case CO_KF_LONG: {
        const double xt = kf->p[0], cb = kf->p[1], ca = kf->p[2];
#define PMJ(r) co_pmom(dj, (r), xt)
#define PMK(r) co_pmom(dk, (r), xt)
        const double full = ca * (co_gmom(dj, p + 1.0) * co_gmom(dk, q) + co_gmom(dj, p) * co_gmom(dk, q + 1.0));
        const double below2 = cb * (PMJ(p + 2.0) * PMK(q) + PMJ(p) * PMK(q + 2.0));
        const double below1 = ca * (PMJ(p + 1.0) * PMK(q) + PMJ(p) * PMK(q + 1.0));
#undef PMJ
#undef PMK
        *mag = full + below2 + below1;
        return full + below2 - below1;
    }
    default: return NAN;
    }
}

/* ln of the normed density of a Gamma-family or Lognormal mode at s (ls = ln s), ParticleDistributions.jl:363-388 */
static double co_ln_normed(const co_dist *d, double s, double ls) {
    if (d->type == CO_LOGNORMAL) {
        const double u = (ls - d->theta) / d->k;
        return -0.5 * u * u - ls - log(d->k * 2.5066282746310002);
    }
    const double k = co_shape(d);
    return (k - 1.0) * ls - s / d->theta - (co_lgam(k) + k * log(d->theta));
}
/* 1 - weighting_fn(s, j + 1) (0-based j) formed from density ratios against mode j's own density (no 0/0, full
 * relative precision where w = 1 - 1e-9) */
static double co_one_minus_w(const co_dist *pdists, int N, int j, double s, double ls) {
    const double own = co_ln_normed(&pdists[j], s, ls);
    double up = 0.0, den = 1.0;
    for (int m = 0; m < N; ++m) {
        if (m == j) continue;
        const double rho = exp(fmin(co_ln_normed(&pdists[m], s, ls) - own, 700.0));
        den += rho;
        if (m > j) up += rho;
    }
    return up / den;
}


/* E_tau[K(s (1 - tau), s tau)], tau ~ Beta(k, k), for the Long kernel */
static double co_long_G(const co_kernel_func *kf, double k, double s) {
    const double xt = kf->p[0], cb = kf->p[1], ca = kf->p[2];
    if (s <= xt) return cb * s * s * (k + 1.0) / (2.0 * k + 1.0);
    if (s >= 2.0 * xt) return ca * s;
    const double x = xt / s;
    const double P0 = 2.0 * co_inc_beta(k, k, x) - 1.0;
    const double r = exp(2.0 * co_lgam(k + 1.0) - co_lgam(2.0 * k + 2.0) - 2.0 * co_lgam(k) + co_lgam(2.0 * k)); /* B(k+1,k+1)/B(k,k) */
    const double P1 = r * (2.0 * co_inc_beta(k + 1.0, k + 1.0, x) - 1.0);
    return ca * s + (cb * s * s - ca * s) * P0 - 2.0 * cb * s * s * P1;
}

/* top: the highest power of s the integrand multiplies the Gamma(A) weight with (2 for the T_m rule) */
static void co_conv_range_top(double A, double top, double *tlo, double *thi) {
    const double Am = A + top;
    *tlo = fmax(-690.0, fmin(-1.0, (log(1e-13) + co_lgam(A + 1.0)) / A)); /* (-690: shapes clamped to k = eps) */
    *thi = log(Am + sqrt(60.0 * Am) + 30.0); /* (t = ln u) */
}
void co_conv_range(double A, double *tlo, double *thi) { co_conv_range_top(A, 2.0, tlo, thi); }

typedef struct {
    const co_dist *pdists;
    int N, j;
    const co_kernel_func *kf;
    double k, A, lgA, theta, lth, tlo;
} co_T_ctx;
static void co_T_node(double t, double *vals, void *v) {
    const co_T_ctx *c = (const co_T_ctx *)v;
    double u, lu;
    const double wt = co_conv_weight(t, c->A, c->lgA, &u, &lu);
    const double s = u * c->theta, ls = lu + c->lth;
    double h = wt * co_one_minus_w(c->pdists, c->N, c->j, s, ls);
    if (c->kf->kind == CO_KF_LONG) h *= co_long_G(c->kf, c->k, s);
    vals[0] = h;
    vals[1] = h * s;
    vals[2] = h * s * s;
}
/* the bound of what is left below b for the rule of co_T_ctx (see co_conv_descending) */
static void co_T_bound(double b, double stop, double *B, void *v) {
    const co_T_ctx *c = (const co_T_ctx *)v;
    const co_dist *dj = &c->pdists[c->j];
    const double ub = exp(b), sb = ub * c->theta, lsb = b + c->lth;
    const double tmode = log(c->A);
    const double lw = b > tmode ? c->A * tmode - c->A - c->lgA : c->A * b - ub - c->lgA;
    const double slo = exp(c->tlo) * c->theta, lslo = c->tlo + c->lth;
    double lsig = 0.0;
    int convex = 1, n_up = 0;
    double lmax = -INFINITY;
    for (int m = c->j + 1; m < c->N; ++m) {
        const co_dist *dm = &c->pdists[m];
        ++n_up;
        if (dm->type == CO_LOGNORMAL || !(1.0 / dm->theta <= 1.0 / dj->theta)) convex = 0;
        const double own_b = co_ln_normed(dj, sb, lsb), own_lo = co_ln_normed(dj, slo, lslo);
        lmax = fmax(lmax, fmax(co_ln_normed(dm, sb, lsb) - own_b, co_ln_normed(dm, slo, lslo) - own_lo));
    }
    if (convex && n_up > 0) lsig = fmin(0.0, lmax + log((double)n_up));
    double Bv = exp(lw + lsig) * (b - stop);
    if (c->kf->kind == CO_KF_LONG) Bv *= c->kf->p[2] * sb + c->kf->p[1] * sb * sb; /* G(s) <= c_a s + c_b s^2 */
    B[0] = Bv;
    B[1] = Bv * sb;
    B[2] = Bv * sb * sb;
}
/* ---- Lognormal modes (LN).  Moments and partial moments are closed forms (co_gmom, co_pmom); P(Y' < X') of the
 * hydrodynamic terms is Phi(.) for a Lognormal pair and the adaptive rule for a Gamma-Lognormal pair (co_conv_H_grid).
 * The sum of two Lognormal variates has no closed law, so T_m of a Lognormal mode keeps two variables,
 *   s = x + y and t = ln(x / y):   f(x) f(y) dx dy = n^2 g(ln x) g(ln y) d(ln s) dt,   g = the normal density of ln x,
 *   T_m = 1/2 n^2 int d(ln s) s^m (1 - w(s)) G2(ln s),   G2 = 2 int_0^inf dt K(x, y) g(ln x) g(ln y),
 *   ln x = ln s - ln(1 + e^-t),  ln y = ln s - ln(1 + e^t)
 * with weighting_fn outside the inner integral (it depends on s alone) and the kink of the hydrodynamic kernel on the
 * boundary t = 0.  Outer: the adaptive rule over ln s in [mu - 8.5 sigma, mu + 8.5 sigma + (gamma + 2) sigma^2 + ln 2]
 * (budget CO_CONV_BUDGET_LN), marks at the other modes' cores (and ln x_t, ln 2 x_t for Long); inner: CO_LN_PAN2
 * panels of q Gauss-Legendre points over t in [0, max(ln s - mu, 0) + 12 sigma], split at the Long kernel's jump (co_TL_node). */
#define CO_LN_PAN2 6   /* (round 6: the range is at most 2 sqrt(84) sigma = 18.3 sigma wide now -- six panels of <= 3 sigma cover it; 12 before the bound) */
/* ... at least; as many as make a panel no wider than CO_LN_PAN2_SIGMAS sigma (the integrand in t is a Gaussian sqrt(2) sigma
 * to 2 sigma wide around its peak), up to CO_LN_PAN2_MAX: 12 for sigma >= 0.03, 256 below 0.001 (round 4; the fixed 12 were
 * off by 2e-10 at sigma = 0.02, 6e-7 at 0.01 and 1e-4 ... 1e-1 below 0.005) */
#define CO_LN_PAN2_SIGMAS 3.0
#define CO_LN_PAN2_MAX 256
static double co_ln_pan2_sigmas_ = CO_LN_PAN2_SIGMAS;
static int co_ln_pan2_max_ = CO_LN_PAN2_MAX;
void co_conv_set_ln_inner(double sigmas, int max_panels) { /* tests: a finer inner rule as the reference of the default one */
    co_ln_pan2_sigmas_ = sigmas;
    co_ln_pan2_max_ = max_panels;
}
static int co_ln_pan2(double Tm, double sg) {
    double np = ceil(Tm / (co_ln_pan2_sigmas_ * sg));
    if (!(np >= (double)CO_LN_PAN2)) np = (double)CO_LN_PAN2;
    if (np > (double)co_ln_pan2_max_) np = (double)co_ln_pan2_max_;
    return (int)np;
}
typedef struct {
    const co_dist *pdists;
    int N, j, q;
    const co_kernel_func *kf;
    const double *xg, *wg;
} co_TL_ctx;
/* Round 5 -- polynomial kernels (constant, linear: K = c s^gamma leaves the inner integral): the inner integral is the density
 * of S = X + Y in ln s,  G2 = c s^gamma / (pi sigma^2) int_0^inf exp(-[(m - q(t))^2 + t^2 / 4] / sigma^2) dt,  m = ln s - mu,
 * q(t) = ln(2 cosh(t / 2)) -- even, analytic, Gaussian decay -- by the TRAPEZOIDAL rule with step h = min(sigma, 1/2) over
 * [0, T], T = min(max(m, 0) + 12 sigma, 2 sqrt(d^2 + CUT sigma^2)), d = m - ln 2; zero where min(d^2, 2 d - 1) > CUT sigma^2
 * (csrc/quad_conv.hpp, conv_T_lognormal_poly: the same rule, <= 4e-13 of the density's peak against mpmath except around
 * sigma = 1/2, where it reaches 3e-10). */
#define CO_LN_CUT 32.0   /* (round 6, late: 42 before -- e^-32 = 1.3e-14 of the peak, still below the trapezoidal rule's own error) */
/* The end of the inner range (round 6): E(t) = [(d - qd(t))^2 + t^2 / 4] / sigma^2, qd = q - ln 2 = ln cosh(t / 2), d = m - ln 2.
 * E(0) = d^2 / sigma^2 bounds the minimum from above.  ln cosh x >= sqrt(1 + x^2) - 1 for every x (equal at 0; the derivative of
 * the difference is tanh x - x / sqrt(1 + x^2) >= 0, which is sinh^2 x >= x^2), so with w = sqrt(1 + t^2 / 4): wherever
 * w - 1 >= d, sigma^2 E >= (w - 1 - d)^2 + w^2 - 1, and that exceeds d^2 + CUT sigma^2 for every
 * w > max(1 + max(d, 0), [(1 + d) + sqrt((d - 1)^2 + 2 CUT sigma^2)] / 2) -- there the integrand is below e^-CUT of its maximum.
 * At d = 0, sigma = ln 2: T = 6.3 against 2 sqrt(d^2 + CUT sigma^2) = 7.8 (the bound that ignores the first square; the true
 * end is 4.9), for d < 0 about half of it; sigma << 1: the same to a few per cent. */
static double co_ln_inner_top(double m, double d, double sg) {
    const double wr = 0.5 * ((1.0 + d) + sqrt((d - 1.0) * (d - 1.0) + 2.0 * CO_LN_CUT * (sg * sg)));
    const double w = fmax(wr, 1.0 + fmax(d, 0.0));
    const double T = fmin(fmax(m, 0.0) + 12.0 * sg, 2.0 * sqrt(d * d + CO_LN_CUT * (sg * sg)));
    return fmin(T, 2.0 * sqrt(w * w - 1.0));
}
static void co_TL_node_poly(double ls, double *vals, void *v) {
    const co_TL_ctx *c = (const co_TL_ctx *)v;
    const double mu = c->pdists[c->j].theta, sg = c->pdists[c->j].k, c1 = 1.0 / (sg * sg), h = fmin(sg, 0.5);
    const double ln2 = 0.6931471805599453, m = ls - mu, d = m - ln2, s = exp(ls);
    const double lb = (d <= 1.0 ? d * d : 2.0 * d - 1.0) * c1;
    double sum = 0.0;
    if (lb <= CO_LN_CUT) {
        const double Tm = co_ln_inner_top(m, d, sg);
        const int npt = (int)ceil(Tm / h);
        sum = 0.5 * exp(-(d * d) * c1);
        for (int i = 1; i <= npt; ++i) {
            const double t = h * i, qd = 0.5 * t + log1p(exp(-t)) - ln2, dq = d - qd;
            sum += exp(-(dq * dq + 0.25 * t * t) * c1);
        }
    }
    double val = co_one_minus_w(c->pdists, c->N, c->j, s, ls) * (c->kf->p[0] * c1 / M_PI * h * sum);
    if (c->kf->kind == CO_KF_LINEAR) val *= s;
    vals[0] = val;
    vals[1] = val * s;
    vals[2] = val * s * s;
}
static void co_TL_node(double ls, double *vals, void *v) {
    const co_TL_ctx *c = (const co_TL_ctx *)v;
    if (c->kf->kind == CO_KF_CONSTANT || c->kf->kind == CO_KF_LINEAR) {
        co_TL_node_poly(ls, vals, v);
        return;
    }
    const double mu = c->pdists[c->j].theta, sg = c->pdists[c->j].k, c2 = 1.0 / (2.0 * sg * sg), nrm = c2 / M_PI;
    const double s = exp(ls);
    /* Round 6: the Gaussian factor of the inner integrand is the polynomial kernels' (co_TL_node_poly) whatever the kernel
     * function -- exp(-[(m - q(t))^2 + t^2 / 4] / sigma^2) -- so the same two bounds hold: nothing where
     * min(d^2, 2 d - 1) > CUT sigma^2 (d = m - ln 2: the density of the sum is below e^-CUT of its peak), and nothing beyond
     * T = 2 sqrt(d^2 + CUT sigma^2).  Without them a shape clamped to sigma = eps ran the full 256 panels at every node -- and the
     * 63 other parcels of its wave waited: 2e5 parcel-RHS/s where the polynomial kernels make 3.5e7. */
    const double md = ls - mu, dd = md - 0.6931471805599453;
    if ((dd <= 1.0 ? dd * dd : 2.0 * dd - 1.0) > CO_LN_CUT * (sg * sg)) {
        vals[0] = vals[1] = vals[2] = 0.0;
        return;
    }
    const double Tm = co_ln_inner_top(md, dd, sg);
    /* the Long kernel jumps where the larger particle x = s / (1 + e^-t) crosses x_t: at t_b = ln(x_t / (s - x_t)) for
     * x_t < s < 2 x_t (below, both stay under x_t; above, x >= s / 2 >= x_t) -- the inner panels are split there */
    double tb = 0.0;
    int n1 = 0;
    const int np2 = co_ln_pan2(Tm, sg);
    if (c->kf->kind == CO_KF_LONG && s > c->kf->p[0] && s < 2.0 * c->kf->p[0]) {
        tb = log(c->kf->p[0] / (s - c->kf->p[0]));
        if (tb > 0.0 && tb < Tm) {
            n1 = (int)(np2 * (tb / Tm) + 0.5);
            n1 = n1 < 1 ? 1 : n1 > np2 - 1 ? np2 - 1 : n1;
        }
    }
    double G2 = 0.0;
    for (int seg = 0; seg < 2; ++seg) {
        const int np = seg == 0 ? n1 : np2 - n1;
        const double a = seg == 0 ? 0.0 : (n1 ? tb : 0.0), b = seg == 0 ? tb : Tm;
        if (np == 0) continue;
        const double h2 = (b - a) / np;
        for (int i2 = 0; i2 < np; ++i2)
            for (int g2 = 0; g2 < c->q; ++g2) {
                const double t = a + h2 * (i2 + 0.5) + 0.5 * h2 * c->xg[g2];
                const double spm = log1p(exp(-t)); /* softplus(-t); softplus(t) = t + softplus(-t) */
                const double lx = ls - spm, ly = ls - t - spm;
                const double dx = lx - mu, dy = ly - mu;
                G2 += (0.5 * h2 * c->wg[g2]) * co_kernel_func_eval(c->kf, exp(lx), exp(ly)) * exp(-(dx * dx + dy * dy) * c2);
            }
    }
    const double val = co_one_minus_w(c->pdists, c->N, c->j, s, ls) * (2.0 * nrm * G2);
    vals[0] = val;
    vals[1] = val * s;
    vals[2] = val * s * s;
}
/* totals[m]: the value of T_m with 1 - w = 1 (closed form), the scale of the error estimate */
static void co_conv_T_lognormal(const co_dist *pdists, int N, int j, const co_kernel_func *kf, int q, const double *xg,
                                const double *wg, double tol, const double totals[3], double T[3]) {
    static const double gtop[4] = {0.0, 1.0, 4.0 / 3.0, 2.0};
    const double mu = pdists[j].theta, sg = pdists[j].k;
    const double L0 = mu - 8.5 * sg, L1 = mu + 8.5 * sg + (gtop[kf->kind] + 2.0) * sg * sg + 0.6931471805599453;
    const double pref = 0.5 * pdists[j].n * pdists[j].n;
    co_TL_ctx c = {pdists, N, j, q, kf, xg, wg};
    /* marks in ln s (no junction): the other modes' cores and the Long kernel's x_t, 2 x_t */
    double marks[CO_CONV_MAX_MARKS], tolS[3];
    int nm = 0;
    for (int m = 0; m < N; ++m)
        if (m != j) nm = co_core_marks(co_ln_mean(&pdists[m]), co_core_width(&pdists[m]), (L1 - L0) / CO_CONV_NINIT, 0.0, marks, nm);
    if (kf->kind == CO_KF_LONG) {
        marks[nm++] = log(kf->p[0]);
        marks[nm++] = log(2.0 * kf->p[0]);
    }
    for (int m = 0; m < 3; ++m) {
        T[m] = 0.0;
        tolS[m] = totals[m] / pref;
    }
    co_conv_adaptive(L0, L1, marks, nm, 3, NULL, 0, tol, tolS, CO_CONV_BUDGET_LN, co_TL_node, &c, T);
    for (int m = 0; m < 3; ++m) T[m] *= pref;
}

/* get_coal_ints(::NumericalCoalStyle, ...) in converged mode; Gamma / Exponential / Lognormal modes.  q = points per
 * panel of the inner rule of a Lognormal mode's T_m (the only fixed rule left), tol = the acceptance tolerance of the
 * adaptive rules (CO_CONV_TOL = 1e-9 in the HIP kernels).  out / scale as co_get_coal_ints_numerical_fixed. */
int co_get_coal_ints_numerical_converged(const co_dist *pdists, int N, const co_kernel_func *kf, int q, double tol,
                                         double *out, double *scale) {
    if (N < 1 || N > CO_MAX_MODES || q < 1 || q > CO_MAX_QUAD || !(tol > 0.0)) return -1;
    int np[CO_MAX_MODES];
    for (int i = 0; i < N; ++i) {
        np[i] = co_nparams(pdists[i].type);
        if (pdists[i].type != CO_GAMMA && pdists[i].type != CO_EXPONENTIAL && pdists[i].type != CO_LOGNORMAL) return -2;
    }
    double xg[CO_MAX_QUAD], wg[CO_MAX_QUAD];
    co_gauss_legendre_rule(q, xg, wg);
    double acc[CO_MAX_MODES][3], mag[CO_MAX_MODES][3];
    memset(acc, 0, sizeof acc);
    memset(mag, 0, sizeof mag);
#define ADD(k_, m_, v_)            \
    do {                           \
        acc[k_][m_] += (v_);       \
        mag[k_][m_] += fabs(v_);   \
    } while (0)
#define ADDM(k_, m_, v_, mg_) \
    do {                      \
        acc[k_][m_] += (v_);  \
        mag[k_][m_] += (mg_); \
    } while (0)
    static const double gam_of_kind[4] = {0.0, 1.0, 4.0 / 3.0, 0.0};
    for (int j = 0; j < N; ++j) {
        const co_dist *dj = &pdists[j];
        const double kj = co_shape(dj);
        /* self collisions: S_1 + S_2 - R_jj = (-s0 / 2, 0, sab); T_m moves to the next mode */
        double m0, m1, m2, m3;
        const double s0 = co_conv_pair(kf, dj, dj, 0, 0, &m0, NULL), sab = co_conv_pair(kf, dj, dj, 1, 1, &m1, NULL);
        ADDM(j, 0, -0.5 * s0, 0.5 * m0);
        ADDM(j, 2, sab, m1);
        co_conv_rule_nodes_[j] = 0;
        if (j < N - 1 && dj->n > 0.0) {
            const long nodes_before = co_conv_nodes_;
            double T[3] = {0.0, 0.0, 0.0};
            if (dj->type == CO_LOGNORMAL) {
                const double totals[3] = {0.5 * s0, co_conv_pair(kf, dj, dj, 1, 0, NULL, NULL),
                                          co_conv_pair(kf, dj, dj, 2, 0, NULL, NULL) + sab};
                co_conv_T_lognormal(pdists, N, j, kf, q, xg, wg, tol, totals, T);
            } else {
                /* homogeneous kernels: T_m = 1/2 s0 E_{Gamma(2k + gamma)}[s^m (1 - w)] (the kernel's homogeneity absorbed
                 * in the weight); Long: T_m = 1/2 n^2 E_{Gamma(2k)}[s^m (1 - w) G(s)] */
                const int lng = kf->kind == CO_KF_LONG;
                const double A = 2.0 * kj + gam_of_kind[kf->kind], pref = lng ? 0.5 * dj->n * dj->n : 0.5 * s0;
                co_T_ctx c = {pdists, N, j, kf, kj, A, co_lgam(A), dj->theta, log(dj->theta), 0.0};
                double tlo, thi, marks[CO_CONV_MAX_MARKS], tolS[3];
                int nm = 0;
                co_conv_range_top(A, 2.0, &tlo, &thi);
                c.tlo = tlo;
                for (int m = 0; m < N; ++m) /* the other modes' cores */
                    if (m != j)
                        nm = co_core_marks(co_ln_mean(&pdists[m]), co_core_width(&pdists[m]), (thi - tlo) / CO_CONV_NINIT, dj->theta, marks, nm);
                if (lng) { /* kinks of G */
                    marks[nm++] = co_conv_t_of_s(kf->p[0], dj->theta);
                    marks[nm++] = co_conv_t_of_s(2.0 * kf->p[0], dj->theta);
                }
                if (lng) {
                    const double totals[3] = {0.5 * s0, co_conv_pair(kf, dj, dj, 1, 0, NULL, NULL),
                                              co_conv_pair(kf, dj, dj, 2, 0, NULL, NULL) + sab};
                    for (int m = 0; m < 3; ++m) tolS[m] = totals[m] / pref;
                } else {
                    tolS[0] = 1.0;
                    tolS[1] = A * dj->theta;
                    tolS[2] = A * (A + 1.0) * dj->theta * dj->theta;
                }
                /* (the Long kernel's G(s) is only finitely smooth at s = x_t and 2 x_t -- the Beta(k, k) law of tau ends like
                 * tau^(k-1) there -- and K15 converges slowly in the panels next to them: measured 1e-9 of scale at tol =
                 * 1e-8 on random mixtures, against 1e-10 ... 1e-13 for the homogeneous kernels; its rules run at tol / 10) */
                const double tol_T = lng ? co_conv_long_tol_factor_ * tol : tol;
                if (co_conv_walk_down_ && lng) {
                    const double ex1 = marks[nm - 2], ex2 = marks[nm - 1];
                    const double hlo = fmin(fmax(ex1, tlo), thi), hhi = fmin(fmax(ex2, tlo), thi);
                    if (co_conv_descending(tlo, thi, marks, nm, tol_T, tolS, CO_CONV_BUDGET, co_T_node, co_T_bound, &c, ex1, 1, hlo, hhi, T))
                        co_conv_descending(tlo, thi, marks, nm, tol_T, tolS, CO_CONV_BUDGET, co_T_node, co_T_bound, &c, ex1, 2, hlo, hhi, T);
                } else if (co_conv_walk_down_)
                    co_conv_descending(tlo, thi, marks, nm, tol_T, tolS, CO_CONV_BUDGET, co_T_node, co_T_bound, &c, NAN, 0, 0.0, 0.0, T);
                else
                    co_conv_adaptive(tlo, thi, marks, nm, 3, NULL, 0, tol_T, tolS, CO_CONV_BUDGET, co_T_node, &c, T);
                if (!lng) { /* the mass below t_lo (1e-13 of the weight; a sizeable part of it for a shape clamped to eps) */
                    const double u_lo = exp(tlo);
                    T[0] += co_one_minus_w(pdists, N, j, u_lo * dj->theta, tlo + c.lth) * exp(A * tlo - co_lgam(A + 1.0));
                }
                for (int m = 0; m < 3; ++m) T[m] *= pref;
            }
            for (int m = 0; m < 3; ++m) {
                ADD(j, m, -T[m]);
                ADD(j + 1, m, T[m]);
            }
            co_conv_rule_nodes_[j] = co_conv_nodes_ - nodes_before;
        }
        for (int k = j + 1; k < N; ++k) {
            const co_dist *dk = &pdists[k];
            co_H_grid hg;
            const int mixed = (dj->type == CO_LOGNORMAL) != (dk->type == CO_LOGNORMAL);
            if (mixed && kf->kind == CO_KF_HYDRODYNAMIC)
                co_conv_H_grid(dj->type == CO_LOGNORMAL ? dk : dj, dj->type == CO_LOGNORMAL ? dj : dk, tol, &hg);
            const double p0 = co_conv_pair(kf, dj, dk, 0, 0, &m0, &hg), sa = co_conv_pair(kf, dj, dk, 1, 0, &m1, &hg),
                         saa = co_conv_pair(kf, dj, dk, 2, 0, &m2, &hg), sab2 = co_conv_pair(kf, dj, dk, 1, 1, &m3, &hg);
            ADDM(j, 0, -p0, m0);
            ADDM(j, 1, -sa, m1);
            ADDM(j, 2, -saa, m2);
            ADDM(k, 1, sa, m1);
            ADDM(k, 2, saa + 2.0 * sab2, m2 + 2.0 * m3);
        }
    }
#undef ADDM
#undef ADD
    int o = 0;
    for (int k = 0; k < N; ++k)
        for (int m = 0; m < np[k]; ++m) {
            out[o] = acc[k][m];
            if (scale) scale[o] = mag[k][m];
            ++o;
        }
    return o;
}

/* rhs_coal!(NumericalCoalStyle(), ...) in converged mode, one parcel / a batch (as co_rhs_coal_numerical[_batch]) */
int co_rhs_coal_numerical_converged(const co_params *p, const co_kernel_func *kf_normalized, int q, double tol,
                                    const double *mom, double *dmom, double *scale) {
    double mom_norms[CO_MAX_MODES * 3], mn[CO_MAX_MODES * 3], ci[CO_MAX_MODES * 3];
    co_dist pdists[CO_MAX_MODES];
    int nmom = co_get_moments_normalizing_factors(p->NProgMoms, p->N, p->norms, mom_norms);
    if (nmom < 0) return -1;
    for (int i = 0; i < nmom; ++i) mn[i] = mom[i] / mom_norms[i];
    int off = 0;
    for (int i = 0; i < p->N; ++i) {
        if (co_update_dist_from_moments(p->dist_type[i], mn + off, p->NProgMoms[i], p->k_range, &pdists[i]) < 0)
            return -1;
        off += p->NProgMoms[i];
    }
    if (co_get_coal_ints_numerical_converged(pdists, p->N, kf_normalized, q, tol, ci, scale) < 0) return -1;
    for (int i = 0; i < nmom; ++i) {
        dmom[i] = ci[i] * mom_norms[i];
        if (scale) scale[i] *= mom_norms[i];
    }
    return nmom;
}

int co_rhs_coal_numerical_converged_batch(const co_params *p, const co_kernel_func *kf_normalized, int q, double tol,
                                          long n_parcels, long ld, const double *mom, double *dmom, double *scale,
                                          int n_threads) {
    int nmom = 0;
    for (int i = 0; i < p->N; ++i) nmom += p->NProgMoms[i];
    int err = 0;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads)
#endif
    for (long i = 0; i < n_parcels; ++i) {
        double m[CO_MAX_MODES * 3], d[CO_MAX_MODES * 3], s[CO_MAX_MODES * 3];
        for (int k = 0; k < nmom; ++k) m[k] = mom[(size_t)k * ld + i];
        if (co_rhs_coal_numerical_converged(p, kf_normalized, q, tol, m, d, scale ? s : NULL) < 0) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            err = 1;
        }
        for (int k = 0; k < nmom; ++k) {
            dmom[(size_t)k * ld + i] = d[k];
            if (scale) scale[(size_t)k * ld + i] = s[k];
        }
    }
    (void)n_threads;
    return err ? -1 : nmom;
}
