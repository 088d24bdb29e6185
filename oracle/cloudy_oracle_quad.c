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
