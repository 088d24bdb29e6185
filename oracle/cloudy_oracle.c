/*
 * cloudy_oracle.c -- CPU restatement of the Cloudy.jl coalescence moment RHS (see cloudy_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: parity oracle + bench.py cpu_baseline ("port").  Never linked into,
 * loaded by, or called from the product path (cloudy.jl_amd/).
 *
 * Reference paths are relative to the Cloudy.jl checkout (v0.6.0).  The restatement keeps the
 * reference's order of floating-point operations (left-to-right tuple sums, the same operand
 * grouping in products) so that differences to the Julia path are confined to the last ulp of
 * libm / special-function calls.
 */
#define _GNU_SOURCE
#include "cloudy_oracle.h"

#include <float.h>
#include <math.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CO_EPS DBL_EPSILON /* eps(Float64) */

/* ------------------------------------------------------------------------------------------ */
/* SpecialFunctions.jl (un-vendored dependency, Project.toml compat "2.5")                    */
/* ------------------------------------------------------------------------------------------ */

/* SpecialFunctions.gamma(::Float64): the Gamma function.  Call sites: ParticleDistributions.jl:180,188 */
double co_gamma(double x) { return tgamma(x); }

/* x^a e^-x / Gamma(a), the common prefactor of the series and the continued fraction */
static double co_gamma_prefactor(double a, double x) {
    if (a < 140.0) {
        double v = pow(x, a) * exp(-x) / tgamma(a);
        if (isfinite(v) && v > 0.0) return v;
    }
    return exp(a * log(x) - x - lgamma(a));
}

/* SpecialFunctions.gamma_inc(a, x)[1]: regularised lower incomplete gamma function
 * P(a,x) = 1/Gamma(a) * int_0^x t^(a-1) e^-t dt   (DLMF 8.2.4).
 * Restated from the published definition: power series DLMF 8.11.4 for x < a+1, Legendre's
 * continued fraction DLMF 8.9.2 (modified Lentz) for Q = 1-P otherwise.
 * Call sites: ParticleDistributions.jl:231,241,577,602. */
double co_gamma_inc_p(double a, double x) {
    if (!(a > 0.0) || isnan(x)) return NAN;
    if (x <= 0.0) return 0.0;
    if (isinf(x)) return 1.0;
    if (x < a + 1.0) {
        double ap = a, del = 1.0 / a, sum = del;
        for (int n = 0; n < 100000; ++n) {
            ap += 1.0;
            del *= x / ap;
            sum += del;
            if (fabs(del) < fabs(sum) * 1e-17) break;
        }
        double r = sum * co_gamma_prefactor(a, x);
        return r > 1.0 ? 1.0 : r;
    } else {
        const double tiny = 1e-300;
        double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
        for (int i = 1; i < 100000; ++i) {
            double an = -(double)i * ((double)i - a);
            b += 2.0;
            d = an * d + b;
            if (fabs(d) < tiny) d = tiny;
            c = b + an / c;
            if (fabs(c) < tiny) c = tiny;
            d = 1.0 / d;
            double del = d * c;
            h *= del;
            if (fabs(del - 1.0) < 1e-17) break;
        }
        double q = h * co_gamma_prefactor(a, x);
        return 1.0 - q;
    }
}

static double co_gamma_inc_q(double a, double x) {
    if (x <= 0.0) return 1.0;
    if (x < a + 1.0) return 1.0 - co_gamma_inc_p(a, x);
    const double tiny = 1e-300;
    double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
    for (int i = 1; i < 100000; ++i) {
        double an = -(double)i * ((double)i - a);
        b += 2.0;
        d = an * d + b;
        if (fabs(d) < tiny) d = tiny;
        c = b + an / c;
        if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 1e-17) break;
    }
    return h * co_gamma_prefactor(a, x);
}

/* SpecialFunctions.gamma_inc_inv(a, p, q): x such that P(a,x) = p, Q(a,x) = q.
 * Restated from the published definition as a safeguarded Halley iteration on whichever of
 * P-p / Q-q is the smaller tail (start: Wilson-Hilferty for a > 1, small-x inversion otherwise).
 * Call site: ParticleDistributions.jl:760. */
double co_gamma_inc_inv(double a, double p, double q) {
    if (!(a > 0.0) || isnan(p) || isnan(q)) return NAN;
    if (p <= 0.0) return 0.0;
    if (q <= 0.0) return INFINITY;
    double x;
    const double a1 = a - 1.0;
    if (a > 1.0) {
        double pp = (p < 0.5) ? p : q;
        double t = sqrt(-2.0 * log(pp));
        double z = (2.30753 + t * 0.27061) / (1.0 + t * (0.99229 + t * 0.04481)) - t;
        if (p < 0.5) z = -z;
        double w = 1.0 - 1.0 / (9.0 * a) - z / (3.0 * sqrt(a));
        x = a * w * w * w;
        if (x < 1e-3) x = 1e-3;
    } else {
        double t = 1.0 - a * (0.253 + a * 0.12);
        if (p < t)
            x = pow(p / t, 1.0 / a);
        else
            x = 1.0 - log(1.0 - (p - t) / (1.0 - t));
    }
    double lo = 0.0, hi = INFINITY;
    const double lga = lgamma(a);
    for (int it = 0; it < 200; ++it) {
        if (!(x > 0.0)) x = (hi < INFINITY) ? 0.5 * (lo + hi) : DBL_MIN;
        double err; /* P(a,x) - p, evaluated on the smaller tail */
        if (p <= 0.5)
            err = co_gamma_inc_p(a, x) - p;
        else
            err = q - co_gamma_inc_q(a, x);
        if (err > 0.0) {
            if (x < hi) hi = x;
        } else if (err < 0.0) {
            if (x > lo) lo = x;
        } else {
            return x;
        }
        double dens = exp(-x + a1 * log(x) - lga); /* dP/dx */
        double xn;
        if (dens > 0.0 && isfinite(dens)) {
            double u = err / dens;
            double corr = u / (1.0 - 0.5 * fmin(1.0, u * (a1 / x - 1.0)));
            xn = x - corr;
        } else {
            xn = -1.0;
        }
        if (!(xn > lo) || !(xn < hi)) { /* leave the bracket -> bisect (or expand) */
            if (hi < INFINITY)
                xn = 0.5 * (lo + hi);
            else
                xn = 2.0 * x;
        }
        if (fabs(xn - x) <= 4.0 * CO_EPS * fabs(xn)) return xn;
        x = xn;
    }
    return x;
}

/* ------------------------------------------------------------------------------------------ */
/* src/helper_functions.jl                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* helper_functions.jl:13-20  (1-based i, m; returns <0 where the reference throws) */
int co_get_dist_moment_ind(const int *NProgMoms, int N, int i, int m) {
    if (i < 1 || i > N) return -1;
    if (!(0 < m && m <= NProgMoms[i - 1])) return -1;
    int s = 0;
    for (int ii = 0; ii < i - 1; ++ii) s += NProgMoms[ii];
    return s + m;
}

/* helper_functions.jl:29-32 */
int co_get_dist_moments_ind_range(const int *NProgMoms, int N, int i, int *first, int *last) {
    if (i < 1 || i > N) return -1;
    int s = 0;
    for (int ii = 0; ii < i - 1; ++ii) s += NProgMoms[ii];
    *first = s + 1;
    *last = s + NProgMoms[i - 1];
    return 0;
}

/* Julia x^n for Int n >= 0 (power by squaring; n <= 2 here is 1, x, x*x exactly as Julia) */
static double co_powi(double x, int n) {
    double r = 1.0;
    for (int i = 0; i < n; ++i) r *= x;
    return r;
}

/* helper_functions.jl:40-53: norms[1] * norms[2]^(j-1), mode-major flattened */
int co_get_moments_normalizing_factors(const int *NProgMoms, int N, const double norms[2], double *out) {
    if (norms[0] <= 0 || norms[1] <= 0) return -1;
    int idx = 0;
    for (int i = 0; i < N; ++i)
        for (int j = 1; j <= NProgMoms[i]; ++j) out[idx++] = norms[0] * co_powi(norms[1], j - 1);
    return idx;
}

/* ------------------------------------------------------------------------------------------ */
/* src/Kernels/KernelTensors.jl, KernelFunctions.jl                                            */
/* ------------------------------------------------------------------------------------------ */

/* KernelTensors.jl:157-171 (exact != comparison) */
int co_check_symmetry(const double *c, int P) {
    for (int i = 0; i < P; ++i)
        for (int j = i + 1; j < P; ++j)
            if (c[i * P + j] != c[j * P + i]) return -1;
    return 0;
}

/* KernelTensors.jl:189-199: c[i,j] * (norms[1] * norms[2]^(FT(i+j-2))), 1-based i,j */
void co_get_normalized_kernel_tensor(const double *c, int P, const double norms[2], double *out) {
    for (int i = 1; i <= P; ++i)
        for (int j = 1; j <= P; ++j)
            out[(i - 1) * P + (j - 1)] = c[(i - 1) * P + (j - 1)] * (norms[0] * pow(norms[1], (double)(i + j - 2)));
}

/* KernelFunctions.jl:94-96 */
double co_constant_kernel(double rate, double x, double y) {
    (void)x;
    (void)y;
    return rate;
}
/* KernelFunctions.jl:98-100 */
double co_linear_kernel(double rate, double x, double y) { return rate * (x + y); }
/* KernelFunctions.jl:102-108 */
double co_hydrodynamic_kernel(double coal_eff, double x, double y) {
    double r1 = pow(3.0 / 4.0 / M_PI * x, 1.0 / 3.0);
    double r2 = pow(3.0 / 4.0 / M_PI * y, 1.0 / 3.0);
    double A1 = M_PI * (r1 * r1);
    double A2 = M_PI * (r2 * r2);
    return coal_eff * ((r1 + r2) * (r1 + r2)) * fabs(A1 - A2);
}
/* KernelFunctions.jl:110-116 */
double co_long_kernel(double x_thr, double below, double above, double x, double y) {
    if (x < x_thr && y < x_thr) return below * (x * x + y * y);
    return above * (x + y);
}

/* ------------------------------------------------------------------------------------------ */
/* src/ParticleDistributions/ParticleDistributions.jl                                          */
/* ------------------------------------------------------------------------------------------ */

/* ParticleDistributions.jl:425-427 */
int co_nparams(int dist_type) {
    switch (dist_type) {
    case CO_EXPONENTIAL: return 2;
    case CO_GAMMA: return 3;
    case CO_MONODISPERSE: return 2;
    case CO_LOGNORMAL: return 3;
    default: return -1;
    }
}

/* constructor checks, ParticleDistributions.jl:72-75,101-104,126-129,153-156 */
int co_dist_valid(const co_dist *d) {
    switch (d->type) {
    case CO_EXPONENTIAL:
    case CO_MONODISPERSE: return !(d->n < 0 || d->theta <= 0);
    case CO_GAMMA: return !(d->n < 0 || d->theta <= 0 || d->k <= 0);
    case CO_LOGNORMAL: return !(d->n < 0 || d->k <= 0);
    default: return 0;
    }
}

/* moment_func / moment, ParticleDistributions.jl:177-207, 216-218 */
double co_moment(const co_dist *d, double q) {
    switch (d->type) {
    case CO_EXPONENTIAL: /* n * theta^q * gamma(q + 1) */
        return d->n * pow(d->theta, q) * co_gamma(q + 1.0);
    case CO_GAMMA: /* n * theta^q * gamma(q + k) / gamma(k) */
        return d->n * pow(d->theta, q) * co_gamma(q + d->k) / co_gamma(d->k);
    case CO_MONODISPERSE: /* n * theta^q */
        return d->n * pow(d->theta, q);
    case CO_LOGNORMAL: /* n * exp(q*mu + q^2 sigma^2 / 2) */
        return d->n * exp(q * d->theta + q * q * (d->k * d->k) / 2.0);
    default: return NAN;
    }
}

/* standard normal distribution function */
static double co_norm_cdf(double z) { return 0.5 * erfc(-z * 0.70710678118654752440); }

/* partial_moment_func / partial_moment, ParticleDistributions.jl:226-285.  Lognormal (:255-269): the reference
 * integrates x^q n f(x) over (0, x_threshold) with quadgk; the integral has the closed form
 *   n exp(q mu + q^2 sigma^2 / 2) Phi((ln x_t - mu - q sigma^2) / sigma)
 * (x^q f(x) is a lognormal density of parameters (mu + q sigma^2, sigma) times the q-th moment), restated here;
 * tests/golden/lognormal_adaptive.json pins it against adaptive quadrature of the reference integrand. */
double co_partial_moment(const co_dist *d, double q, double x_threshold) {
    switch (d->type) {
    case CO_EXPONENTIAL:
        return d->n * pow(d->theta, q) * co_gamma_inc_p(q + 1.0, x_threshold / d->theta) * co_gamma(q + 1.0);
    case CO_GAMMA:
        return d->n * pow(d->theta, q) * co_gamma_inc_p(q + d->k, x_threshold / d->theta) * co_gamma(q + d->k) /
               co_gamma(d->k);
    case CO_MONODISPERSE: return (x_threshold < d->theta) ? 0.0 : d->n * pow(d->theta, q);
    case CO_LOGNORMAL: { /* theta = mu, k = sigma */
        const double mu = d->theta, sg = d->k;
        return d->n * exp(q * mu + 0.5 * q * q * sg * sg) * co_norm_cdf((log(x_threshold) - mu - q * sg * sg) / sg);
    }
    default: return NAN;
    }
}

/* get_standard_N_q, ParticleDistributions.jl:634-687: out = (N_liq, N_rai, M_liq, M_rai) */
void co_get_standard_N_q(const co_dist *pdists, int N, double size_cutoff, double *out) {
    double N_liq = 0, N_rai = 0, M_liq = 0, M_rai = 0;
    for (int j = 0; j < N; ++j) N_liq += co_partial_moment(&pdists[j], 0.0, size_cutoff);
    for (int j = 0; j < N; ++j) M_liq += co_partial_moment(&pdists[j], 1.0, size_cutoff);
    for (int j = 0; j < N; ++j) N_rai += co_moment(&pdists[j], 0.0) - co_partial_moment(&pdists[j], 0.0, size_cutoff);
    for (int j = 0; j < N; ++j) M_rai += co_moment(&pdists[j], 1.0) - co_partial_moment(&pdists[j], 1.0, size_cutoff);
    out[0] = N_liq;
    out[1] = N_rai;
    out[2] = M_liq;
    out[3] = M_rai;
}

/* get_moments, ParticleDistributions.jl:293-315 */
void co_get_moments(const co_dist *d, double *out) {
    switch (d->type) {
    case CO_GAMMA:
        out[0] = d->n;
        out[1] = d->n * d->k * d->theta;
        out[2] = d->n * d->k * (d->k + 1.0) * (d->theta * d->theta);
        break;
    case CO_LOGNORMAL:
        out[0] = d->n;
        out[1] = d->n * exp(d->theta + d->k * d->k / 2.0);
        out[2] = d->n * exp(2.0 * d->theta + 2.0 * (d->k * d->k));
        break;
    default:
        out[0] = d->n;
        out[1] = d->n * d->theta;
    }
}

/* normed_density_func, ParticleDistributions.jl:363-388 */
double co_normed_density(const co_dist *d, double x) {
    switch (d->type) {
    case CO_EXPONENTIAL: return 1.0 / d->theta * exp(-x / d->theta);
    case CO_GAMMA: return pow(x, d->k - 1.0) / pow(d->theta, d->k) / co_gamma(d->k) * exp(-x / d->theta);
    case CO_LOGNORMAL: {
        double l = log(x) - d->theta;
        return exp(-(l * l) / (2.0 * (d->k * d->k))) / (x * d->k * sqrt(2.0 * M_PI));
    }
    default: return NAN;
    }
}

/* density_func, ParticleDistributions.jl:323-355 */
double co_density(const co_dist *d, double x) {
    switch (d->type) {
    case CO_EXPONENTIAL: return d->n / d->theta * exp(-x / d->theta);
    case CO_GAMMA:
        return d->n * pow(x, d->k - 1.0) / pow(d->theta, d->k) / co_gamma(d->k) * exp(-x / d->theta);
    case CO_LOGNORMAL: {
        double l = log(x) - d->theta;
        return d->n * exp(-((l * l) / (2.0 * (d->k * d->k)))) / (x * d->k * sqrt(2.0 * M_PI));
    }
    case CO_MONODISPERSE:
        return (fabs(x - d->theta) < d->theta / 10.0) ? d->n / (2.0 * d->theta / 10.0) : 0.0;
    default: return NAN;
    }
}

/* update_dist_from_moments: Gamma ParticleDistributions.jl:456-476, Lognormal :483-505,
 * Exponential :512-523, Monodisperse :530-541.  Returns -1 on wrong arity (MethodError). */
int co_update_dist_from_moments(int dist_type, const double *m, int n_moments, const double k_range[2],
                                co_dist *out) {
    if (n_moments != co_nparams(dist_type)) return -1;
    out->type = dist_type;
    switch (dist_type) {
    case CO_GAMMA:
        if (m[0] > CO_EPS && m[1] > CO_EPS) {
            double kmin = k_range ? k_range[0] : CO_EPS, kmax = k_range ? k_range[1] : 10.0;
            double kk = (m[1] / m[0]) / (m[2] / m[1] - m[1] / m[0]);
            /* max(kmin, min(kmax, kk)) with Julia's NaN-propagating min/max */
            double inner = (isnan(kk)) ? NAN : (kk < kmax ? kk : kmax);
            double k = isnan(inner) ? NAN : (inner > kmin ? inner : kmin);
            out->n = m[0];
            out->k = k;
            out->theta = m[1] / m[0] / k;
        } else {
            out->n = 0.0;
            out->theta = 1.0;
            out->k = 1.0;
        }
        return 0;
    case CO_LOGNORMAL:
        if (m[0] > CO_EPS && m[1] > CO_EPS && m[2] > CO_EPS) {
            double mu = log((m[1] * m[1]) / pow(m[0], 3.0 / 2.0) / pow(m[2], 1.0 / 2.0));
            double sg = sqrt(log(m[0] * m[2] / (m[1] * m[1])));
            if (!(sg > CO_EPS)) sg = CO_EPS; /* max(eps, min(Inf, .)) */
            out->theta = mu;
            out->k = sg;
            out->n = m[1] / exp(mu + 1.0 / 2.0 * (sg * sg));
        } else {
            out->n = 0.0;
            out->theta = 1.0;
            out->k = 1.0;
        }
        return 0;
    case CO_EXPONENTIAL:
    case CO_MONODISPERSE:
        if (m[0] > CO_EPS && m[1] > CO_EPS) {
            out->n = m[0];
            out->theta = m[1] / m[0];
        } else {
            out->n = 0.0;
            out->theta = 1.0;
        }
        out->k = 1.0;
        return 0;
    default: return -1;
    }
}

/* integrate_SimpsonEvenFast, ParticleDistributions.jl:698-710 */
double co_integrate_simpson_even_fast(int n_bins, double dx, double (*y)(int, void *), void *ctx) {
    if (n_bins < 3) return NAN; /* reference: error("n_bins must be at least 3") */
    int e = n_bins + 1;
    double s = 0.0;
    for (int j = 5; j <= n_bins - 3; ++j) s += y(j, ctx);
    double retval = s + (17 * (y(1, ctx) + y(e, ctx)) + 59 * (y(2, ctx) + y(e - 1, ctx)) +
                         43 * (y(3, ctx) + y(e - 2, ctx)) + 49 * (y(4, ctx) + y(e - 3, ctx))) /
                            48;
    return dx * retval;
}

typedef struct {
    int type;
    double theta, k, p1, p2, x_threshold, gamma_p2k, x_min, dx;
    int n_bins;
} co_msh_ctx;

/* y_func(j), ParticleDistributions.jl:583-585 (Exp) / :608-610 (Gamma); logx :566 */
static double co_msh_y(int j, void *vctx) {
    const co_msh_ctx *c = (const co_msh_ctx *)vctx;
    if (j > c->n_bins) return 0.0;
    double lx = c->x_min + (j - 1) * c->dx;
    double x = exp(lx);
    double f;
    if (c->type == CO_EXPONENTIAL) /* :577 */
        f = pow(x, c->p1) * exp(-x / c->theta) * co_gamma_inc_p(c->p2 + 1.0, (c->x_threshold - x) / c->theta) *
            c->gamma_p2k;
    else /* :601-602 */
        f = pow(x, c->p1 + c->k - 1.0) * exp(-x / c->theta) *
            co_gamma_inc_p(c->p2 + c->k, (c->x_threshold - x) / c->theta) * c->gamma_p2k;
    return exp(lx) * f;
}

/* moment_source_helper(::Lognormal...), ParticleDistributions.jl:614-625:
 *   int_0^xt y^p2 n f(y) [ int_0^(xt-y) x^p1 n f(x) dx ] dy      (nested adaptive quadgk in the reference).
 * SAME-RULE restatement of what the HIP kernels evaluate (kernels.hpp, msh_lognormal): the inner integral is the closed
 * form of co_partial_moment; the outer one is a CO_LN_NODES-point midpoint rule in v, y = xt / (1 + e^-v) -- in v the
 * integrand is analytic and decays like a Gaussian on both sides (ln y ~ v for v << 0, ln(xt - y) ~ -v for v >> 0), so
 * the equispaced rule converges geometrically; the range is the window of co_lognormal_msh_range.  With p the size-biased laws,
 *   result = M_p1 M_p2 sum_nodes h G_p2(ln y) (1 - y/xt) Phi((ln(xt - y) - mu - p1 sigma^2) / sigma),
 *   G_q(l) = exp(-(l - mu - q sigma^2)^2 / (2 sigma^2)) / (sigma sqrt(2 pi)).
 * Error against adaptive quadrature of the reference integrand: <= 1e-12 M_p1 M_p2 (tests/test_oracle_kats.py,
 * tests/golden/lognormal_adaptive.json); the reference's own KATs (test_ParticleDistributions_correctness.jl:215-218)
 * are reproduced to their 4 digits. */
#define CO_LN_NODES 48
#define CO_LN_SIGMA_FLOOR 1e-6
#define CO_LN_MAXORDER 7 /* M = P + 2 <= 7 orders share one range, as the kernel's single pass over the nodes does */
static double co_softplus(double x) { return x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
/* The window of v in which the integrand lives, through the EXACT map v(ln y) = d - ln(1 - e^d), d = ln y - ln xt < 0
 * (round 4): 8.5 sigma of the density of the lowest order on the left, and on the right whichever comes first
 * of the end of the density of the top order and the decay of the inner Phi, ln(xt - y) = mu - 8.5 sigma, i.e.
 * v = ln(e^delta - 1), delta = ln xt - mu + 8.5 sigma.  The window is then ~17 sigma wide in ln y whatever ln xt - mu is,
 * so the node spacing scales with sigma.  (Rounds 2-3 took v = ln y - ln xt on the left -- only true for v << 0 -- and
 * v = delta on the right: for x_t ~ 2 e^mu most of the 48 nodes then fell outside the window, and the rule was off by
 * 2.6e-6 at sigma = 0.01, 6e-3 at 0.005.)  An empty window (vhi <= vlo): the integral is below e^-36 of M_p1 M_p2 -> 0. */
void co_lognormal_msh_range(double mu, double sg, double lxt, int n_orders, double *vlo, double *vhi) {
    const double d_lo = (mu - 8.5 * sg) - lxt;                            /* lower end of the density (order 0) */
    const double d_hi = (mu + (n_orders - 1) * sg * sg + 8.5 * sg) - lxt; /* upper end of the density of the top order */
    const double delta = (lxt - mu) + 8.5 * sg;                           /* where the inner Phi has decayed */
    *vlo = *vhi = 0.0;
    if (!(d_lo < -1e-9) || !(delta > 1e-9)) return; /* empty */
    *vlo = d_lo - log(-expm1(d_lo));
    *vhi = delta > 36.0 ? delta : log(expm1(delta));
    if (d_hi < -1e-9) *vhi = fmin(*vhi, d_hi - log(-expm1(d_hi)));
    if (*vhi < *vlo) *vhi = *vlo;
}
double co_moment_source_helper_lognormal(const co_dist *dist, double p1, double p2, double x_threshold, int n_orders) {
    const double mu = dist->theta, sg = dist->k, lxt = log(x_threshold);
    /* a closure clamped to sigma = eps is a point mass at e^mu: Prob(X + Y < xt) = [2 e^mu < xt] (below CO_LN_SIGMA_FLOOR
     * the rule's own arithmetic is rounding noise) */
    if (sg < CO_LN_SIGMA_FLOOR) return co_moment(dist, p1) * co_moment(dist, p2) * ((mu + 0.6931471805599453 < lxt) ? 1.0 : 0.0);
    double vlo, vhi;
    co_lognormal_msh_range(mu, sg, lxt, n_orders, &vlo, &vhi);
    const double h = (vhi - vlo) / CO_LN_NODES;
    double sum = 0.0;
    for (int j = 0; j < CO_LN_NODES; ++j) {
        const double v = vlo + h * (j + 0.5);
        const double ly = lxt - co_softplus(-v), lz = lxt - co_softplus(v); /* ln y, ln(xt - y) */
        const double u = (ly - mu - p2 * sg * sg) / sg;
        const double G = exp(-0.5 * u * u) / (sg * 2.5066282746310002);
        sum += G * exp(lz - lxt) * co_norm_cdf((lz - mu) / sg - p1 * sg);
    }
    return co_moment(dist, p1) * co_moment(dist, p2) * (h * sum);
}

/* moment_source_helper: Monodisperse ParticleDistributions.jl:557-564, Exponential :567-587,
 * Gamma :589-612, Lognormal :614-625 (same-rule restatement above). */
double co_moment_source_helper(const co_dist *d, double p1, double p2, double x_threshold,
                               int n_bins_per_log_unit) {
    if (d->type == CO_MONODISPERSE)
        return (d->theta < x_threshold / 2.0) ? (d->n * d->n) * pow(d->theta, p1 + p2) : 0.0;
    if (d->type == CO_LOGNORMAL) return co_moment_source_helper_lognormal(d, p1, p2, x_threshold, CO_LN_MAXORDER);
    if (d->type != CO_EXPONENTIAL && d->type != CO_GAMMA) return NAN;
    co_msh_ctx c;
    c.type = d->type;
    c.theta = d->theta;
    c.k = d->k;
    c.p1 = p1;
    c.p2 = p2;
    c.x_threshold = x_threshold;
    double x_lowerbound = fmin(1e-5, 1e-5 * x_threshold);
    c.n_bins = (int)floor(n_bins_per_log_unit * log10(x_threshold / x_lowerbound));
    c.x_min = log(x_lowerbound);
    c.dx = (log(x_threshold) - log(x_lowerbound)) / c.n_bins;
    if (d->type == CO_EXPONENTIAL) {
        c.gamma_p2k = co_gamma(p2 + 1.0);
        return (d->n * d->n) * pow(d->theta, p2 - 1.0) *
               co_integrate_simpson_even_fast(c.n_bins, c.dx, co_msh_y, &c);
    } else {
        double gamma_k = co_gamma(d->k);
        c.gamma_p2k = co_gamma(p2 + d->k);
        return (d->n * d->n) * pow(d->theta, p2 - d->k) / (gamma_k * gamma_k) *
               co_integrate_simpson_even_fast(c.n_bins, c.dx, co_msh_y, &c);
    }
}

/* compute_threshold, ParticleDistributions.jl:747-761 */
double co_compute_threshold(const co_dist *d, double percentile, double minx) {
    if (d->type == CO_EXPONENTIAL) return fmax(-d->theta * log(1.0 - percentile), minx);
    if (d->type == CO_GAMMA) return fmax(d->theta * co_gamma_inc_inv(d->k, percentile, 1.0 - percentile), minx);
    return NAN;
}

/* compute_thresholds, ParticleDistributions.jl:734-745 (percentiles == NULL -> :721-732, 0.97) */
void co_compute_thresholds(const co_dist *pdists, int N, const double *percentiles, double *out) {
    for (int i = 0; i < N; ++i) {
        if (i == N - 1)
            out[i] = INFINITY;
        else
            out[i] = co_compute_threshold(&pdists[i], percentiles ? percentiles[i] : 0.97, 1e-18);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* src/Sources/Coalescence.jl -- AnalyticalCoalStyle                                           */
/* ------------------------------------------------------------------------------------------ */

/* CoalescenceData constructor, Coalescence.jl:55-87.  kernel_c is [N][N][P][P] (un-normalised). */
int co_coalescence_data_init(co_coal_data *cd, int N, int P, const double *kernel_c, const int *NProgMoms,
                             const double *dist_thresholds, const double norms[2], int threshold_style) {
    if (N < 1 || N > CO_MAX_MODES || P < 1 || P > CO_MAX_P) return -1;
    memset(cd, 0, sizeof(*cd));
    cd->N = N;
    cd->P = P;
    for (int j = 0; j < N; ++j)
        for (int k = 0; k < N; ++k) {
            const double *c = kernel_c + ((size_t)(j * N + k)) * P * P;
            if (co_check_symmetry(c, P) != 0) return -2;
            double tmp[CO_MAX_P * CO_MAX_P];
            co_get_normalized_kernel_tensor(c, P, norms, tmp); /* :63-67 */
            for (int a = 0; a < P; ++a)
                for (int b = 0; b < P; ++b) cd->c[j][k][a][b] = tmp[a * P + b];
        }
    int mx = 0;
    for (int i = 0; i < N; ++i)
        if (NProgMoms[i] > mx) mx = NProgMoms[i];
    cd->N_mom_max = mx + (P - 1); /* :69 */
    for (int i = 0; i < N; ++i) { /* :70-76 */
        if (i < N - 1)
            cd->N_2d_ints[i] = (P - 1) + (NProgMoms[i] > NProgMoms[i + 1] ? NProgMoms[i] : NProgMoms[i + 1]);
        else
            cd->N_2d_ints[i] = (P - 1) + NProgMoms[i];
    }
    for (int i = 0; i < N; ++i) /* :78-84 */
        cd->dist_thresholds[i] =
            (threshold_style == CO_FIXED_THRESHOLD) ? dist_thresholds[i] / norms[1] : dist_thresholds[i];
    return 0;
}

/* get_moments_matrix, Coalescence.jl:187-198: moments[i][j] = j < N_mom_max ? moment(pdists[i], j) : 0 */
void co_get_moments_matrix(const co_dist *pdists, int N, int M, int N_mom_max, double *moments) {
    for (int i = 0; i < N; ++i)
        for (int j = 1; j <= M; ++j)
            moments[i * M + (j - 1)] = (j <= N_mom_max) ? co_moment(&pdists[i], (double)(j - 1)) : 0.0;
}

/* get_finite_2d_integrals, Coalescence.jl:200-244.  F is [N][M][M], symmetric per mode. */
void co_get_finite_2d_integrals(const co_dist *pdists, int N, int M, const double *thresholds,
                                const double *moments, const int *N_2d_ints, double *F) {
    for (int i = 0; i < N; ++i) {
        double *Fi = F + (size_t)i * M * M;
        for (int j = 1; j <= M; ++j)
            for (int k = 1; k <= M; ++k) {
                double v;
                double mom_times_mom = moments[i * M + (j - 1)] * moments[i * M + (k - 1)];
                if (mom_times_mom < CO_EPS || k < j || N_2d_ints[i] < j || N_2d_ints[i] < k)
                    v = 0.0;
                else if (i == N - 1 || isinf(thresholds[i]))
                    v = mom_times_mom;
                else {
                    /* (Lognormal: the kernel's single pass serves all M orders of the mode with one node range) */
                    double h = pdists[i].type == CO_LOGNORMAL
                                   ? co_moment_source_helper_lognormal(&pdists[i], (double)(j - 1), (double)(k - 1),
                                                                       thresholds[i], M)
                                   : co_moment_source_helper(&pdists[i], (double)(j - 1), (double)(k - 1),
                                                             thresholds[i], 15);
                    v = (h < mom_times_mom) ? h : mom_times_mom; /* min(mom_times_mom, h) */
                    if (isnan(h)) v = NAN;
                }
                Fi[(j - 1) * M + (k - 1)] = v;
            }
        for (int j = 1; j <= M; ++j) /* :232-240 mirror the upper triangle */
            for (int k = 1; k < j; ++k) Fi[(j - 1) * M + (k - 1)] = Fi[(k - 1) * M + (j - 1)];
    }
}

static double co_binomial(int n, int k) {
    double r = 1.0;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return r;
}

#define MOM(i, q) moments[(i) * M + (q)]

/* Q_jk, Coalescence.jl:283-309 (0-based j,k) */
static double co_Q_jk(int mo, int j, int k, const double *moments, int M, int P,
                      const double (*c)[CO_MAX_P], double *magout) {
    double s = 0.0, mag = 0.0;
    for (int a = 0; a < P; ++a) {
        double sa = 0.0;
        for (int b = 0; b < P; ++b) {
            double sb = 0.0;
            for (int cc = 0; cc <= mo; ++cc) {
                double t = c[a][b] * co_binomial(mo, cc) * MOM(j, a + cc) * MOM(k, b + mo - cc);
                sb += t;
                mag += fabs(t);
            }
            sa += sb;
        }
        s += sa;
    }
    *magout += mag;
    return s;
}

/* R_jk, Coalescence.jl:334-351 */
static double co_R_jk(int mo, int j, int k, const double *moments, int M, int P,
                      const double (*c)[CO_MAX_P], double *magout) {
    double s = 0.0, mag = 0.0;
    for (int a = 0; a < P; ++a) {
        double sa = 0.0;
        for (int b = 0; b < P; ++b) {
            double t = c[a][b] * MOM(j, a) * MOM(k, b + mo);
            sa += t;
            mag += fabs(t);
        }
        s += sa;
    }
    *magout += mag;
    return s;
}

/* S_1k, Coalescence.jl:398-424 */
static double co_S_1k(int mo, int k, int M, int P, const double *Fk, const double (*c)[CO_MAX_P], double *magout) {
    double s = 0.0, mag = 0.0;
    for (int a = 0; a < P; ++a) {
        double sa = 0.0;
        for (int b = 0; b < P; ++b) {
            double sb = 0.0;
            for (int cc = 0; cc <= mo; ++cc) {
                double t = 0.5 * c[a][b] * co_binomial(mo, cc) * Fk[(a + cc) * M + (b + mo - cc)];
                sb += t;
                mag += fabs(t);
            }
            sa += sb;
        }
        s += sa;
    }
    (void)k;
    *magout += mag;
    return s;
}

/* S_2k, Coalescence.jl:426-455 */
static double co_S_2k(int mo, int k, const double *moments, int M, int P, const double *Fk,
                      const double (*c)[CO_MAX_P], double *magout) {
    double s = 0.0, mag = 0.0;
    for (int a = 0; a < P; ++a) {
        double sa = 0.0;
        for (int b = 0; b < P; ++b) {
            double sb = 0.0;
            for (int cc = 0; cc <= mo; ++cc) {
                double t = 0.5 * c[a][b] * co_binomial(mo, cc) *
                           (MOM(k, a + cc) * MOM(k, b + mo - cc) - Fk[(a + cc) * M + (b + mo - cc)]);
                sb += t;
                mag += fabs(0.5 * c[a][b] * co_binomial(mo, cc)) *
                       (fabs(MOM(k, a + cc) * MOM(k, b + mo - cc)) + fabs(Fk[(a + cc) * M + (b + mo - cc)]));
            }
            sa += sb;
        }
        s += sa;
    }
    *magout += mag;
    return s;
}

/* get_coal_ints, Coalescence.jl:115-150 (FixedThreshold) and :152-185 (MovingThreshold);
 * Q/R/S matrices :260-281, :311-332, :353-396. */
int co_get_coal_ints(const co_dist *pdists, const co_coal_data *cd, int threshold_style, double *out,
                     double *scale) {
    const int N = cd->N, P = cd->P, M = P + 2;
    int NProgMoms[CO_MAX_MODES];
    for (int i = 0; i < N; ++i) NProgMoms[i] = co_nparams(pdists[i].type);

    double moments[CO_MAX_MODES * CO_MAX_M];
    double F[CO_MAX_MODES * CO_MAX_M * CO_MAX_M];
    double thresholds[CO_MAX_MODES];
    co_get_moments_matrix(pdists, N, M, cd->N_mom_max, moments);
    if (threshold_style == CO_MOVING_THRESHOLD)
        co_compute_thresholds(pdists, N, cd->dist_thresholds, thresholds); /* :164 */
    else
        memcpy(thresholds, cd->dist_thresholds, sizeof(double) * N);
    co_get_finite_2d_integrals(pdists, N, M, thresholds, moments, cd->N_2d_ints, F);

    /* Q[mo][j][k], R[mo][j][k], S[mo][0..1][k] */
    double Q[3][CO_MAX_MODES][CO_MAX_MODES], R[3][CO_MAX_MODES][CO_MAX_MODES], S[3][2][CO_MAX_MODES];
    double Qm[3][CO_MAX_MODES][CO_MAX_MODES], Rm[3][CO_MAX_MODES][CO_MAX_MODES], Sm[3][2][CO_MAX_MODES];
    memset(Qm, 0, sizeof(Qm));
    memset(Rm, 0, sizeof(Rm));
    memset(Sm, 0, sizeof(Sm));
    for (int mo = 0; mo < 3; ++mo) {
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < N; ++j) {
                /* :272  k <= j || NProgMoms[k] <= moment_order */
                if (k <= j || NProgMoms[k] <= mo)
                    Q[mo][j][k] = 0.0;
                else
                    Q[mo][j][k] = co_Q_jk(mo, j, k, moments, M, P, cd->c[j][k], &Qm[mo][j][k]);
                /* :323 */
                if (NProgMoms[k] <= mo)
                    R[mo][j][k] = 0.0;
                else
                    R[mo][j][k] = co_R_jk(mo, j, k, moments, M, P, cd->c[j][k], &Rm[mo][j][k]);
            }
        for (int k = 0; k < N; ++k) { /* :365-391 */
            if (k < N - 1 && NProgMoms[k] <= mo && NProgMoms[k + 1] <= mo) {
                S[mo][0][k] = S[mo][1][k] = 0.0;
            } else if (k == N - 1 && NProgMoms[k] <= mo) {
                S[mo][0][k] = S[mo][1][k] = 0.0;
            } else {
                const double *Fk = F + (size_t)k * M * M;
                S[mo][0][k] = co_S_1k(mo, k, M, P, Fk, cd->c[k][k], &Sm[mo][0][k]);
                S[mo][1][k] = co_S_2k(mo, k, moments, M, P, Fk, cd->c[k][k], &Sm[mo][1][k]);
            }
        }
    }
    /* assembly :140-149 / :175-184 */
    int idx = 0;
    for (int k = 0; k < N; ++k)
        for (int m = 0; m < NProgMoms[k]; ++m) {
            double sq = 0.0, sr = 0.0, mq = 0.0, mr = 0.0;
            for (int j = 0; j < N; ++j) {
                sq += Q[m][j][k];
                mq += Qm[m][j][k];
            }
            for (int j = 0; j < N; ++j) {
                sr += R[m][j][k];
                mr += Rm[m][j][k];
            }
            double v, mg;
            if (k == 0) {
                v = sq - sr + S[m][0][k];
                mg = mq + mr + Sm[m][0][k];
            } else {
                v = sq - sr + S[m][0][k] + S[m][1][k - 1];
                mg = mq + mr + Sm[m][0][k] + Sm[m][1][k - 1];
            }
            out[idx] = v;
            if (scale) scale[idx] = mg;
            ++idx;
        }
    return idx;
}

/* weighting_fn, Coalescence.jl:624-642 (1-based k; NaN where the reference throws) */
double co_weighting_fn(double x, int k, const co_dist *pdists, int N) {
    double denom = 0.0, num = 0.0;
    if (k > N) return NAN;
    for (int j = 1; j <= N; ++j) {
        denom += co_normed_density(&pdists[j - 1], x);
        if (j <= k) num += co_normed_density(&pdists[j - 1], x);
    }
    if (denom == 0.0) return 0.0;
    return num / denom;
}

/* ------------------------------------------------------------------------------------------ */
/* src/Sources/Sedimentation.jl:22-37                                                          */
/* ------------------------------------------------------------------------------------------ */
void co_get_sedimentation_flux(const co_dist *pdists, int N, const double (*vel)[2], int n_vel, double *out) {
    int idx = 0;
    for (int i = 0; i < N; ++i) {
        int np = co_nparams(pdists[i].type);
        for (int j = 1; j <= np; ++j) {
            double s = 0.0;
            for (int k = 0; k < n_vel; ++k) s += -vel[k][0] * co_moment(&pdists[i], (double)(j - 1) + vel[k][1]);
            out[idx++] = s;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* src/Sources/Condensation.jl:22-37                                                           */
/* ------------------------------------------------------------------------------------------ */
void co_get_cond_evap(const co_dist *pdists, int N, double s, double xi, double rho_l, double *out) {
    int idx = 0;
    for (int i = 0; i < N; ++i) {
        int np = co_nparams(pdists[i].type);
        for (int j = 1; j <= np; ++j) {
            if (j < 2)
                out[idx++] = 0.0;
            else /* 3 * xi * s * (j - 1) * moment(pdist, j - 1 - 2/3) * (4pi/3)^(2/3) / rho_l^(1/3) */
                out[idx++] = 3 * xi * s * (j - 1) * co_moment(&pdists[i], (double)j - 1.0 - 2.0 / 3.0) *
                             pow(4 * M_PI / 3, 2.0 / 3.0) / pow(rho_l, 1.0 / 3.0);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* test/examples/utils/box_model_helpers.jl:29-53  rhs_coal!                                    */
/* ------------------------------------------------------------------------------------------ */
static int co_invert_all(const co_params *p, const double *mom_normalized, co_dist *pdists) {
    int off = 0;
    for (int i = 0; i < p->N; ++i) { /* :32-38 */
        if (co_update_dist_from_moments(p->dist_type[i], mom_normalized + off, p->NProgMoms[i], p->k_range,
                                        &pdists[i]) != 0)
            return -1;
        off += p->NProgMoms[i];
    }
    return off;
}

int co_rhs_coal(const co_params *p, const double *mom, double *dmom, double *scale) {
    double mom_norms[CO_MAX_MODES * 3], mom_normalized[CO_MAX_MODES * 3], coal_ints[CO_MAX_MODES * 3];
    co_dist pdists[CO_MAX_MODES];
    int nmom = co_get_moments_normalizing_factors(p->NProgMoms, p->N, p->norms, mom_norms); /* :30 */
    if (nmom < 0) return -1;
    for (int q = 0; q < nmom; ++q) mom_normalized[q] = mom[q] / mom_norms[q]; /* :31 */
    if (co_invert_all(p, mom_normalized, pdists) < 0) return -1;
    co_get_coal_ints(pdists, &p->coal_data, p->threshold_style, coal_ints, scale); /* :40-46 */
    for (int q = 0; q < nmom; ++q) {
        dmom[q] = coal_ints[q] * mom_norms[q]; /* :52 */
        if (scale) scale[q] *= mom_norms[q];
    }
    return nmom;
}

/* rhs_condensation!, box_model_helpers.jl:55-67: xi_normalized = p.xi / norms[2]^(2/3) */
int co_rhs_condensation(const co_params *p, double xi, double s, const double *mom, double *dmom) {
    double mom_norms[CO_MAX_MODES * 3], mom_normalized[CO_MAX_MODES * 3], ce[CO_MAX_MODES * 3];
    co_dist pdists[CO_MAX_MODES];
    int nmom = co_get_moments_normalizing_factors(p->NProgMoms, p->N, p->norms, mom_norms);
    if (nmom < 0) return -1;
    for (int q = 0; q < nmom; ++q) mom_normalized[q] = mom[q] / mom_norms[q];
    if (co_invert_all(p, mom_normalized, pdists) < 0) return -1;
    double xi_normalized = xi / pow(p->norms[1], 2.0 / 3.0);
    co_get_cond_evap(pdists, p->N, s, xi_normalized, 1000.0, ce);
    for (int q = 0; q < nmom; ++q) dmom[q] = ce[q] * mom_norms[q];
    return nmom;
}

int co_rhs_condensation_batch(const co_params *p, double xi, const double *s_per_parcel, double s_scalar, long n,
                              long ld, const double *mom, double *dmom) {
    int nmom = 0;
    for (int i = 0; i < p->N; ++i) nmom += p->NProgMoms[i];
    for (long i = 0; i < n; ++i) {
        double m[CO_MAX_MODES * 3], d[CO_MAX_MODES * 3];
        for (int q = 0; q < nmom; ++q) m[q] = mom[(size_t)q * ld + i];
        co_rhs_condensation(p, xi, s_per_parcel ? s_per_parcel[i] : s_scalar, m, d);
        for (int q = 0; q < nmom; ++q) dmom[(size_t)q * ld + i] = d[q];
    }
    return nmom;
}

int co_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* moment-major SoA batch: mom[q*ld + parcel], the layout of Julia's m[parcel, moment]
 * (rainshaft_helpers.jl:48-56). */
int co_rhs_coal_batch(const co_params *p, long n_parcels, long ld, const double *mom, double *dmom, double *scale,
                      int n_threads) {
    int nmom = 0;
    for (int i = 0; i < p->N; ++i) nmom += p->NProgMoms[i];
    int err = 0;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
    for (long i = 0; i < n_parcels; ++i) {
        double m[CO_MAX_MODES * 3], d[CO_MAX_MODES * 3], s[CO_MAX_MODES * 3];
        for (int q = 0; q < nmom; ++q) m[q] = mom[(size_t)q * ld + i];
        if (co_rhs_coal(p, m, d, scale ? s : NULL) < 0) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            err = 1;
        }
        for (int q = 0; q < nmom; ++q) {
            dmom[(size_t)q * ld + i] = d[q];
            if (scale) scale[(size_t)q * ld + i] = s[q];
        }
    }
    (void)n_threads;
    return err ? -1 : nmom;
}

/* rainshaft_helpers.jl:52-78: one cell, without the inter-cell flux divergence (:80-86).
 * negatives clamped to zero (:52), coalescence skipped when all normalised moments < eps (:67-68),
 * velocity coefficients rescaled by norms[2]^v[2] (:74-76). */
int co_rainshaft_cell(const co_params *p, const double *mom_in, double *coal_source, double *sedi_flux) {
    double mom_norms[CO_MAX_MODES * 3], mz[CO_MAX_MODES * 3], ci[CO_MAX_MODES * 3], sf[CO_MAX_MODES * 3];
    co_dist pdists[CO_MAX_MODES];
    int nmom = co_get_moments_normalizing_factors(p->NProgMoms, p->N, p->norms, mom_norms);
    if (nmom < 0) return -1;
    int all_small = 1;
    for (int q = 0; q < nmom; ++q) {
        double m = mom_in[q] < 0 ? 0.0 : mom_in[q];
        mz[q] = m / mom_norms[q];
        if (!(mz[q] < CO_EPS)) all_small = 0;
    }
    if (co_invert_all(p, mz, pdists) < 0) return -1;
    if (all_small) {
        for (int q = 0; q < nmom; ++q) coal_source[q] = 0.0;
    } else {
        co_get_coal_ints(pdists, &p->coal_data, CO_FIXED_THRESHOLD, ci, NULL);
        for (int q = 0; q < nmom; ++q) coal_source[q] = ci[q] * mom_norms[q];
    }
    double veln[CO_MAX_VEL][2];
    for (int k = 0; k < p->n_vel; ++k) {
        veln[k][0] = p->vel[k][0] * pow(p->norms[1], p->vel[k][1]);
        veln[k][1] = p->vel[k][1];
    }
    co_get_sedimentation_flux(pdists, p->N, (const double(*)[2])veln, p->n_vel, sf);
    for (int q = 0; q < nmom; ++q) sedi_flux[q] = sf[q] * mom_norms[q];
    return nmom;
}

int co_rainshaft_cell_batch(const co_params *p, long n, long ld, const double *mom, double *coal_source,
                            double *sedi_flux, int n_threads) {
    int nmom = 0;
    for (int i = 0; i < p->N; ++i) nmom += p->NProgMoms[i];
#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
    for (long i = 0; i < n; ++i) {
        double m[CO_MAX_MODES * 3], cs[CO_MAX_MODES * 3], sf[CO_MAX_MODES * 3];
        for (int q = 0; q < nmom; ++q) m[q] = mom[(size_t)q * ld + i];
        co_rainshaft_cell(p, m, cs, sf);
        for (int q = 0; q < nmom; ++q) {
            coal_source[(size_t)q * ld + i] = cs[q];
            sedi_flux[(size_t)q * ld + i] = sf[q];
        }
    }
    (void)n_threads;
    return nmom;
}

/* normalise + update_dist_from_moments for a batch; params planes: (n, theta, k) per mode */
/* check_moment_consistency, ParticleDistributions.jl:437-449: 0 = consistent, 1 = a negative moment (:439), 2 = a negative
 * even-ordered central moment (:443-448; mapreduce(+) over i = 0..order: left to right).  IEEE division as in Julia. */
int co_check_moment_consistency(const double *m, int n_moments) {
    for (int i = 0; i < n_moments; ++i)
        if (m[i] < 0.0) return 1;
    for (int order = 2; order <= n_moments - 1; order += 2) {
        double cm = 0.0;
        for (int i = 0; i <= order; ++i) {
            const double sgn = (i % 2) ? -1.0 : 1.0, r = m[1] / m[0];
            double ri = 1.0;
            for (int e = 0; e < i; ++e) ri *= r; /* (m[2] / m[1])^i: Julia's integer power by repeated multiplication for i <= 3 */
            const double t = (co_binomial(order, i) * sgn) * ri * (m[order - i] / m[0]);
            cm = (i == 0) ? t : cm + t;
        }
        if (cm < 0.0) return 2;
    }
    return 0;
}

/* the batch form of the reference's silent clamps and of check_moment_consistency (csrc: cloudy_closure_stats):
 * out[4 * mode + c], c = 0 fallback (0, 1, 1), 1 shape at the lower clamp, 2 at the upper clamp, 3 inconsistent moments */
int co_closure_stats(const co_params *p, long n, long ld, const double *mom, unsigned long long *out) {
    double mom_norms[CO_MAX_MODES * 3];
    int nmom = co_get_moments_normalizing_factors(p->NProgMoms, p->N, p->norms, mom_norms);
    if (nmom < 0) return -1;
    for (int q = 0; q < 4 * p->N; ++q) out[q] = 0;
    for (long i = 0; i < n; ++i) {
        double mz[CO_MAX_MODES * 3];
        co_dist pd[CO_MAX_MODES];
        for (int q = 0; q < nmom; ++q) mz[q] = mom[(size_t)q * ld + i] / mom_norms[q];
        if (co_invert_all(p, mz, pd) < 0) return -1;
        int off = 0;
        for (int m = 0; m < p->N; ++m) {
            const double *mm = mz + off;
            const int np = p->NProgMoms[m], t = p->dist_type[m];
            const int fb = t == CO_LOGNORMAL ? !(mm[0] > CO_EPS && mm[1] > CO_EPS && mm[2] > CO_EPS) : !(mm[0] > CO_EPS && mm[1] > CO_EPS);
            const double kmin = p->k_range[0], kmax = p->k_range[1];
            out[4 * m + 0] += fb;
            out[4 * m + 1] += !fb && ((t == CO_GAMMA && pd[m].k == kmin) || (t == CO_LOGNORMAL && pd[m].k == CO_EPS));
            out[4 * m + 2] += !fb && t == CO_GAMMA && pd[m].k == kmax;
            out[4 * m + 3] += co_check_moment_consistency(mm, np) != 0;
            off += np;
        }
    }
    return 0;
}

int co_update_dist_batch(const co_params *p, long n, long ld, const double *mom, double *params) {
    double mom_norms[CO_MAX_MODES * 3];
    int nmom = co_get_moments_normalizing_factors(p->NProgMoms, p->N, p->norms, mom_norms);
    if (nmom < 0) return -1;
    for (long i = 0; i < n; ++i) {
        double mz[CO_MAX_MODES * 3];
        co_dist pd[CO_MAX_MODES];
        for (int q = 0; q < nmom; ++q) mz[q] = mom[(size_t)q * ld + i] / mom_norms[q];
        if (co_invert_all(p, mz, pd) < 0) return -1;
        for (int m = 0; m < p->N; ++m) {
            params[(size_t)(3 * m + 0) * ld + i] = pd[m].n;
            params[(size_t)(3 * m + 1) * ld + i] = pd[m].theta;
            params[(size_t)(3 * m + 2) * ld + i] = pd[m].k;
        }
    }
    return 0;
}
