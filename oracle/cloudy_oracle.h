/*
 * cloudy_oracle.h -- CPU restatement of the Cloudy.jl collision-coalescence moment RHS.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP path in
 * cloudy.jl_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link, load or call it; the product path never does.
 *
 * The reference (CliMA/Cloudy.jl v0.6.0) is Julia and cannot run in this image
 * (no julia binary), so this file restates the reference algorithm function by
 * function in plain C.  Every function cites the reference file:line it follows
 * (paths relative to the reference checkout).  Third-party numerics the reference
 * pulls from un-vendored packages are restated from their published definitions:
 *   SpecialFunctions.jl (compat 2.5): gamma, gamma_inc(a,x)[1] = P(a,x),
 *                                     gamma_inc_inv(a,p,q)
 * QuadGK / Optim are not on the AnalyticalCoalStyle path and are not restated.
 *
 * Pinning: tests/test_oracle_kats.py checks this oracle against every known-answer
 * value the reference's own unit tests hold for this path (tests/golden/
 * reference_kats.json, each entry citing its test file:line).  What those tests do
 * not pin is listed in DESIGN.md ("parity unpinned" items).
 */
#ifndef CLOUDY_ORACLE_H
#define CLOUDY_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define CO_MAX_MODES 8
#define CO_MAX_P 8            /* tensor order + 1 */
#define CO_MAX_M (CO_MAX_P + 2)
#define CO_MAX_VEL 8
#define CO_MAX_QUAD 64        /* points of the fixed Gauss rule (cloudy_oracle_quad.c) */

enum { CO_EXPONENTIAL = 0, CO_GAMMA = 1, CO_MONODISPERSE = 2, CO_LOGNORMAL = 3 };
enum { CO_FIXED_THRESHOLD = 0, CO_MOVING_THRESHOLD = 1 };

/* PrimitiveParticleDistribution (ParticleDistributions.jl:66-159).
 * Exponential: (n, theta); Gamma: (n, theta, k); Monodisperse: (n, theta);
 * Lognormal: (n, mu, sigma) stored as (n, theta=mu, k=sigma). */
typedef struct {
    int type;
    double n, theta, k;
} co_dist;

/* CoalescenceData (Coalescence.jl:45-106). */
typedef struct {
    int N, P;
    int N_mom_max;
    int N_2d_ints[CO_MAX_MODES];
    double dist_thresholds[CO_MAX_MODES];
    double c[CO_MAX_MODES][CO_MAX_MODES][CO_MAX_P][CO_MAX_P]; /* kernels[j][k].c[a+1,b+1] */
} co_coal_data;

/* the ODE_parameters NamedTuple read by rhs_coal! (box_model_helpers.jl:29-53) */
typedef struct {
    int N;
    int dist_type[CO_MAX_MODES];
    int NProgMoms[CO_MAX_MODES];
    double norms[2];
    double k_range[2];         /* param_range.k, ParticleDistributions.jl:459 */
    int threshold_style;
    co_coal_data coal_data;
    int n_vel;
    double vel[CO_MAX_VEL][2]; /* rainshaft p.vel */
} co_params;

/* ---- SpecialFunctions.jl restatements ---- */
double co_gamma(double x);
double co_gamma_inc_p(double a, double x);              /* gamma_inc(a, x)[1] */
double co_gamma_inc_inv(double a, double p, double q);  /* gamma_inc_inv(a, p, q) */

/* ---- helper_functions.jl ---- */
int co_get_dist_moment_ind(const int *NProgMoms, int N, int i, int m);       /* 1-based, <0 on error */
int co_get_dist_moments_ind_range(const int *NProgMoms, int N, int i, int *first, int *last);
int co_get_moments_normalizing_factors(const int *NProgMoms, int N, const double norms[2], double *out);

/* ---- KernelTensors.jl / KernelFunctions.jl ---- */
int co_check_symmetry(const double *c, int P);                                /* 0 ok */
void co_get_normalized_kernel_tensor(const double *c, int P, const double norms[2], double *out);
double co_constant_kernel(double rate, double x, double y);
double co_linear_kernel(double rate, double x, double y);
double co_hydrodynamic_kernel(double coal_eff, double x, double y);
double co_long_kernel(double x_thr, double below, double above, double x, double y);

/* ---- ParticleDistributions.jl ---- */
int co_nparams(int dist_type);
int co_dist_valid(const co_dist *d);
double co_moment(const co_dist *d, double q);
void co_get_moments(const co_dist *d, double *out);
double co_partial_moment(const co_dist *d, double q, double x_threshold);
void co_get_standard_N_q(const co_dist *pdists, int N, double size_cutoff, double *out);
double co_density(const co_dist *d, double x);
double co_normed_density(const co_dist *d, double x);
int co_update_dist_from_moments(int dist_type, const double *moments, int n_moments,
                                const double k_range[2], co_dist *out);
/* Lognormal :614-625 by the rule the HIP kernels use (n_orders = P + 2: the orders that share one node range) */
double co_moment_source_helper_lognormal(const co_dist *dist, double p1, double p2, double x_threshold, int n_orders);
double co_moment_source_helper(const co_dist *d, double p1, double p2, double x_threshold,
                               int n_bins_per_log_unit);
double co_integrate_simpson_even_fast(int n_bins, double dx, double (*y)(int j, void *ctx), void *ctx);
double co_compute_threshold(const co_dist *d, double percentile, double minx);
void co_compute_thresholds(const co_dist *pdists, int N, const double *percentiles, double *out);

/* ---- Coalescence.jl (AnalyticalCoalStyle) ---- */
int co_coalescence_data_init(co_coal_data *cd, int N, int P, const double *kernel_c /* [N][N][P][P] */,
                             const int *NProgMoms, const double *dist_thresholds,
                             const double norms[2], int threshold_style);
/* out: sum(NProgMoms) tendencies, mode-major; scale (optional): sum of |terms| per output,
 * the magnitude against which cancellation-limited tolerances are set in tests. */
int co_get_coal_ints(const co_dist *pdists, const co_coal_data *cd, int threshold_style,
                     double *out, double *scale);
void co_get_moments_matrix(const co_dist *pdists, int N, int M, int N_mom_max, double *moments /* [N][M] */);
void co_get_finite_2d_integrals(const co_dist *pdists, int N, int M, const double *thresholds,
                                const double *moments, const int *N_2d_ints,
                                double *F /* [N][M][M] */);
double co_weighting_fn(double x, int k, const co_dist *pdists, int N);         /* Coalescence.jl:624-642 */

/* ---- Sedimentation.jl ---- */
void co_get_sedimentation_flux(const co_dist *pdists, int N, const double (*vel)[2], int n_vel, double *out);

/* ---- Condensation.jl / rhs_condensation! (box_model_helpers.jl:55-67) ---- */
void co_get_cond_evap(const co_dist *pdists, int N, double s, double xi, double rho_l, double *out);
int co_rhs_condensation(const co_params *p, double xi, double s, const double *mom, double *dmom);
int co_rhs_condensation_batch(const co_params *p, double xi, const double *s_per_parcel, double s_scalar, long n,
                              long ld, const double *mom, double *dmom);

/* ---- box_model_helpers.jl rhs_coal!, one parcel and a moment-major SoA batch ---- */
int co_rhs_coal(const co_params *p, const double *mom, double *dmom, double *scale);
int co_rhs_coal_batch(const co_params *p, long n_parcels, long ld, const double *mom, double *dmom,
                      double *scale /* may be NULL */, int n_threads);
/* rainshaft_helpers.jl:55-78 per-cell sources without the flux divergence */
int co_rainshaft_cell(const co_params *p, const double *mom, double *coal_source, double *sedi_flux);
int co_rainshaft_cell_batch(const co_params *p, long n, long ld, const double *mom, double *coal_source,
                            double *sedi_flux, int n_threads);
int co_update_dist_batch(const co_params *p, long n, long ld, const double *mom, double *params /* [3N][ld] */);
int co_check_moment_consistency(const double *m, int n_moments); /* ParticleDistributions.jl:437-449: 0 ok, 1 / 2 = which check throws */
int co_closure_stats(const co_params *p, long n, long ld, const double *mom, unsigned long long *out /* [4N] */);
int co_max_threads(void);

/* ---- Coalescence.jl (NumericalCoalStyle) with a FIXED Gauss rule in place of quadgk: cloudy_oracle_quad.c ---- */
enum { CO_KF_CONSTANT = 0, CO_KF_LINEAR = 1, CO_KF_HYDRODYNAMIC = 2, CO_KF_LONG = 3 };
/* CoalescenceKernelFunction, KernelFunctions.jl:39-86: p = (rate) | (rate) | (coal_eff) | (x_threshold, below, above) */
typedef struct {
    int kind;
    double p[3];
} co_kernel_func;
double co_kernel_func_eval(const co_kernel_func *kf, double x, double y);                        /* KernelFunctions.jl:94-116 */
int co_get_normalized_kernel_func(const co_kernel_func *kf, const double norms[2], co_kernel_func *out); /* :124-154 */
int co_gauss_gamma_rule(int nq, double k, double *u, double *W);   /* weight u^(k-1) e^-u / Gamma(k) */
int co_gauss_hermite_rule(int nq, double *t, double *W);           /* weight e^(-t^2) / sqrt(pi) */
int co_dist_rule(const co_dist *d, int nq, double *x, double *w);
int co_get_coal_ints_numerical_fixed(const co_dist *pdists, int N, const co_kernel_func *kf, int nq, double *out,
                                     double *scale, double *noise);
/* cloudy_oracle_adaptive.c: the same operator by NESTED ADAPTIVE Gauss-Kronrod quadrature, as the reference evaluates it
 * (Coalescence.jl:503-708) -- slow; generates tests/golden/numerical_adaptive.json.  Q, R: [orders][N][N]; S: [orders][2][N] */
void co_adaptive_set_endpoint_map(int q); /* cloudy_oracle_adaptive.c: 0 (default) = the reference's formulation; q >= 2: end-point map + half-range inner integrals */
int co_get_coal_ints_numerical_adaptive(const co_dist *pdists, int N, const co_kernel_func *kf, double eps_outer,
                                        double eps_inner, double *out, double *Q, double *R, double *S);
/* cloudy_oracle_quad.c, converged mode (CLOUDY_QUAD_CONVERGED): closed forms for Q and R, adaptive Gauss-Kronrod (7, 15)
 * rules for the weighting_fn split (and a Gamma-Lognormal pair's P(Y' < X')); tol = their acceptance tolerance
 * (CO_CONV_TOL in the HIP kernels), q = points per panel of the inner rule of a Lognormal mode's T_m */
#define CO_CONV_TOL 1e-9
double co_inc_beta(double a, double b, double x);
int co_gauss_legendre_rule(int q, double *x, double *w);
void co_conv_range(double A, double *zlo, double *zhi);
long co_conv_node_count(int reset); /* integrand evaluations of the adaptive rules on this thread since the last reset */
void co_conv_set_ln_inner(double sigmas, int max_panels); /* inner rule of a Lognormal mode's T_m: panel width in sigma, cap */
void co_conv_rule_node_counts(long *out); /* ... of the T_m rule of each mode (CO_MAX_MODES entries) in the last call */
int co_get_coal_ints_numerical_converged(const co_dist *pdists, int N, const co_kernel_func *kf, int q, double tol,
                                         double *out, double *scale);
int co_rhs_coal_numerical_converged(const co_params *p, const co_kernel_func *kf_normalized, int q, double tol,
                                    const double *mom, double *dmom, double *scale);
int co_rhs_coal_numerical_converged_batch(const co_params *p, const co_kernel_func *kf_normalized, int q, double tol,
                                          long n_parcels, long ld, const double *mom, double *dmom, double *scale,
                                          int n_threads);
int co_rhs_coal_numerical(const co_params *p, const co_kernel_func *kf_normalized, int nq, const double *mom,
                          double *dmom, double *scale, double *noise);
int co_rhs_coal_numerical_batch(const co_params *p, const co_kernel_func *kf_normalized, int nq, long n_parcels, long ld,
                                const double *mom, double *dmom, double *scale, double *noise, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
