// kernels.hpp -- gfx950 kernels for the batched collision-coalescence moment tendency.
//
// One wavefront lane owns one parcel (two in the all-Inf kernel).  State is moment-major SoA (plane q at
// base + q*ld), so a wave's load of one moment is one contiguous 512-byte (1-KiB) request.  All parcel-independent
// constants (normalised kernel tensors, norms, thresholds, Simpson nodes) are wave-uniform: compile-time constants
// when the kernel is compiled for its plan (jit.hpp), otherwise kernel arguments / uniform-address loads that live
// in SGPRs, the CDNA home for broadcast data (no LDS traffic, no per-lane registers).  LDS carries what crosses
// lanes: the regime ranking and parcel exchange of the threshold kernels, the flux exchange of the rainshaft column.
//
// The per-parcel chain fused here (reference file:line, CliMA/Cloudy.jl v0.6.0):
//   normalise            test/examples/utils/box_model_helpers.jl:30-31
//   closure inversion    src/ParticleDistributions/ParticleDistributions.jl:456-476 (Gamma), :512-523 (Exp)
//   moments matrix       src/Sources/Coalescence.jl:187-198   (integer orders via M_q = M_{q-1} theta (k+q-1))
//   finite 2-D integrals src/Sources/Coalescence.jl:200-244, ParticleDistributions.jl:567-612, :698-710
//   Q, R, S and assembly src/Sources/Coalescence.jl:115-185, :260-455
//   de-normalise         test/examples/utils/box_model_helpers.jl:52
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.hpp"

namespace cloudy {

enum { IN_MOMENTS = 0, IN_PARAMS = 1 };
enum { MODE_ALLINF = 0, MODE_FIXED = 1, MODE_MOVING = 2 };
enum { DIST_EXP = 0, DIST_GAMMA = 1, DIST_MONO = 2, DIST_LOGNORMAL = 3 };
// terms and radius (t <= kEarlyTmax, u (a_top - 1) <= kEarlyUa) of the early-node series of the fp64 pass (see msh_grid);
// the macros exist for timing / accuracy experiments through CLOUDY_HIP_JIT_DEFS
#ifndef CLOUDY_EARLY_SERIES
#define CLOUDY_EARLY_SERIES 30
#define CLOUDY_EARLY_TMAX 3.0
#define CLOUDY_EARLY_UA 1.5
#endif
constexpr int kEarlySeries = CLOUDY_EARLY_SERIES;
constexpr double kEarlyTmax = CLOUDY_EARLY_TMAX, kEarlyUa = CLOUDY_EARLY_UA;
constexpr int kNodeStride = 5;  // x, ln x, x_t - x, ln(x_t - x), w * dx  (the early nodes need no table: msh_grid)
constexpr int kBlock = 256;
constexpr int kInvTerms = 16;  // coefficients of the start-value polynomial of the percentile threshold (moving_threshold)

template <int N, int P>
struct KArgs {
    int32_t dist_type[N], np[N], off[N];
    int32_t finite[N];                 // FIXED: 1 if mode i < N-1 has a finite threshold
    int32_t node_off[N], n_bins[N];    // FIXED: slice of the node table of mode i
    int32_t n_2d[N];
    int32_t n_mom_max, input_kind, rainshaft, nbpl;
    double thr[N];                     // FIXED: threshold / m0; MOVING: percentile
    double norm[3 * N], inv_norm[3 * N], out_scale[3 * N];
    double kmin, kmax;
    double c[N][N][P][P];              // normalised tensors, c[j][k][a][b]
    // MOVING: start value of the percentile inversion x = P^-1(k; percentile_i) as a polynomial in t = inv_map[0] +
    // inv_map[1] ln k (highest power first; ln x = poly(t)) for k in [inv_klo, kmax], built at plan creation from the
    // plan's percentiles (host_gamma.hpp); inv_map[1] == 0: no table (percentile 0 or 1, or a k range it cannot cover)
    double inv_map[2], inv_klo;
    double inv_tab[N][kInvTerms];
};

// Streaming accesses: every moment is read once and every tendency written once per launch, so they carry the
// nontemporal hint (measured with tools/hbm_ceiling.hip on MI355X: 6-plane copy 5.6 -> 6.0 TB/s).
typedef double dvec2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double ld_stream(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ dvec2 ld_stream2(const double *p) {
    return __builtin_nontemporal_load(reinterpret_cast<const dvec2 *>(p));
}
__device__ __forceinline__ void st_stream2(double *p, double a, double b) {
    dvec2 v;
    v.x = a;
    v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<dvec2 *>(p));
}
__device__ __forceinline__ void st_stream(double *p, double v) { __builtin_nontemporal_store(v, p); }
// CLOUDY_F32 plans: planes are float in HBM (half the traffic), arithmetic stays fp64 in registers
typedef float fvec2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double ld_stream(const float *p) { return (double)__builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(float *p, double v) { __builtin_nontemporal_store((float)v, p); }
__device__ __forceinline__ dvec2 ld_stream2(const float *p) {
    const fvec2 v = __builtin_nontemporal_load(reinterpret_cast<const fvec2 *>(p));
    dvec2 r;
    r.x = (double)v.x;
    r.y = (double)v.y;
    return r;
}
__device__ __forceinline__ void st_stream2(float *p, double a, double b) {
    fvec2 v;
    v.x = (float)a;
    v.y = (float)b;
    __builtin_nontemporal_store(v, reinterpret_cast<fvec2 *>(p));
}

template <int M>
__host__ __device__ constexpr int tri(int p, int q) {  // packed upper triangle, p <= q < M
    return p * M - (p * (p - 1)) / 2 + (q - p);
}
template <int M>
__host__ __device__ constexpr int trisym(int p, int q) {
    return p <= q ? tri<M>(p, q) : tri<M>(q, p);
}

// x / d for a wave-uniform divisor d with r = RN(1/d): q = RN(x r), e = x - q d (exact by FMA),
// RN(q + e r) is the correctly rounded quotient (Markstein's correction step) -- 3 VALU ops, no v_div_*.
__device__ __forceinline__ double div_by_const(double x, double d, double r) {
    const double q = x * r;
    // Inf / NaN of the first product must survive: there the residual is NaN, and v_min_f64 (minNum: the non-NaN operand)
    // turns it into a finite number, so the FMA returns q itself -- one instruction instead of a compare and two selects.
    // A FINITE x whose quotient overflows (q = +-Inf, residual = -+Inf) needs the clamp on the other side as well: the
    // FMA then adds a finite number to q and returns the Inf Julia's x / d gives (without it: Inf - Inf = NaN).
    const double e = __builtin_fmax(__builtin_fmin(fma(-q, d, x), 1.7976931348623157e308), -1.7976931348623157e308);
    return fma(e, r, q);
}

// x / d for a divisor in the normal range (eps < d < 1e150, the guarded moments and the clamped shape of the closure
// inversion): the reciprocal refinement, quotient, exact residual and final FMA of the IEEE division sequence without its
// v_div_scale / v_div_fixup range handling -- the same bits as x / d there (the final FMA rounds the same exact value),
// 8 instead of 11 instructions.  A non-finite numerator gives NaN where IEEE gives Inf.
__device__ __forceinline__ double div_normal(double x, double d) {
    const double r = recip_fast(d);
    const double q = x * r;
    return fma(fma(-d, q, x), r, q);
}

// update_dist_from_moments, ParticleDistributions.jl:456-476 / :512-523 (normalised moments in)
__device__ __forceinline__ void invert_closure(int dist_type, double m0, double m1, double m2, double kmin,
                                               double kmax, double &n, double &th, double &k) {
    if (dist_type == DIST_LOGNORMAL) {
        // ParticleDistributions.jl:483-505: (n, mu, sigma) kept in the (n, theta, k) slots
        if (m0 > kEps && m1 > kEps && m2 > kEps) {
            const double mu = log((m1 * m1) / pow(m0, 1.5) / pow(m2, 0.5));
            const double sg0 = sqrt(log(m0 * m2 / (m1 * m1)));
            // max(eps, min(Inf, .)); moments with M0 M2 < M1^2 make Julia's sqrt throw a DomainError -- a batch cannot
            // throw per parcel, so they take the sigma = eps clamp as well (the oracle does the same)
            const double sg = (sg0 > kEps) ? sg0 : kEps;
            th = mu;
            k = sg;
            n = m1 / exp(mu + 0.5 * (sg * sg));
        } else {
            n = 0.0;
            th = 1.0;
            k = 1.0;
        }
        return;
    }
    if (m0 > kEps && m1 > kEps) {
        n = m0;
        const double mean = div_normal(m1, m0);
        if (dist_type == DIST_GAMMA) {
            // (the second quotient keeps the IEEE sequence: its divisor may be zero or subnormal -> k = kmax / kmin)
            double kk = mean / (div_normal(m2, m1) - mean);
            // max(kmin, min(kmax, kk)) with Julia's NaN-propagating min/max
            // (v_min / v_max drop a NaN operand, so the NaN of 0/0 is put back by one select)
            const double clamped = __builtin_fmax(__builtin_fmin(kk, kmax), kmin);
            k = (kk != kk) ? kk : clamped;
            th = div_normal(mean, k);
        } else {  // Exponential :512-523, Monodisperse :530-541
            k = 1.0;
            th = mean;
        }
    } else {
        n = 0.0;
        th = 1.0;
        k = 1.0;
    }
}

// get_moments_matrix row, Coalescence.jl:187-198, columns >= N_mom_max zero.  Integer-order moments by recurrence:
//   Gamma / Exponential (k = 1):  M_q = n theta^q Gamma(q+k)/Gamma(k)  ->  M_q = M_{q-1} theta (k + q - 1)
//   Monodisperse:                 M_q = n theta^q                      ->  M_q = M_{q-1} theta
//   Lognormal (theta = mu, k = sigma): M_q = n exp(q mu + q^2 sigma^2/2) -> M_{q+1}/M_q = e^(mu + sigma^2/2) (e^(sigma^2))^q
template <int M>
__device__ __forceinline__ void moment_row(int dist_type, double n, double th, double k, int n_mom_max, double (&Mk)[M]) {
    Mk[0] = n;
    if (dist_type == DIST_LOGNORMAL) {
        double ratio = exp(th + 0.5 * (k * k));
        const double r = exp(k * k);
#pragma unroll
        for (int q = 1; q < M; ++q) {
            Mk[q] = Mk[q - 1] * ratio;
            ratio *= r;
        }
    } else {
        const double base = (dist_type == DIST_MONO) ? 0.0 : k;  // Monodisperse: multiplier theta * 1
        const double step = (dist_type == DIST_MONO) ? 0.0 : 1.0;
#pragma unroll
        for (int q = 1; q < M; ++q)
            Mk[q] = Mk[q - 1] * (th * ((dist_type == DIST_MONO) ? 1.0 : (base + step * double(q - 1))));
    }
    if (n_mom_max < M) {
#pragma unroll
        for (int q = 0; q < M; ++q)
            if (q >= n_mom_max) Mk[q] = 0.0;
    }
}

// Simpson end weights of integrate_SimpsonEvenFast (ParticleDistributions.jl:698-710) as a weight per node
__host__ __device__ inline double simpson_weight(int j /*1-based*/, int n_bins) {
    const int e = n_bins + 1;
    double w = (j >= 5 && j <= n_bins - 3) ? 1.0 : 0.0;
    if (j == 1 || j == e) w += 17.0 / 48.0;
    if (j == 2 || j == e - 1) w += 59.0 / 48.0;
    if (j == 3 || j == e - 2) w += 43.0 / 48.0;
    if (j == 4 || j == e - 3) w += 49.0 / 48.0;
    return w;
}

// the same for the per-parcel grids of the moving threshold, where it is evaluated per node: every node but the first
// four and the last three visited ones has weight 1 (n_bins >= 8: the end corrections do not overlap)
__device__ __forceinline__ double simpson_weight_node(int j /*1-based*/, int n_bins) {
    if (n_bins < 8) return simpson_weight(j, n_bins);
    const int lo = j - 1, hi = n_bins + 1 - j;  // distance to either end point (the end point e = n_bins + 1 is never visited)
    const int d = lo < hi ? lo : hi;
    double w = 1.0;
    w = d == 3 ? 49.0 / 48.0 : w;
    w = d == 2 ? 43.0 / 48.0 : w;
    w = d == 1 ? 59.0 / 48.0 : w;
    w = d == 0 ? 17.0 / 48.0 : w;
    return w;
}

// One Simpson node of the log-uniform grid of moment_source_helper (ParticleDistributions.jl:604-610).
struct SimpsonNode {
    double x, lx, xmx, lxmx, wdx;  // x_j, ln x_j, x_t - x_j, ln(x_t - x_j), w_j * dx
};
// FixedThreshold: the grid depends on the plan only -> node table built on the host, read at uniform addresses.
struct FixedGrid {
    const double *__restrict__ nd;
    int nb;
    __device__ __forceinline__ int n_bins() const { return nb; }
    __device__ __forceinline__ double node_x(int j) const { return nd[kNodeStride * j]; }
    // running-abscissa interface shared with MovingGrid (a table needs none)
    __device__ __forceinline__ double first_x() const { return nd[0]; }
    __device__ __forceinline__ double first_lx() const { return nd[1]; }
    __device__ __forceinline__ double log_step() const { return nd[kNodeStride + 1] - nd[1]; }  // dx of the log-uniform grid
    __device__ __forceinline__ double next_x(double, int, int) const { return 0.0; }
    __device__ __forceinline__ double last_x() const { return 0.0; }
    __device__ __forceinline__ double prev_x(double, int, int) const { return 0.0; }
    __device__ __forceinline__ double node_x(int j, double) const { return nd[kNodeStride * j]; }
    __device__ __forceinline__ SimpsonNode node(int j, double, bool late) const { return node(j, late); }
    __device__ __forceinline__ SimpsonNode node(int j, bool /*late*/) const {
        return SimpsonNode{nd[kNodeStride * j + 0], nd[kNodeStride * j + 1], nd[kNodeStride * j + 2],
                           nd[kNodeStride * j + 3], nd[kNodeStride * j + 4]};
    }
};
// MovingThreshold: the threshold, hence the grid, is per parcel (computed with the reference's expressions).
struct MovingGrid {
    double xt, x_min, dx, ratio, inv_ratio;
    int nb;
    __device__ __forceinline__ MovingGrid(double xt_, int nbpl) : xt(xt_) {
        const double x_lb = fmin(1e-5, 1e-5 * xt_);
        nb = (int)floor(double(nbpl) * log10(xt_ / x_lb));
        x_min = log_pos(x_lb);
        dx = (log_pos(xt_) - x_min) / double(nb);
        ratio = exp_fin(dx);
        inv_ratio = exp_fin(-dx);
    }
    __device__ __forceinline__ int n_bins() const { return nb; }
    __device__ __forceinline__ double node_x(int j) const { return exp(x_min + double(j) * dx); }
    // The nodes are visited in order, so x_j = exp(x_min + j dx) (ParticleDistributions.jl:566) is carried along as
    // x_{j-1} e^{dx} and re-anchored with a true exp() every 16th node: at most 15 roundings of drift (< 2e-15), one
    // multiplication instead of ~31 instructions for the others.
    __device__ __forceinline__ double first_x() const { return exp_fin(x_min); }
    __device__ __forceinline__ double first_lx() const { return x_min; }
    __device__ __forceinline__ double log_step() const { return dx; }
    // `phase` is a wave-uniform iteration counter: lanes sit at different nodes j of their own grids, and an anchor keyed
    // on j would make every lane's exp() run (masked) in almost every iteration of the wave
    __device__ __forceinline__ double next_x(double x_run, int j_next, int phase) const {
        return (phase & 15) == 15 ? exp_fin(x_min + double(j_next) * dx) : x_run * ratio;
    }
    // (the late nodes are visited from the last one down)
    __device__ __forceinline__ double last_x() const { return exp_fin(x_min + double(nb - 1) * dx); }
    __device__ __forceinline__ double prev_x(double x_run, int j_prev, int phase) const {
        return (phase & 15) == 15 ? exp_fin(x_min + double(j_prev) * dx) : x_run * inv_ratio;
    }
    __device__ __forceinline__ double node_x(int, double x_run) const { return x_run; }
    __device__ __forceinline__ SimpsonNode node(int j, double x_run, bool late) const {
        SimpsonNode s;
        s.lx = x_min + double(j) * dx;  // logx(x_min, j+1, dx)
        s.x = x_run;
        s.xmx = xt - s.x;
        s.lxmx = late ? log_pos(s.xmx) : 0.0;  // (x_t - x > 0 on the reference grid; a node at x >= x_t is masked by the caller)
        s.wdx = simpson_weight_node(j + 1, nb) * dx;
        return s;
    }
};

// ---- The early nodes of one Simpson grid: no loop over them at all.
// With u = x / x_t (t = z0 u) the integrand of one order a = k + p2 is
//   u^(k+p1) G_a(u),   G_a(u) = e^{-z0 u} P(a, z0 (1 - u)) = sum_s d_s u^s,
// and G_a has a two-term recurrence, because the exponentials cancel in its derivative:
//   G_a' = -z0 G_a - z0 C_a (1 - u)^(a-1),  C_a = z0^(a-1) e^-z0 / Gamma(a)
//   =>  d_0 = P(a, z0),  d_{s+1} = -z0 (d_s + C_a beta_s) / (s + 1),  (1 - u)^(a-1) = sum_s beta_s u^s,
// the orders are linked by P(a-1, z) = P(a, z) + z^(a-1) e^-z / Gamma(a):  G_{a-1} = G_a + C_a (1 - u)^(a-1),
// and the early nodes are a geometric progression u_j = u_0 rho^j, rho = e^dx (ParticleDistributions.jl:566,
// 604-610) with Simpson weights 1 but for the first four (:698-710), so their power sums are closed forms:
//   V_s = sum_{j<J} w_j dx u_j^(k+s) = u_J^(k+s) W_s,   W_s = dx [ (1 - b_s)/(r_s - 1) - b_s c(r_s) ],
//   r_s = rho^(k+s),  b_s = r_s^-J,  c(r) = 31/48 - 11/48 r + 5/48 r^2 - 1/48 r^3   (u_J: the first node NOT early).
// Hence  sum_{j<J} (w_j dx) x_j^p1 t_j^k e^{-t_j} P(a, z0 - t_j) = t_J^k x_J^p1 sum_s (d_s u_J^s) W_{p1+s}:
// kEarlySeries terms of ~48 operations for all orders and all p1 together, whatever J (45-110 nodes), one
// incomplete gamma (at z0) for the whole group, every lane the same work.  Radius t <= 3 and u (a_top - 1) <= 1.5
// (u <= 1/2): with 30 terms the truncation is below the rounding floor of the sum (~5e-14 relative to the early
// sum, set by the roundings of k ln t_J and a ln z0 and by the alternating e^-t terms, e^(2t) eps), checked against
// mpmath over k in [1e-9, 10], z0 in [1e-3, 300]; measured on the GPU: 24 terms at (2, 1) cfg3b 2.59 ms, 30 terms at
// (3, 1.5) 2.39 ms, 32 terms at (3.5, 1.5) 2.42 ms, same worst parity error (2.6e-14 of scale over 40 random plans).
// Everything is kept relative to u_J^k (t_J ~ 1): nothing over- or underflows for clamped closures.  (Round 1 expanded P alone about z0 and summed the nodes one by one -- one exp and 21 FMAs per
// node; the first closed form of this round expanded e^-t and P separately: 16 x 19 + 5 x 16 x 8 operations and a
// fifth of the radius.)
template <int P, typename Grid>
__device__ __forceinline__ int early_series(const Grid &grid, int nb, double xt, double th, double k, double inv_th,
                                            double lnth, double a_top, double lg_top, double z0,
                                            double (&acc)[(P + 2) * (P + 3) / 2]) {  // returns J: nodes 0 .. J-1 are done
    constexpr int M = P + 2;
    const double x_early = fmin(kEarlyTmax * th, kEarlyUa * xt / fmax(a_top - 1.0, 3.0));
    int j = 0;
    const double xr = grid.first_x();
    {
        const double lx0 = grid.first_lx(), dxl = grid.log_step();
        int J = 0;
        if (x_early >= xr) {  // xr = the first node
            const double jf = floor((log_pos(x_early) - lx0) / dxl) + 1.0;  // nodes with x_j <= x_early
            // never the last three nodes, whose weights differ (u <= 1/2 keeps 4.5 nodes of distance on the reference's
            // 15-per-decade grid; a coarser grid gives up a node or two of the group)
            J = jf < double(nb - 4) ? (int)jf : nb - 4;
        }
        if (J < 4) J = 0;  // fewer than four early nodes (the end correction assumes the first four): none
        if (J > 0) {
            const double Jd = double(J);
            const double xJ = exp_fin(fma(Jd, dxl, lx0));  // node J itself, as the reference computes it
            const double uJ = xJ * (1.0 / xt), tJ = xJ * inv_th;
            const double rho = exp_fin(dxl), rhoJinv = exp_fin(-Jd * dxl);
            double r = exp_fin(k * dxl), b = exp_fin(-Jd * k * dxl);
            // P(a_top, z0), and C_a B_s for the M - 1 orders a = a_top ... k + 1  (B_s = beta_s u_J^s of exponent a - 1)
            const double invz0 = 1.0 / z0;
            const double Etop = exp_fin(fma(a_top, log_pos(xt) - lnth, -z0) - lg_top);  // z0^a e^-z0 / Gamma(a+1)
            double e_top = inc_gamma_p_from_E(a_top, z0, Etop, nullptr);
            double CB[M > 1 ? M - 1 : 1], al[M > 1 ? M - 1 : 1];
            {
                double a = a_top, g = Etop;
#pragma unroll
                for (int i = 0; i < M - 1; ++i) {
                    g *= a * invz0;  // C_a = E(a-1, z0)
                    CB[i] = g;
                    al[i] = a - 1.0;
                    a -= 1.0;
                }
            }
            double win[M];  // W_s ... W_{s+M-1}
            const auto next_W = [&](bool first) {
                // s = 0: k dx can be tiny (clamped closures): the two differences through expm1
                const double omb = first ? -expm1(-Jd * k * dxl) : 1.0 - b;
                const double rm1 = first ? expm1(k * dxl) : r - 1.0;
                const double c = fma(r, fma(r, fma(r, -1.0 / 48.0, 5.0 / 48.0), -11.0 / 48.0), 31.0 / 48.0);
                const double W = dxl * fma(omb, recip_fast(rm1), -(b * c));
                r *= rho;
                b *= rhoJinv;
                return W;
            };
#pragma unroll
            for (int s = 0; s < M - 1; ++s) win[s] = next_W(s == 0);
            // (unrolled for the small orders; rolled for M > 5, where the scheduler otherwise hoists the independent W
            // chain over the whole body and spills hundreds of registers -- 480 fully unrolled, 76 / 122 unrolled by 2 / 3)
            constexpr int kUnrollSeries = M <= 5 ? kEarlySeries : 1;
            double nd = 0.0;  // n as a double
#pragma unroll kUnrollSeries
            for (int n = 0; n < kEarlySeries; ++n) {
                win[M - 1] = next_W(false);  // this step adds (d_n u_J^n) W_{p1+n}
                double e = e_top;
#pragma unroll
                for (int p2 = M - 1; p2 >= 0; --p2) {
#pragma unroll
                    for (int p1 = 0; p1 <= p2; ++p1) acc[tri<M>(p1, p2)] = fma(e, win[p1], acc[tri<M>(p1, p2)]);
                    if (p2 > 0) e += CB[M - 1 - p2];
                }
                const double inv_n1 = M <= 5 ? 1.0 / double(n + 1) : recip_fast(nd + 1.0);
                e_top = (e_top + CB[0]) * (-tJ * inv_n1);
                const double f = uJ * inv_n1;
#pragma unroll
                for (int i = 0; i < M - 1; ++i) CB[i] *= (nd - al[i]) * f;
#pragma unroll
                for (int i = 0; i < M - 1; ++i) win[i] = win[i + 1];
                nd += 1.0;
            }
            double f = exp_fin(k * log_pos(tJ));  // t_J^k x_J^p1
#pragma unroll
            for (int p1 = 0; p1 < M; ++p1) {
#pragma unroll
                for (int p2 = p1; p2 < M; ++p2) acc[tri<M>(p1, p2)] *= f;
                f *= xJ;
            }
            j = J;
        }
    }
    return j;
}

// moment_source_helper for all (p1 <= p2) of one mode in ONE pass over its Simpson grid:
//   msh[p1][p2] = n M_p2 / Gamma(k) * sum_j (w_j dx) x_j^p1 t_j^k e^{-t_j} P(k + p2, z_j),
//   t_j = x_j / theta,  z_j = (x_t - x_j) / theta = z0 - t_j,  z0 = x_t / theta
// which is ParticleDistributions.jl:589-612 (Gamma) / :567-587 (Exponential, k = 1) regrouped.
//
// Late nodes: one incomplete-gamma evaluation at the top order a_top = k + M - 1 per node, every lower order
// by the stable downward recurrence P(a-1, z) = P(a, z) + z^(a-1) e^-z / Gamma(a).
//
// The early nodes go through early_series (closed form), the others get one incomplete-gamma evaluation each.
template <int P, typename Grid>
__device__ __forceinline__ void msh_grid(const Grid &grid, double xt, double th, double k, bool is_gamma,
                                         double (&msh)[(P + 2) * (P + 3) / 2]) {  // WITHOUT msh_pref(n, k) M_p2: the caller applies them
    constexpr int M = P + 2;
    constexpr int T = M * (M + 1) / 2;
    const int nb = grid.n_bins();
    // (log_pos / lgamma_pos of device_math.hpp: branch-free, a third of the library routines' instructions)
    const double inv_th = 1.0 / th, lnth = log_pos(th);
    const double a_top = k + double(M - 1);
    const double lg_top = lgamma_pos(a_top + 1.0);
    const double z0 = xt * inv_th;
    double acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.0;

    // ---- early nodes: closed form (early_series)
    int j = early_series<P>(grid, nb, xt, th, k, inv_th, lnth, a_top, lg_top, z0, acc);
    // ---- late nodes: one incomplete-gamma evaluation each
    // From the LAST node down: in a regime-sorted wave every lane is then at the same node index in the same iteration
    // (the lanes differ in where their early group ends, not in where the grid ends), so z = (x_t - x_j) / theta is as
    // homogeneous across the wave as theta is -- and the node table of a FixedGrid is read at wave-uniform addresses.
    const int j_first = j;
    double xr = grid.last_x();  // running abscissa (MovingGrid); a table (FixedGrid) ignores it
#pragma unroll 1
    for (int it = 0, jj = nb - 1; jj >= j_first; --jj, ++it) {
        const SimpsonNode nd = grid.node(jj, xr, true);
        xr = grid.prev_x(xr, jj - 1, it);
        const double t = nd.x * inv_th, zr = nd.xmx * inv_th;
        const bool zpos = zr > 0.0;  // P(a, z <= 0) = 0: such a node (never on the reference grid) contributes nothing
        const double z = zpos ? zr : 1.0;
        const double h0 = zpos ? nd.wdx * exp_fin(fma(k, nd.lx - lnth, -t)) : 0.0;
        const double E0 = exp_fin(fma(a_top, nd.lxmx - lnth, -z) - lg_top);
        double Pz[M];
        Pz[M - 1] = inc_gamma_p_from_E(a_top, z, E0, nullptr);
        const double invz = recip_fast(z);
        double E = E0, a = a_top;
#pragma unroll
        for (int p2 = M - 2; p2 >= 0; --p2) {
            E *= a * invz;
            a -= 1.0;
            Pz[p2] = Pz[p2 + 1] + E;
        }
        double h = h0;
#pragma unroll
        for (int p1 = 0; p1 < M; ++p1) {
#pragma unroll
            for (int p2 = p1; p2 < M; ++p2) acc[tri<M>(p1, p2)] = fma(h, Pz[p2], acc[tri<M>(p1, p2)]);
            h *= nd.x;
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) msh[t] = acc[t];
}

// CLOUDY_F32_FAST plans: the same Simpson pass with single-precision arithmetic for everything that is done per late
// node (exp via v_exp_f32, series / continued fraction to ~1e-7, fp32 accumulators); the exponents of the two exp()
// calls are still formed in fp64 (they are differences of O(100) terms).  The early nodes take the fp64 closed form
// (early_series): it is cheaper than any loop over them.  Everything outside this function (closure inversion, moments,
// F/min, Q/R/S) stays fp64.  Expected accuracy of the msh entries ~1e-6 relative; tests report the error against the
// fp64 oracle.
template <int P, typename Grid>
__device__ __forceinline__ void msh_grid_f32(const Grid &grid, double xt, double th, double k, bool is_gamma,
                                             double (&msh)[(P + 2) * (P + 3) / 2]) {  // WITHOUT the factor M_p2
    constexpr int M = P + 2;
    constexpr int T = M * (M + 1) / 2;
    const int nb = grid.n_bins();
    const double inv_th = 1.0 / th, lnth = log_pos(th);
    const double a_top = k + double(M - 1);
    const double lg_top = lgamma_pos(a_top + 1.0);
    const double z0 = xt * inv_th;
    const float a_topf = (float)a_top;
    // early nodes: the fp64 closed form (early_series) -- it costs less than a single-precision loop over the nodes
    double acc_early[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc_early[t] = 0.0;
    const int j = early_series<P>(grid, nb, xt, th, k, inv_th, lnth, a_top, lg_top, z0, acc_early);
    float acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.0f;
    // From the LAST node down: in a regime-sorted wave every lane is then at the same node index in the same iteration
    // (the lanes differ in where their early group ends, not in where the grid ends), so z = (x_t - x_j) / theta is as
    // homogeneous across the wave as theta is -- and the node table of a FixedGrid is read at wave-uniform addresses.
    const int j_first = j;
    double xr = grid.last_x();  // running abscissa (MovingGrid); a table (FixedGrid) ignores it
#pragma unroll 1
    for (int it = 0, jj = nb - 1; jj >= j_first; --jj, ++it) {
        const SimpsonNode nd = grid.node(jj, xr, true);
        xr = grid.prev_x(xr, jj - 1, it);
        const double td = nd.x * inv_th, zd = nd.xmx * inv_th;
        if (!(zd > 0.0)) continue;
        const float z = (float)zd, xf = (float)nd.x;
        const float h0 = (float)nd.wdx * __expf((float)fma(k, nd.lx - lnth, -td));
        const float E0 = __expf((float)(fma(a_top, nd.lxmx - lnth, -zd) - lg_top));
        float Pz[M];
        Pz[M - 1] = inc_gamma_p_from_E_f32(a_topf, z, E0);
        const float invz = 1.0f / z;
        float E = E0, a = a_topf;
#pragma unroll
        for (int p2 = M - 2; p2 >= 0; --p2) {
            E *= a * invz;
            a -= 1.0f;
            Pz[p2] = Pz[p2 + 1] + E;
        }
        float h = h0;
#pragma unroll
        for (int p1 = 0; p1 < M; ++p1) {
#pragma unroll
            for (int p2 = p1; p2 < M; ++p2) acc[tri<M>(p1, p2)] = fmaf(h, Pz[p2], acc[tri<M>(p1, p2)]);
            h *= xf;
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) msh[t] = acc_early[t] + (double)acc[t];
}

// moment_source_helper(::Lognormal...), ParticleDistributions.jl:614-625 -- the reference nests adaptive quadgk:
//   int_0^xt y^p2 n f(y) [ int_0^(xt - y) x^p1 n f(x) dx ] dy = M_p1 M_p2 Prob_{p1,p2}(X + Y < xt)
// (X, Y drawn from the size-biased laws x^p f(x) / M_p, which are Lognormal(mu + p sigma^2, sigma) again).  The inner
// integral is a closed form, Phi((ln(xt - y) - mu - p1 sigma^2) / sigma); the outer one is a kLnNodes-point midpoint rule
// in v, y = xt / (1 + e^-v): in v the integrand is analytic and decays like a Gaussian on both sides (ln y ~ v for
// v << 0, ln(xt - y) ~ -v for v >> 0), so the equispaced rule converges geometrically -- measured against adaptive
// quadrature of the reference integrand over sigma in [0.005, 2], thresholds at 1.8 ... 2.5 e^mu among them
// (tests/golden/lognormal_adaptive.json; tests/test_oracle_kats.py states the bound).
// One pass serves all (p1 <= p2): per node one Gaussian, its ratios for the other p2 (M_q ratio recurrence), and M
// values of Phi.  Every lane does identical work (no data-dependent trip counts).  theta = mu, k = sigma.
// Output convention of msh_grid: WITHOUT the factors n and M_p2 the caller applies: msh[p1][p2] = Prob * M_p1 / n.
constexpr int kLnNodes = 48;
constexpr double kLnSigmaFloor = 1e-6;  // below: the point-mass limit (msh_lognormal); the rule alone is off by 8e-12 at 1e-6, 5e-9 at 1.5e-8
__device__ __forceinline__ double norm_cdf(double z) { return 0.5 * erfc(-z * 0.70710678118654752440); }
__device__ __forceinline__ double softplus_pos(double x) {  // ln(1 + e^x)
    return x > 0.0 ? x + log1p(exp_fin(-x)) : log1p(exp_fin(x));
}
template <int P>
__device__ __forceinline__ void msh_lognormal(double xt, double mu, double sg, double (&msh)[(P + 2) * (P + 3) / 2]) {
    constexpr int M = P + 2;
    constexpr int T = M * (M + 1) / 2;
    const double lxt = log_pos(xt), inv_sg = 1.0 / sg, s2 = sg * sg;
    // The window of v in which the integrand lives, through the exact map v(ln y) = d - ln(1 - e^d), d = ln y - ln xt < 0
    // (round 4): 8.5 sigma of the density of order 0 on the left; on the right the end of the density of the top order or
    // the decay of the inner Phi (ln(xt - y) = mu - 8.5 sigma: v = ln(e^delta - 1)), whichever comes first.  ~17 sigma of
    // ln y whatever ln xt - mu is, so the node spacing scales with sigma (rounds 2-3 took v = ln y - ln xt and v = delta,
    // true only far from v = 0: for x_t ~ 2 e^mu and sigma <= 0.03 most nodes fell outside the window -- 2.6e-6 at
    // sigma = 0.01).  Empty window: the integral is below e^-36 M_p1 M_p2 -> h = 0.
    double vlo = 0.0, vhi = 0.0;
    {
        const double d_lo = fma(-8.5, sg, mu) - lxt, d_hi = (mu + double(M - 1) * s2 + 8.5 * sg) - lxt;
        const double delta = (lxt - mu) + 8.5 * sg;
        if (d_lo < -1e-9 && delta > 1e-9) {
            vlo = d_lo - log(-expm1(d_lo));
            vhi = delta > 36.0 ? delta : log(expm1(delta));
            if (d_hi < -1e-9) vhi = fmin(vhi, d_hi - log(-expm1(d_hi)));
            vhi = fmax(vhi, vlo);
        }
    }
    const double h = (vhi - vlo) * (1.0 / double(kLnNodes));
    double acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.0;
    const double c0 = h * inv_sg * 0.3989422804014327;  // h / (sigma sqrt(2 pi))
    const double er = exp_fin(-s2);                       // G_{q+1} / G_q = e^(u sigma - sigma^2 / 2) (e^(-sigma^2))^q, u of q = 0
#pragma unroll 1
    for (int j = 0; j < kLnNodes; ++j) {
        const double v = fma(h, double(j) + 0.5, vlo);
        const double ly = lxt - softplus_pos(-v), lzr = -softplus_pos(v);  // ln y, ln(xt - y) - ln xt
        const double u = (ly - mu) * inv_sg;
        double G = c0 * exp_fin(fma(-0.5 * u, u, lzr));                    // h G_0(ln y) (1 - y / xt)
        double ratio = exp_fin(fma(u, sg, -0.5 * s2));
        const double w0 = (lzr + lxt - mu) * inv_sg;
        double Ph[M];
#pragma unroll
        for (int p1 = 0; p1 < M; ++p1) Ph[p1] = norm_cdf(w0 - double(p1) * sg);
#pragma unroll
        for (int p2 = 0; p2 < M; ++p2) {
#pragma unroll
            for (int p1 = 0; p1 <= p2; ++p1) acc[tri<M>(p1, p2)] = fma(G, Ph[p1], acc[tri<M>(p1, p2)]);
            G *= ratio;
            ratio *= er;
        }
    }
    // A closure clamped to sigma = eps (moments with M0 M2 < M1^2: Julia's sqrt throws, a batch cannot) is a point mass
    // at e^mu: Prob(X + Y < xt) = [2 e^mu < xt].  Below kLnSigmaFloor the rule's own arithmetic (ln y - mu over sigma)
    // is rounding noise, so the limit is taken instead.
    if (sg < kLnSigmaFloor) {
        const double step = (mu + 0.6931471805599453 < lxt) ? 1.0 : 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = step;
    }
    // M_p1 / n = exp(p1 mu + p1^2 sigma^2 / 2), by the ratio recurrence of moment_row
    double mp = 1.0, rr = exp(mu + 0.5 * s2);
    const double r2 = exp(s2);
#pragma unroll
    for (int p1 = 0; p1 < M; ++p1) {
#pragma unroll
        for (int p2 = p1; p2 < M; ++p2) msh[tri<M>(p1, p2)] = acc[tri<M>(p1, p2)] * mp;
        mp *= rr;
        rr *= r2;
    }
}

// compute_threshold, ParticleDistributions.jl:747-761.  The percentile is a plan constant, so gamma_inc_inv(k, p, 1 - p)
// is a function of k alone: a polynomial in ln k staged with the plan gives it to ~1e-7, and the safeguarded Halley
// iteration of inc_gamma_inv then converges in one or two steps instead of five to eight from the generic start value
// (measured on the 4-mode moving example: the inversions were 3 ms of an 8.2 ms launch).
template <int N, int P>
__device__ __forceinline__ double moving_threshold(const KArgs<N, P> &A, int i, bool is_gamma, double th, double k) {
    const double minx = 1e-18;
    const double percentile = A.thr[i];
    if (!is_gamma) return fmax(-th * log(1.0 - percentile), minx);
    double x0 = 0.0;  // 0: the generic start value
    if (A.inv_map[1] != 0.0 && k >= A.inv_klo) {
        const double t = fma(log_pos(k), A.inv_map[1], A.inv_map[0]);
        double y = A.inv_tab[i][0];
#pragma unroll
        for (int d = 1; d < kInvTerms; ++d) y = fma(y, t, A.inv_tab[i][d]);
        x0 = exp_fin(y);
    }
    return fmax(th * inc_gamma_inv(k, percentile, 1.0 - percentile, x0), minx);
}

// get_finite_2d_integrals entries of one mode (Coalescence.jl:213-227, upper triangle) and the
// "promoted part" D[p][q] = M_p M_q - F[p][q], so that S_1k = 1/2 sum c C (MM - D), S_2k = 1/2 sum c C D.
template <int P>
__device__ __forceinline__ void finite_2d_and_promoted(const double (&Mk)[P + 2], bool thresholded,
                                                       const double (&msh)[(P + 2) * (P + 3) / 2],
                                                       double (&F)[(P + 2) * (P + 3) / 2],
                                                       double (&D)[(P + 2) * (P + 3) / 2]) {
    constexpr int M = P + 2;
#pragma unroll
    for (int p = 0; p < M; ++p)
#pragma unroll
        for (int q = p; q < M; ++q) {
            const double mm = Mk[p] * Mk[q];
            double f;
            if (mm < kEps) {
                f = 0.0;
            } else if (!thresholded) {
                f = mm;
            } else {
                const double h = msh[tri<M>(p, q)];
                f = (h < mm || h != h) ? h : mm;  // Julia min(mm, h)
            }
            F[tri<M>(p, q)] = f;
            D[tri<M>(p, q)] = mm - f;
        }
}

// T_m = 1/2 sum_{a,b} c_ab sum_{c<=m} C(m,c) D[a+c][b+m-c]  (c, D symmetric)
template <int P>
__device__ __forceinline__ void contract_promoted(const double (&ckk)[P][P], const double (&D)[(P + 2) * (P + 3) / 2],
                                                  double &T0, double &T1, double &T2) {
    constexpr int M = P + 2;
    T0 = 0.0;
    T1 = 0.0;
    T2 = 0.0;
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) {
            const double c = ckk[a][b];
            T0 = fma(c, D[trisym<M>(a, b)], T0);
            T1 = fma(c, D[trisym<M>(a, b + 1)], T1);
            T2 = fma(c, D[trisym<M>(a, b + 2)] + D[trisym<M>(a + 1, b + 1)], T2);
        }
    T0 *= 0.5;
}


// Regime ranking of a workgroup's 256 parcels by a non-negative float key (the threshold kernels sort their lanes so
// that waves become regime-homogeneous): a counting sort on 255 logarithmic buckets -- eight per binade over
// [2^-8, 2^23.75], i.e. the key to ~9 % -- with one LDS atomic per lane and a workgroup prefix sum, instead of
// comparing every key with every other (1030 VALU instructions per lane, 6 % of the cfg3b kernel; measured equal
// quality of the resulting waves).  Invalid / empty parcels (valid = false) rank last.  The order inside a bucket is
// whatever order the atomics were served in: any permutation gives the same results, parcels are independent.
// sh_cnt: BS (= workgroup size) counters; returns through sh_perm the lane -> slot map: slot `rank` is processed by lane sh_perm[rank].
// `bucket_if_valid` in [0, BS - 2]: regime_bucket<BS>(key).
template <int BS = kBlock>
__device__ __forceinline__ int regime_bucket(float key) {
    const int code = (int)(__float_as_uint(fmaxf(key, 0.0f)) >> 20) - (127 - 8) * 8;
    return code < 0 ? 0 : (code > BS - 2 ? BS - 2 : code);
}

template <int BS = kBlock>
__device__ __forceinline__ void regime_rank(bool valid, int bucket_if_valid, unsigned int *sh_cnt,
                                            unsigned short *sh_perm) {
    // (the lane index through an opaque copy, and the scan through ds_bpermute on addresses derived from it rather than
    // __shfl_up: otherwise the shuffle addresses and LDS addresses of one ranking are kept -- spilled to scratch in the
    // P = 5 families -- across a whole Simpson pass just to be reused by the next ranking)
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int bucket = valid ? bucket_if_valid : BS - 1;
    sh_cnt[t] = 0u;
    __syncthreads();
    const unsigned int pos = atomicAdd(&sh_cnt[bucket], 1u);
    __syncthreads();
    // exclusive prefix sum of the counters: inclusive scan inside each wave, wave totals through LDS
    const unsigned int c = sh_cnt[t];
    unsigned int incl = c;
    const int lane = t & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int up = (unsigned int)__builtin_amdgcn_ds_bpermute(((lane - d) & 63) << 2, (int)incl);
        if (lane >= d) incl += up;
    }
    __shared__ unsigned int sh_wave_tot[BS / 64];
    if (lane == 63) sh_wave_tot[t >> 6] = incl;
    __syncthreads();
    unsigned int offs = 0;
#pragma unroll
    for (int w = 0; w < BS / 64; ++w)
        if (w < (t >> 6)) offs += sh_wave_tot[w];
    sh_cnt[t] = offs + incl - c;  // exclusive prefix of bucket t
    __syncthreads();
    sh_perm[sh_cnt[bucket] + pos] = (unsigned short)t;
    __syncthreads();
}

// Plan-time specialisation (jit.hpp): when the plan constants are compile-time values (SPEC), terms whose tensor
// coefficient is exactly zero are dropped -- fma(0, M, v) == v for finite M, the arithmetic is unchanged -- and the
// remaining coefficients become literals.  In the ahead-of-time kernels (SPEC = false) the test is constant-true.
template <bool SPEC>
__device__ __forceinline__ constexpr bool nzc(double c) {
    return SPEC ? c != 0.0 : true;
}

// The promoted part of the self collisions of ONE mode k (Coalescence.jl:200-244, 353-455): with
// F = get_finite_2d_integrals and D = M_p M_q - F, returns T_m = 1/2 sum c^{kk}_ab C(m,c) D[a+c][b+m-c], the amount
// that order m loses from mode k to mode k+1 (lost altogether for k = N-1).  Zero unless the mode carries a threshold
// or one of its moments is below 2^-26 (the only way M_p M_q < eps can fire, Coalescence.jl:213).
// `has_pass`: whether this mode runs a Simpson pass (FIXED: finite threshold; MOVING: every mode but the last).
template <int N, int P, int MODE>
__device__ __forceinline__ bool mode_has_pass(const KArgs<N, P> &A, int k) {
    if (MODE == MODE_FIXED) return k < N - 1 && A.finite[k];
    if (MODE == MODE_MOVING) return k < N - 1;
    return false;
}

// Split in two so that the regime-sorted kernel can drop every register it does not need across the Simpson pass:
// promoted_pass is the pass itself (needs theta, k and whether n > 0), promoted_finish forms the products and the
// contraction from the mode's moments (needs n).
// the factor n / Gamma(k) (Gamma mode) or n (Exponential) that msh_grid leaves out
__device__ __forceinline__ double msh_pref(bool is_gamma, double n, double k) { return is_gamma ? n / tgamma(k) : n; }

struct PromotedFlags {
    bool thresholded, mono, mono_below;
};

template <int N, int P, int MODE, bool FAST = false>
__device__ __forceinline__ PromotedFlags promoted_pass(const KArgs<N, P> &A, const double *__restrict__ nodes, int k,
                                                       bool n_pos, double th, double kk,
                                                       double (&msh)[(P + 2) * (P + 3) / 2]) {
    constexpr int M = P + 2;
    constexpr int T = M * (M + 1) / 2;
    PromotedFlags f{false, false, false};
#pragma unroll
    for (int t = 0; t < T; ++t) msh[t] = 0.0;  // moment_source_helper / M_p2 (Simpson branches)
    if (MODE == MODE_FIXED) {
        if (k < N - 1 && A.finite[k]) {  // wave-uniform
            f.thresholded = true;
            if (A.dist_type[k] == DIST_MONO) {
                // moment_source_helper, ParticleDistributions.jl:557-564: n^2 theta^(p1+p2) if theta < x_t/2, else 0
                f.mono = true;
                f.mono_below = th < 0.5 * A.thr[k];
            } else if (A.dist_type[k] == DIST_LOGNORMAL) {
                if (n_pos) msh_lognormal<P>(A.thr[k], th, kk, msh);  // (theta, k) slots hold (mu, sigma)
            } else if (n_pos) {
                const FixedGrid grid{nodes + (size_t)A.node_off[k] * kNodeStride, A.n_bins[k]};
                if (FAST)
                    msh_grid_f32<P>(grid, A.thr[k], th, kk, A.dist_type[k] == DIST_GAMMA, msh);
                else
                    msh_grid<P>(grid, A.thr[k], th, kk, A.dist_type[k] == DIST_GAMMA, msh);
            }
        }
    } else if (MODE == MODE_MOVING) {
        if (k < N - 1) {
            const bool is_gamma = A.dist_type[k] == DIST_GAMMA;
            const double xt = moving_threshold<N, P>(A, k, is_gamma, th, kk);
            f.thresholded = !(xt == INFINITY);
            if (f.thresholded && n_pos) {
                if (FAST)
                    msh_grid_f32<P>(MovingGrid(xt, A.nbpl), xt, th, kk, is_gamma, msh);
                else
                    msh_grid<P>(MovingGrid(xt, A.nbpl), xt, th, kk, is_gamma, msh);
            }
        }
    }
    return f;
}

template <int N, int P>
__device__ __forceinline__ void promoted_finish(const KArgs<N, P> &A, int k, double n, double kk, const double (&Mk)[P + 2],
                                                PromotedFlags f, double (&msh)[(P + 2) * (P + 3) / 2], double &T0,
                                                double &T1, double &T2) {
    constexpr int M = P + 2;
    constexpr int T = M * (M + 1) / 2;
    T0 = T1 = T2 = 0.0;
    const double pref = msh_pref(A.dist_type[k] == DIST_GAMMA, n, kk);
#pragma unroll
    for (int p1 = 0; p1 < M; ++p1)
#pragma unroll
        for (int p2 = p1; p2 < M; ++p2)
            msh[tri<M>(p1, p2)] = f.mono ? (f.mono_below ? Mk[p1] * Mk[p2] : 0.0) : (pref * msh[tri<M>(p1, p2)]) * Mk[p2];
    // without a threshold D is non-zero only where M_p M_q < eps: impossible when every moment >= 2^-26
    double mn = Mk[0];
#pragma unroll
    for (int q = 1; q < M; ++q)
        if (q < A.n_mom_max) mn = fmin(mn, Mk[q]);
    const bool need = (n > 0.0) && (f.thresholded || mn < kSqrtEps);
    if (need) {
        double F[T], D[T];
        finite_2d_and_promoted<P>(Mk, f.thresholded, msh, F, D);
        contract_promoted<P>(A.c[k][k], D, T0, T1, T2);
    }
}

template <int N, int P, int MODE, bool FAST = false>
__device__ __forceinline__ void promoted_mode(const KArgs<N, P> &A, const double *__restrict__ nodes, int k, double n,
                                              double th, double kk, const double (&Mk)[P + 2], double &T0, double &T1,
                                              double &T2) {
    double msh[(P + 2) * (P + 3) / 2];
    const PromotedFlags f = promoted_pass<N, P, MODE, FAST>(A, nodes, k, n > 0.0, th, kk, msh);
    promoted_finish<N, P>(A, k, n, kk, Mk, f, msh, T0, T1, T2);
}

// ---- pair terms: Q - R for j < k, -R for j > k, S_1(full products) - R for j == k, with the
//      products common to Q and R (and to S_1 and R) cancelled analytically:
//      Q_0 = R_0;  Q_1 = R_1 + sum v1_b Mk_b;  Q_2 = R_2 + 2 sum v1_b Mk_{b+1} + sum v2_b Mk_b
//      with v_s[b] = sum_a c_ab Mj_{a+s}.   Adds to acc.
template <int N, int P, bool SPEC>
__device__ __forceinline__ void pair_terms(const KArgs<N, P> &A, const double (&Mm)[N][P + 2], double (&acc)[N][3]) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j < k) {
                double s1 = 0.0, s2a = 0.0, s2b = 0.0;
#pragma unroll
                for (int b = 0; b < P; ++b) {
                    double v1 = 0.0, v2 = 0.0;
#pragma unroll
                    for (int a = 0; a < P; ++a)
                        if (!SPEC || A.c[j][k][a][b] != 0.0) {
                            v1 = fma(A.c[j][k][a][b], Mm[j][a + 1], v1);
                            v2 = fma(A.c[j][k][a][b], Mm[j][a + 2], v2);
                        }
                    s1 = fma(v1, Mm[k][b], s1);
                    s2a = fma(v1, Mm[k][b + 1], s2a);
                    s2b = fma(v2, Mm[k][b], s2b);
                }
                acc[k][1] += s1;
                acc[k][2] += fma(2.0, s2a, s2b);
            } else if (j > k) {
                double r0 = 0.0, r1 = 0.0, r2 = 0.0;
#pragma unroll
                for (int b = 0; b < P; ++b) {
                    double v = 0.0;
#pragma unroll
                    for (int a = 0; a < P; ++a)
                        if (!SPEC || A.c[j][k][a][b] != 0.0) v = fma(A.c[j][k][a][b], Mm[j][a], v);
                    r0 = fma(v, Mm[k][b], r0);
                    r1 = fma(v, Mm[k][b + 1], r1);
                    r2 = fma(v, Mm[k][b + 2], r2);
                }
                acc[k][0] -= r0;
                acc[k][1] -= r1;
                acc[k][2] -= r2;
            } else {
                double r0 = 0.0, s2 = 0.0;
#pragma unroll
                for (int b = 0; b < P; ++b) {
                    double v = 0.0, v1 = 0.0;
#pragma unroll
                    for (int a = 0; a < P; ++a)
                        if (!SPEC || A.c[k][k][a][b] != 0.0) {
                            v = fma(A.c[k][k][a][b], Mm[k][a], v);
                            v1 = fma(A.c[k][k][a][b], Mm[k][a + 1], v1);
                        }
                    r0 = fma(v, Mm[k][b], r0);
                    s2 = fma(v1, Mm[k][b + 1], s2);
                }
                acc[k][0] -= 0.5 * r0;  // S_1 - R, order 0
                acc[k][2] += s2;        // S_1 - R, order 2 (order 1 cancels: mass conservation)
            }
        }
    }

}

// get_coal_ints for one parcel: acc[k][m], normalised units.
template <int N, int P, int MODE, bool FAST = false, bool SPEC = false>
__device__ __forceinline__ void coal_ints_parcel(const KArgs<N, P> &A, const double *__restrict__ nodes,
                                                 const double (&nn)[N], const double (&th)[N], const double (&kk)[N],
                                                 double (&acc)[N][3]) {
    constexpr int M = P + 2;
    double Mm[N][M];
#pragma unroll
    for (int i = 0; i < N; ++i) moment_row<M>(A.dist_type[i], nn[i], th[i], kk[i], A.n_mom_max, Mm[i]);
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0;
    // ---- promoted part of the self collisions: moves T_m from mode k to mode k+1 (lost for k = N-1)
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double T0, T1, T2;
        promoted_mode<N, P, MODE, FAST>(A, nodes, k, nn[k], th[k], kk[k], Mm[k], T0, T1, T2);
        acc[k][0] -= T0;
        acc[k][1] -= T1;
        acc[k][2] -= T2;
        if (k + 1 < N) {
            acc[k + 1][0] += T0;
            acc[k + 1][1] += T1;
            acc[k + 1][2] += T2;
        }
    }
    pair_terms<N, P, SPEC>(A, Mm, acc);
}

// get_coal_ints for the parcel of every lane of a workgroup of BS threads, computed cooperatively: phases 1 and 2 of
// coal_rhs_sorted_body.inc for parcels that are already in registers (the fused integrators call this once per RHS
// evaluation, so their lanes are re-ranked on the CURRENT state of every stage, per thresholded mode; round 1 ranked them
// once, on the initial state and the first thresholded mode only: 64 % active lanes in the rainshaft integrator).
// Every lane of the workgroup must call it (barriers); `valid` = the lane holds a parcel.
// RELOAD: nn / th / kk come back as the lane's LDS slots hold them (the same bits for a valid lane; (0, 1, 1) otherwise), so
// that a caller that needs them after the passes does not keep them in registers across the passes.
template <int N, int P, int MODE, bool SPEC, int BS, bool RELOAD>
__device__ __forceinline__ void coal_ints_ranked_impl(const KArgs<N, P> &A, const double *__restrict__ nodes, bool valid,
                                                      double (&nn)[N], double (&th)[N], double (&kk)[N], double (&acc)[N][3],
                                                      double (*&par_rows)[BS]) {
    constexpr int M_ = P + 2;
    __shared__ double sh_par[3 * N][BS];
    par_rows = sh_par;  // (3N rows the caller may reuse once every lane has its parameters back: after its next barrier)
    __shared__ double sh_T[(N > 1 ? N - 1 : 1) * 3][BS];
    __shared__ unsigned int sh_cnt[BS];
    __shared__ unsigned short sh_perm[BS];
    {
        const int t0 = threadIdx.x;
#pragma unroll
        for (int m = 0; m < N; ++m) {
            sh_par[3 * m + 0][t0] = valid ? nn[m] : 0.0;
            sh_par[3 * m + 1][t0] = valid ? th[m] : 1.0;
            sh_par[3 * m + 2][t0] = valid ? kk[m] : 1.0;
        }
    }
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
        if (!mode_has_pass<N, P, MODE>(A, k)) continue;  // wave-uniform
        // (as in coal_rhs_sorted_body.inc: nothing but the pass's operands stays in registers across it)
        int tk = threadIdx.x;
        asm volatile("" : "+v"(tk));
        const double th_own = sh_par[3 * k + 1][tk], kk_own = sh_par[3 * k + 2][tk], nn_own = sh_par[3 * k + 0][tk];
        const float r = (MODE == MODE_FIXED) ? (float)(A.thr[k] / th_own) : (float)kk_own;
        regime_rank<BS>(nn_own > 0.0 && r == r, regime_bucket<BS>(r), sh_cnt, sh_perm);
        double msh[M_ * (M_ + 1) / 2];
        PromotedFlags pf;
        {
            const int src = sh_perm[tk];
            pf = promoted_pass<N, P, MODE, false>(A, nodes, k, sh_par[3 * k + 0][src] > 0.0, sh_par[3 * k + 1][src],
                                                  sh_par[3 * k + 2][src], msh);
        }
        int tf = threadIdx.x;
        asm volatile("" : "+v"(tf));
        const int src = sh_perm[tf];
        const double n_s = sh_par[3 * k + 0][src], th_s = sh_par[3 * k + 1][src], kk_s = sh_par[3 * k + 2][src];
        double Mk[M_], T0, T1, T2;
        moment_row<M_>(A.dist_type[k], n_s, th_s, kk_s, A.n_mom_max, Mk);
        promoted_finish<N, P>(A, k, n_s, kk_s, Mk, pf, msh, T0, T1, T2);
        sh_T[3 * k + 0][src] = T0;
        sh_T[3 * k + 1][src] = T1;
        sh_T[3 * k + 2][src] = T2;
        __syncthreads();
    }
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    double nn2[N], th2[N], kk2[N], Mm[N][M_];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        nn2[m] = sh_par[3 * m + 0][t];
        th2[m] = sh_par[3 * m + 1][t];
        kk2[m] = sh_par[3 * m + 2][t];
        moment_row<M_>(A.dist_type[m], nn2[m], th2[m], kk2[m], A.n_mom_max, Mm[m]);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double T0, T1, T2;
        if (k < N - 1 && mode_has_pass<N, P, MODE>(A, k)) {
            T0 = sh_T[3 * k + 0][t];
            T1 = sh_T[3 * k + 1][t];
            T2 = sh_T[3 * k + 2][t];
        } else {  // no threshold on this mode: only the (rare) M_p M_q < eps rule can promote anything
            promoted_mode<N, P, MODE, false>(A, nodes, k, nn2[k], th2[k], kk2[k], Mm[k], T0, T1, T2);
        }
        acc[k][0] -= T0;
        acc[k][1] -= T1;
        acc[k][2] -= T2;
        if (k + 1 < N) {
            acc[k + 1][0] += T0;
            acc[k + 1][1] += T1;
            acc[k + 1][2] += T2;
        }
    }
    pair_terms<N, P, SPEC>(A, Mm, acc);
    if (RELOAD) {  // (read again rather than kept across pair_terms: 3N doubles the moment rows and pair sums do not need)
        int t3 = threadIdx.x;
        asm volatile("" : "+v"(t3));
#pragma unroll
        for (int m = 0; m < N; ++m) {
            nn[m] = sh_par[3 * m + 0][t3];
            th[m] = sh_par[3 * m + 1][t3];
            kk[m] = sh_par[3 * m + 2][t3];
        }
    }
}
template <int N, int P, int MODE, bool SPEC, int BS>
__device__ __forceinline__ void coal_ints_ranked(const KArgs<N, P> &A, const double *__restrict__ nodes, bool valid,
                                                 const double (&nn)[N], const double (&th)[N], const double (&kk)[N],
                                                 double (&acc)[N][3]) {
    double a[N], b[N], c[N];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        a[m] = nn[m];
        b[m] = th[m];
        c[m] = kk[m];
    }
    double (*unused)[BS];
    coal_ints_ranked_impl<N, P, MODE, SPEC, BS, false>(A, nodes, valid, a, b, c, acc, unused);
}

struct SediArgs {
    int32_t n_vel, pad;
    double vel[4][2];  // already rescaled by norms[1]^vel[k][1] (rainshaft_helpers.jl:74-76)
};

// get_sedimentation_flux, Sedimentation.jl:22-37: flux[i][j] = -sum_v vel_v0 * M^i_{j-1+vel_v1}
// with the fractional-order moment n theta^q Gamma(q+k)/Gamma(k); normalised units.
template <int N>
__device__ __forceinline__ void sedi_flux_parcel(const int32_t (&dist_type)[N], const SediArgs &S, const double (&nn)[N],
                                                 const double (&th)[N], const double (&kk)[N], double (&fl)[N][3]) {
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const int dtp = dist_type[m];
        const bool gam = (dtp == DIST_GAMMA || dtp == DIST_EXP);
        const double lnth = (dtp == DIST_LOGNORMAL) ? th[m] : log_pos(th[m]);
        double s[3] = {0.0, 0.0, 0.0};
        for (int v = 0; v < S.n_vel; ++v) {
            const double qv = S.vel[v][1];
            if (gam) {
                // M_qv from the log-gamma ratio once, the higher orders by M_{q+1} = M_q theta (k + q)
                double mom = nn[m] * (qv == 0.0 ? 1.0 : exp_fin(fma(qv, lnth, log_gamma_ratio(kk[m], qv))));
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    s[j] = fma(-S.vel[v][0], mom, s[j]);
                    mom *= th[m] * (kk[m] + qv + double(j));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double q = double(j) + qv;  // log(M_q / n), ParticleDistributions.jl:193-207
                    const double e = (dtp == DIST_MONO) ? q * lnth : fma(q, lnth, 0.5 * q * q * (kk[m] * kk[m]));
                    s[j] = fma(-S.vel[v][0], nn[m] * exp(e), s[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) fl[m][j] = s[j];
    }
}

// load one parcel (moments -> normalise -> invert, or parameters as given)
template <int N, int P, typename TIO = double>
__device__ __forceinline__ bool load_parcel(const KArgs<N, P> &A, size_t i, size_t ld, const TIO *__restrict__ in,
                                            double (&nn)[N], double (&th)[N], double (&kk)[N]) {
    bool all_small = true;
    if (A.input_kind == IN_MOMENTS) {
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const int off = A.off[m];
            double m0 = ld_stream(in + (size_t)(off + 0) * ld + i);
            double m1 = ld_stream(in + (size_t)(off + 1) * ld + i);
            double m2 = 0.0;
            const bool three = A.np[m] == 3;
            if (three) m2 = ld_stream(in + (size_t)(off + 2) * ld + i);
            if (A.rainshaft) {  // rainshaft_helpers.jl:52
                m0 = m0 < 0.0 ? 0.0 : m0;
                m1 = m1 < 0.0 ? 0.0 : m1;
                m2 = m2 < 0.0 ? 0.0 : m2;
            }
            // mom ./ mom_norms (box_model_helpers.jl:31) as a correctly rounded division: the closure inversion
            // below is ill-conditioned at zero variance, where one ulp decides the sign of M2/M1 - M1/M0.
            m0 = div_by_const(m0, A.norm[3 * m + 0], A.inv_norm[3 * m + 0]);
            m1 = div_by_const(m1, A.norm[3 * m + 1], A.inv_norm[3 * m + 1]);
            m2 = div_by_const(m2, A.norm[3 * m + 2], A.inv_norm[3 * m + 2]);
            all_small = all_small && (m0 < kEps) && (m1 < kEps) && (!three || m2 < kEps);
            invert_closure(A.dist_type[m], m0, m1, m2, A.kmin, A.kmax, nn[m], th[m], kk[m]);
        }
    } else {
#pragma unroll
        for (int m = 0; m < N; ++m) {
            nn[m] = ld_stream(in + (size_t)(3 * m + 0) * ld + i);
            th[m] = ld_stream(in + (size_t)(3 * m + 1) * ld + i);
            kk[m] = (A.np[m] == 3) ? ld_stream(in + (size_t)(3 * m + 2) * ld + i) : 1.0;
        }
        all_small = false;
    }
    return all_small;
}

// The body of the one-parcel kernel is a source fragment included twice: in the ahead-of-time kernel, where the plan
// constants are the by-value kernel argument, and in coal_rhs_body for the plan-time compiled kernels (jit.hpp), where
// they are a constexpr object.  (A shared inline function taking `const KArgs &` would do -- but binding a reference
// to a by-value kernel argument makes clang copy the whole struct out of the kernarg segment at kernel entry: all of
// it live in SGPRs at once, 185-270 spilled-SGPR reloads in the hot path, +35 % VALU instructions measured.)
template <int N, int P, int MODE, typename TIO, bool SPEC = false>
__device__ __forceinline__ void coal_rhs_body(const KArgs<N, P> &A, const double *__restrict__ nodes, size_t n, size_t ld,
                                              const TIO *__restrict__ in, TIO *__restrict__ out,
                                              const SediArgs *sedi = nullptr, TIO *__restrict__ out2 = nullptr) {
#define CLOUDY_SPEC SPEC
#include "coal_rhs_body.inc"
#undef CLOUDY_SPEC
}

template <int N, int P, int MODE, typename TIO>
__global__ void __launch_bounds__(kBlock)
    coal_rhs_kernel(const KArgs<N, P> A, const double *__restrict__ nodes, size_t n, size_t ld,
                    const TIO *__restrict__ in, TIO *__restrict__ out) {
    constexpr const SediArgs *sedi = nullptr;  // (the fused sedimentation flux exists in the plan-time compiled kernels)
    constexpr TIO *out2 = nullptr;
#define CLOUDY_SPEC false
#include "coal_rhs_body.inc"
#undef CLOUDY_SPEC
}

// ALLINF with two parcels per lane: every plane is read and written as one 16-byte access per lane (1 KiB per
// wave instruction).  Requires 16-byte aligned planes: base pointers 16-B aligned and ld even (checked by the
// host, which otherwise launches the one-parcel kernel).  The two parcels are independent instruction streams,
// which also gives the scheduler ILP across the division sequences of the closure inversion.
template <int N, int P, typename TIO, bool SPEC = false>
__device__ __forceinline__ void coal_rhs_allinf2_body(const KArgs<N, P> &A, size_t n, size_t ld,
                                                      const TIO *__restrict__ in, TIO *__restrict__ out) {
#define CLOUDY_SPEC SPEC
#include "coal_rhs_allinf2_body.inc"
#undef CLOUDY_SPEC
}

template <int N, int P, typename TIO>
__global__ void __launch_bounds__(kBlock)
    coal_rhs_allinf2_kernel(const KArgs<N, P> A, size_t n, size_t ld, const TIO *__restrict__ in,
                            TIO *__restrict__ out) {
#define CLOUDY_SPEC false
#include "coal_rhs_allinf2_body.inc"
#undef CLOUDY_SPEC
}

// Threshold modes (FIXED / MOVING): the cost of a parcel is dominated by the incomplete-gamma evaluations of its
// Simpson nodes, and which algorithm a node takes (power series for z <= a+1, continued fraction above; P == 1
// far above) is decided by r = (x_t/theta) / (a_top + 1) of the parcel and mode.  Lanes of one wave execute every
// branch any of them takes, so for each thresholded mode the 256 parcels of a workgroup are ranked by that mode's r
// (regime_rank) and each lane integrates the mode of the parcel of its rank: waves become regime-homogeneous.  The
// three phases are described in coal_rhs_sorted_body.inc.
// Workgroup size of the threshold kernel: 256, or 512 for plans with N <= 2 and a single thresholded mode (a ranking
// over 512 parcels gives more homogeneous waves: cfg3b -6 %; with two thresholded modes, moving thresholds or the
// larger LDS footprint of N >= 3 it measured slower).
template <int N, int P, int MODE, typename TIO, bool FAST = false, bool SPEC = false, int BS = kBlock>
__device__ __forceinline__ void coal_rhs_sorted_body(const KArgs<N, P> &A, const double *__restrict__ nodes, size_t n,
                                                     size_t ld, const TIO *__restrict__ in, TIO *__restrict__ out,
                                                     const SediArgs *sedi = nullptr, TIO *__restrict__ out2 = nullptr) {
#define CLOUDY_SPEC SPEC
#define CLOUDY_BS BS
#include "coal_rhs_sorted_body.inc"
#undef CLOUDY_BS
#undef CLOUDY_SPEC
}

// Occupancy target of the threshold kernels: the Simpson pass is a long chain of dependent fp64 operations, and going
// from 2 to 3-4 resident waves per SIMD is worth more than the few spilled registers it costs the P = 5 families
// (cfg4, N = 3, P = 5: 184 VGPRs / occupancy 2 -> 128 VGPRs + 18 spilled / occupancy 4: 17.5 -> 14.3 ms).
#ifndef CLOUDY_SORTED_WAVES
#define CLOUDY_SORTED_WAVES 4
#endif
template <int N, int P, int MODE, typename TIO, bool FAST = false, int BS = kBlock>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(CLOUDY_SORTED_WAVES)))
    coal_rhs_sorted_kernel(const KArgs<N, P> A, const double *__restrict__ nodes, size_t n, size_t ld,
                           const TIO *__restrict__ in, TIO *__restrict__ out) {
    constexpr const SediArgs *sedi = nullptr;  // (the fused sedimentation flux exists in the plan-time compiled kernels)
    constexpr TIO *out2 = nullptr;
#define CLOUDY_SPEC false
#define CLOUDY_BS BS
#include "coal_rhs_sorted_body.inc"
#undef CLOUDY_BS
#undef CLOUDY_SPEC
}

// ---- on-device explicit time stepping -------------------------------------------------------
// The reference drivers integrate du/dt = rhs!(u) with SSPRK33 and a fixed dt (e.g. box_single_gamma.jl:35-36,
// 3 RHS evaluations per step, each a full pass over the state in OrdinaryDiffEq).  Here a lane keeps its parcel's
// moments in registers across all stages and steps: HBM traffic is one read and one write of the state per CALL
// (n_steps steps), and there are no per-stage launches.
// (thresholded plans: every lane of the workgroup calls this, `valid` = the lane holds a parcel; see coal_ints_ranked)
template <int N, int P, int MODE, bool SPEC = false, int BS = kBlock>
__device__ __forceinline__ void rhs_physical(const KArgs<N, P> &A, const double *__restrict__ nodes, bool valid,
                                             const double (&u)[N][3], double (&f)[N][3]) {
    double nn[N], th[N], kk[N], acc[N][3];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const double m0 = div_by_const(u[m][0], A.norm[3 * m + 0], A.inv_norm[3 * m + 0]);
        const double m1 = div_by_const(u[m][1], A.norm[3 * m + 1], A.inv_norm[3 * m + 1]);
        const double m2 = div_by_const(u[m][2], A.norm[3 * m + 2], A.inv_norm[3 * m + 2]);
        invert_closure(A.dist_type[m], m0, m1, m2, A.kmin, A.kmax, nn[m], th[m], kk[m]);
    }
    if (MODE == MODE_ALLINF)
        coal_ints_parcel<N, P, MODE, false, SPEC>(A, nodes, nn, th, kk, acc);
    else
        coal_ints_ranked<N, P, MODE, SPEC, BS>(A, nodes, valid, nn, th, kk, acc);
#pragma unroll
    for (int m = 0; m < N; ++m) {
        f[m][0] = acc[m][0] * A.out_scale[3 * m + 0];
        f[m][1] = acc[m][1] * A.out_scale[3 * m + 1];
        f[m][2] = (A.np[m] == 3) ? acc[m][2] * A.out_scale[3 * m + 2] : 0.0;
    }
}

// The same for a state kept in NORMALISED units (all-Inf plans inside the fused SSPRK33 integrator): d(mom / norms)/dt is
// coal_ints itself, so the six correctly rounded "./ norms" and the six ".* norms" of every evaluation
// (box_model_helpers.jl:31, :52) are paid once per call, on load and store -- 30 of ~265 instructions per evaluation.  The
// stage values then differ from the reference's physical-unit sequence by roundings (1e-16 relative per operation), far
// inside the stepping tolerance; the single-launch operator keeps the reference's sequence.
template <int N, int P, bool SPEC>
__device__ __forceinline__ void rhs_normalised(const KArgs<N, P> &A, const double (&un)[N][3], double (&f)[N][3]) {
    double nn[N], th[N], kk[N], acc[N][3];
#pragma unroll
    for (int m = 0; m < N; ++m) invert_closure(A.dist_type[m], un[m][0], un[m][1], un[m][2], A.kmin, A.kmax, nn[m], th[m], kk[m]);
    coal_ints_parcel<N, P, MODE_ALLINF, false, SPEC>(A, nullptr, nn, th, kk, acc);
#pragma unroll
    for (int m = 0; m < N; ++m) {
        f[m][0] = acc[m][0];
        f[m][1] = acc[m][1];
        f[m][2] = (A.np[m] == 3) ? acc[m][2] : 0.0;
    }
}

// BS: workgroup size (256; 512 for thresholded plans with N <= 2 and one thresholded mode, as coal_rhs_sorted_kernel)
template <int N, int P, int MODE, typename TIO, bool SPEC = false, int BS = kBlock>
__device__ __forceinline__ void ssprk33_body(const KArgs<N, P> *__restrict__ Ag, const double *__restrict__ nodes, size_t n,
                                             size_t ld, const TIO *u_in, TIO *u_out, double dt, int n_steps) {
    // The plan constants come through a device pointer that is re-derived once per stage through an opaque zero offset: with by-value kernel arguments LICM hoists every scalar load of the tensors out of the step/stage
    // loops and then spills ~170 SGPRs into VGPR lanes (348 v_readlane per pass measured); re-deriving the pointer
    // keeps those s_loads inside the stage, where they hit the scalar cache.
    const KArgs<N, P> &A = *Ag;
    const size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    // Thresholded plans: the workgroup re-ranks its parcels in every RHS evaluation (coal_ints_ranked), so lanes without
    // a parcel stay for the barriers; the lane -> parcel map is the natural one (coalesced loads and stores).
    const bool valid = i < n;
    if (MODE == MODE_ALLINF && !valid) return;
    constexpr bool kNormalisedState = MODE == MODE_ALLINF;  // see rhs_normalised
    double u[N][3], up[N][3], f[N][3];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const int off = A.off[m];
        u[m][0] = valid ? (double)u_in[(size_t)(off + 0) * ld + i] : 0.0;
        u[m][1] = valid ? (double)u_in[(size_t)(off + 1) * ld + i] : 0.0;
        u[m][2] = (valid && A.np[m] == 3) ? (double)u_in[(size_t)(off + 2) * ld + i] : 0.0;
        if (kNormalisedState) {
#pragma unroll
            for (int q = 0; q < 3; ++q) u[m][q] = div_by_const(u[m][q], A.norm[3 * m + q], A.inv_norm[3 * m + q]);
        }
    }
#pragma unroll 1
    for (int step = 0; step < n_steps; ++step) {
#pragma unroll
        for (int m = 0; m < N; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) up[m][q] = u[m][q];
#pragma unroll 1
        for (int stage = 0; stage < 3; ++stage) {
            size_t opaque_zero = 0;  // launder an OFFSET, not the pointer: the pointer keeps its global address space
            if (!SPEC) asm volatile("" : "+s"(opaque_zero));  // (compile-time plan constants need no loads at all)
            const KArgs<N, P> *Ap = Ag + opaque_zero;
            if (kNormalisedState)
                rhs_normalised<N, P, SPEC>(*Ap, u, f);
            else
                rhs_physical<N, P, MODE, SPEC, BS>(*Ap, nodes, valid, u, f);
            // OrdinaryDiffEq SSPRK33: u = uprev + dt k;  u = (3 uprev + u + dt k)/4;  u = (uprev + 2u + 2dt k)/3
            // (wave-uniform branch on the stage OUTSIDE the element loops: selects per element would triple the work)
            if (stage == 0) {
#pragma unroll
                for (int m = 0; m < N; ++m)
#pragma unroll
                    for (int q = 0; q < 3; ++q) u[m][q] = up[m][q] + dt * f[m][q];
            } else if (stage == 1) {
#pragma unroll
                for (int m = 0; m < N; ++m)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        u[m][q] = (3.0 * up[m][q] + u[m][q] + dt * f[m][q]) * 0.25;  // "/ 4" is exact
            } else {
#pragma unroll
                for (int m = 0; m < N; ++m)
#pragma unroll
                    for (int q = 0; q < 3; ++q)  // "/ 3" as a correctly rounded division
                        u[m][q] = div_by_const(up[m][q] + 2.0 * u[m][q] + 2.0 * dt * f[m][q], 3.0, 1.0 / 3.0);
            }
        }
    }
    if (!valid) return;
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const int off = A.off[m];
        if (kNormalisedState) {
#pragma unroll
            for (int q = 0; q < 3; ++q) u[m][q] *= A.norm[3 * m + q];
        }
        u_out[(size_t)(off + 0) * ld + i] = (TIO)u[m][0];
        u_out[(size_t)(off + 1) * ld + i] = (TIO)u[m][1];
        if (A.np[m] == 3) u_out[(size_t)(off + 2) * ld + i] = (TIO)u[m][2];
    }
}

template <int N, int P, int MODE, typename TIO, int BS = kBlock>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(CLOUDY_SORTED_WAVES)))
    ssprk33_kernel(const KArgs<N, P> *__restrict__ Ag, const double *__restrict__ nodes, size_t n, size_t ld,
                   const TIO *u_in, TIO *u_out, double dt, int n_steps) {
    ssprk33_body<N, P, MODE, TIO, false, BS>(Ag, nodes, n, ld, u_in, u_out, dt, n_steps);
}

// Fixed-step Tsit5 (Tsitouras 2011, the tableau of OrdinaryDiffEq's Tsit5()) -- BASELINE configs[0] names "CPU Tsit5" for
// the single-box Golovin case.  No reference driver uses it (they all call solve(prob, SSPRK33(), dt = ...)), and
// OrdinaryDiffEq's Tsit5 is adaptive by default with a GLOBAL error norm over the state; a batch of independent parcels has
// no global norm, so this is the explicit 7-stage FSAL tableau applied with the caller's fixed dt, per parcel, state and
// stage derivatives in registers (6 RHS evaluations per step, one read + one write of the state per call).  The 5th-order
// convergence of the tableau is a GPU test (error ratio ~32 per halving of dt against Smoluchowski's closed form).
namespace tsit5 {
constexpr double a21 = 0.161;
constexpr double a31 = -0.008480655492356989, a32 = 0.335480655492357;
constexpr double a41 = 2.8971530571054935, a42 = -6.359448489975075, a43 = 4.3622954328695815;
constexpr double a51 = 5.325864828439257, a52 = -11.748883564062828, a53 = 7.4955393428898365, a54 = -0.09249506636175525;
constexpr double a61 = 5.86145544294642, a62 = -12.92096931784711, a63 = 8.159367898576159, a64 = -0.071584973281401,
                 a65 = -0.028269050394068383;
constexpr double a71 = 0.09646076681806523, a72 = 0.01, a73 = 0.4798896504144996, a74 = 1.379008574103742,
                 a75 = -3.290069515436081, a76 = 2.324710524099774;
}  // namespace tsit5

// The tableau itself, for any right-hand side: rhs(state, derivative) on [N][3] register arrays.  u is advanced in place by
// n_steps steps of size dt; 6 evaluations per step (FSAL: the 7th stage derivative is the next step's first).
template <int N, class RHS>
__device__ __forceinline__ void tsit5_advance(double (&u)[N][3], double dt, int n_steps, RHS &&rhs) {
    using namespace tsit5;
    double k1[N][3], k2[N][3], k3[N][3], k4[N][3], k5[N][3], k6[N][3], w[N][3];
    if (n_steps > 0) rhs(u, k1);
#pragma unroll 1
    for (int step = 0; step < n_steps; ++step) {
#pragma unroll
        for (int m = 0; m < N; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) w[m][q] = fma(dt * a21, k1[m][q], u[m][q]);
        rhs(w, k2);
#pragma unroll
        for (int m = 0; m < N; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) w[m][q] = fma(dt, fma(a31, k1[m][q], a32 * k2[m][q]), u[m][q]);
        rhs(w, k3);
#pragma unroll
        for (int m = 0; m < N; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) w[m][q] = fma(dt, fma(a41, k1[m][q], fma(a42, k2[m][q], a43 * k3[m][q])), u[m][q]);
        rhs(w, k4);
#pragma unroll
        for (int m = 0; m < N; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                w[m][q] = fma(dt, fma(a51, k1[m][q], fma(a52, k2[m][q], fma(a53, k3[m][q], a54 * k4[m][q]))), u[m][q]);
        rhs(w, k5);
#pragma unroll
        for (int m = 0; m < N; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                w[m][q] = fma(dt, fma(a61, k1[m][q], fma(a62, k2[m][q], fma(a63, k3[m][q], fma(a64, k4[m][q], a65 * k5[m][q])))),
                              u[m][q]);
        rhs(w, k6);
#pragma unroll
        for (int m = 0; m < N; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                u[m][q] = fma(dt, fma(a71, k1[m][q], fma(a72, k2[m][q], fma(a73, k3[m][q], fma(a74, k4[m][q],
                                                                                            fma(a75, k5[m][q], a76 * k6[m][q]))))),
                              u[m][q]);
        if (step + 1 < n_steps) rhs(u, k1);  // FSAL
    }
}

// Every plan family the SSPRK33 integrator serves (round 4): thresholds Inf (state kept in NORMALISED units between load
// and store, as ssprk33_body: the tableau is linear in the state), fixed or MOVING (the workgroup re-ranks its parcels in
// every evaluation), fp64 or float planes; SPEC: compiled for the plan (jit.hpp part 4), no loads of plan constants.
template <int N, int P, int MODE, typename TIO, bool SPEC = false, int BS = kBlock>
__device__ __forceinline__ void tsit5_body(const KArgs<N, P> *__restrict__ Ag, const double *__restrict__ nodes, size_t n,
                                           size_t ld, const TIO *u_in, TIO *u_out, double dt, int n_steps) {
    const KArgs<N, P> &A = *Ag;
    const size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    const bool valid = i < n;
    if (MODE == MODE_ALLINF && !valid) return;
    constexpr bool kNormalisedState = MODE == MODE_ALLINF;  // see rhs_normalised
    double u[N][3];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const int off = A.off[m];
        u[m][0] = valid ? (double)u_in[(size_t)(off + 0) * ld + i] : 0.0;
        u[m][1] = valid ? (double)u_in[(size_t)(off + 1) * ld + i] : 0.0;
        u[m][2] = (valid && A.np[m] == 3) ? (double)u_in[(size_t)(off + 2) * ld + i] : 0.0;
        if (kNormalisedState) {
#pragma unroll
            for (int q = 0; q < 3; ++q) u[m][q] = div_by_const(u[m][q], A.norm[3 * m + q], A.inv_norm[3 * m + q]);
        }
    }
    // the plan constants through an opaque zero offset per RHS evaluation (see ssprk33_body); none when compiled for the plan
    tsit5_advance<N>(u, dt, n_steps, [&](const double (&state)[N][3], double (&deriv)[N][3]) {
        size_t oz = 0;
        if (!SPEC) asm volatile("" : "+s"(oz));
        if (kNormalisedState)
            rhs_normalised<N, P, SPEC>(*(Ag + oz), state, deriv);
        else
            rhs_physical<N, P, MODE, SPEC, BS>(*(Ag + oz), nodes, valid, state, deriv);
    });
    if (!valid) return;
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const int off = A.off[m];
        if (kNormalisedState) {
#pragma unroll
            for (int q = 0; q < 3; ++q) u[m][q] *= A.norm[3 * m + q];
        }
        u_out[(size_t)(off + 0) * ld + i] = (TIO)u[m][0];
        u_out[(size_t)(off + 1) * ld + i] = (TIO)u[m][1];
        if (A.np[m] == 3) u_out[(size_t)(off + 2) * ld + i] = (TIO)u[m][2];
    }
}

template <int N, int P, int MODE, typename TIO, int BS = kBlock>
__global__ void __launch_bounds__(BS)
    tsit5_kernel(const KArgs<N, P> *__restrict__ Ag, const double *__restrict__ nodes, size_t n, size_t ld, const TIO *u_in,
                 TIO *u_out, double dt, int n_steps) {
    tsit5_body<N, P, MODE, TIO, false, BS>(Ag, nodes, n, ld, u_in, u_out, dt, n_steps);
}

// ---- diagnostics / the callers either side of the operator ---------------------------------------

template <int N, int P>
__device__ __forceinline__ void update_dist_body(const KArgs<N, P> &A, size_t n, size_t ld, const double *__restrict__ in,
                       double *__restrict__ params) {
    // one parcel per lane, no grid-stride loop: a parcel loop would let LICM hoist every libm polynomial
    // constant of the body into registers for the whole kernel (measured: 256 VGPRs + spills vs ~100).
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        double nn[N], th[N], kk[N];
        load_parcel<N, P>(A, i, ld, in, nn, th, kk);
#pragma unroll
        for (int m = 0; m < N; ++m) {
            params[(size_t)(3 * m + 0) * ld + i] = nn[m];
            params[(size_t)(3 * m + 1) * ld + i] = th[m];
            params[(size_t)(3 * m + 2) * ld + i] = kk[m];
        }
    }
}
template <int N, int P>
__global__ void __launch_bounds__(kBlock)
    update_dist_kernel(const KArgs<N, P> A, size_t n, size_t ld, const double *__restrict__ in,
                       double *__restrict__ params) {
    update_dist_body<N, P>(A, n, ld, in, params);
}

// get_finite_2d_integrals (Coalescence.jl:200-244) and the thresholds it used; F planes (i, p1, p2)
template <int N, int P, int MODE>
__device__ __forceinline__ void finite_2d_body(const KArgs<N, P> &A, const double *__restrict__ nodes, size_t n, size_t ld,
                     const double *__restrict__ in, double *__restrict__ F, double *__restrict__ thr_out) {
    constexpr int M = P + 2;
    constexpr int T = M * (M + 1) / 2;
    // one parcel per lane, no grid-stride loop: a parcel loop would let LICM hoist every libm polynomial
    // constant of the body into registers for the whole kernel (measured: 256 VGPRs + spills vs ~100).
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        double nn[N], th[N], kk[N];
        load_parcel<N, P>(A, i, ld, in, nn, th, kk);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double Mk[M];
            moment_row<M>(A.dist_type[k], nn[k], th[k], kk[k], A.n_mom_max, Mk);
            bool thresholded = false;
            double xt = INFINITY;
            double msh[T];
#pragma unroll
            for (int t = 0; t < T; ++t) msh[t] = 0.0;
            const bool is_gamma = A.dist_type[k] == DIST_GAMMA;
            if (MODE == MODE_FIXED) {
                if (k < N - 1 && A.finite[k]) {
                    thresholded = true;
                    xt = A.thr[k];
                    if (A.dist_type[k] == DIST_MONO) {
                        const bool below = th[k] < 0.5 * A.thr[k];
#pragma unroll
                        for (int p1 = 0; p1 < M; ++p1)
#pragma unroll
                            for (int p2 = p1; p2 < M; ++p2) msh[tri<M>(p1, p2)] = below ? Mk[p1] * Mk[p2] : 0.0;
                    } else if (A.dist_type[k] == DIST_LOGNORMAL) {
                        if (nn[k] > 0.0) msh_lognormal<P>(A.thr[k], th[k], kk[k], msh);
                    } else if (nn[k] > 0.0)
                        msh_grid<P>(FixedGrid{nodes + (size_t)A.node_off[k] * kNodeStride, A.n_bins[k]}, A.thr[k], th[k],
                                    kk[k], is_gamma, msh);
                }
            } else if (MODE == MODE_MOVING) {
                if (k < N - 1) {
                    xt = moving_threshold<N, P>(A, k, is_gamma, th[k], kk[k]);
                    thresholded = !(xt == INFINITY);
                    if (thresholded && nn[k] > 0.0)
                        msh_grid<P>(MovingGrid(xt, A.nbpl), xt, th[k], kk[k], is_gamma, msh);
                }
            }
            if (thr_out) thr_out[(size_t)k * ld + i] = xt;
            if (F) {
                if (!(MODE == MODE_FIXED && A.dist_type[k] == DIST_MONO)) {  // msh_grid leaves msh_pref and M_p2 to its caller
                    const double pref = msh_pref(is_gamma, nn[k], kk[k]);
#pragma unroll
                    for (int p1 = 0; p1 < M; ++p1)
#pragma unroll
                        for (int p2 = p1; p2 < M; ++p2) msh[tri<M>(p1, p2)] = (pref * msh[tri<M>(p1, p2)]) * Mk[p2];
                }
                double Fm[T], D[T];
                finite_2d_and_promoted<P>(Mk, thresholded, msh, Fm, D);
#pragma unroll
                for (int p = 0; p < M; ++p)
#pragma unroll
                    for (int q = 0; q < M; ++q) {
                        double v = Fm[trisym<M>(p, q)];
                        if (p >= A.n_2d[k] || q >= A.n_2d[k]) v = 0.0;  // Coalescence.jl:213 N_2d_ints mask
                        F[(size_t)((k * M + p) * M + q) * ld + i] = v;
                    }
            }
        }
    }
}
template <int N, int P, int MODE>
__global__ void __launch_bounds__(kBlock)
    finite_2d_kernel(const KArgs<N, P> A, const double *__restrict__ nodes, size_t n, size_t ld,
                     const double *__restrict__ in, double *__restrict__ F, double *__restrict__ thr_out) {
    finite_2d_body<N, P, MODE>(A, nodes, n, ld, in, F, thr_out);
}

#define CLOUDY_STAGE_BARRIER() __syncthreads()

#ifndef CLOUDY_RS_BLOCK
#define CLOUDY_RS_BLOCK 256
#endif
constexpr int kRainshaftBlock = CLOUDY_RS_BLOCK;  // workgroup size of the fused column integrator

template <int N, int P, typename TIO>
__device__ __forceinline__ void sedi_flux_body(const KArgs<N, P> &A, const SediArgs &S, size_t n, size_t ld, const TIO *__restrict__ in,
                     TIO *__restrict__ out) {
    // one parcel per lane, no grid-stride loop: a parcel loop would let LICM hoist every libm polynomial
    // constant of the body into registers for the whole kernel (measured: 256 VGPRs + spills vs ~100).
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        double nn[N], th[N], kk[N], fl[N][3];
        load_parcel<N, P, TIO>(A, i, ld, in, nn, th, kk);
        sedi_flux_parcel<N>(A.dist_type, S, nn, th, kk, fl);
#pragma unroll
        for (int m = 0; m < N; ++m)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < A.np[m]) out[(size_t)(A.off[m] + j) * ld + i] = (TIO)(fl[m][j] * A.out_scale[3 * m + j]);
    }
}
template <int N, int P, typename TIO>
__global__ void __launch_bounds__(kBlock)
    sedi_flux_kernel(const KArgs<N, P> A, const SediArgs S, size_t n, size_t ld, const TIO *__restrict__ in,
                     TIO *__restrict__ out) {
    sedi_flux_body<N, P, TIO>(A, S, n, ld, in, out);
}

// solve(ODEProblem(make_rainshaft_rhs(...), m, tspan, p), SSPRK33(), dt) of the rainshaft drivers
// (test/examples/Analytical/rainshaft_gamma_mixture.jl:59-60, rainshaft_helpers.jl:45-89) for many independent
// columns of nz <= kBlock cells: a workgroup owns floor(kBlock / nz) whole columns, the state stays in registers over
// all stages and steps, and the only cross-cell coupling -- the upwind flux of the cell above -- goes through LDS.
// Every RHS evaluation first clamps negative moments of its argument to zero IN PLACE (rainshaft_helpers.jl:52 mutates
// the array the integrator passed), which includes the FSAL evaluation on the final state of each step.
// (SPEC: the plan constants Ag / Sg are compile-time objects of a kernel compiled for the plan, jit.hpp)
// BS: workgroup size = cells per workgroup (floor(BS / nz) whole columns).  The ahead-of-time kernel has 256 threads; the
// kernel compiled for the plan exists with 256, 512 and 1024 threads, and jit_rainshaft_part() picks by column height (512 for
// the reference's 20 cells: round 5; taller columns than 1024 cells are stepped stage by stage, cloudy_hip.hip).
// RHS_ONLY (round 5): ONE evaluation of the column right-hand side -- make_rainshaft_rhs(...)'s rhs!(dm, m, par, t), the drop-in
// boundary of the rainshaft drivers -- in one launch: u_out receives coal_source .+ sedi_source of the (clamped) state, flux_out
// the cell fluxes.  The same stage as the integrator's without the update and without anything parked (cloudy_rainshaft_rhs
// otherwise runs the cell kernel and a divergence launch: 2.9 GB of HBM traffic per 1e7 cells instead of 1.4).
template <int N, int P, int MODE, typename TIO, bool SPEC = false, int BS = kRainshaftBlock, bool RHS_ONLY = false>
__device__ __forceinline__ void rainshaft_ssprk33_body(const KArgs<N, P> *__restrict__ Ag, const SediArgs *__restrict__ Sg,
                                                       const double *__restrict__ nodes, int nz, size_t n_columns,
                                                       size_t ld, const TIO *u_in, TIO *u_out, double dt, double dz,
                                                       int n_steps, TIO *flux_out = nullptr) {
    // (workgroup sizes that were measured and dropped, 20-cell columns: 384 threads 40 % slower, 320 threads 40 % slower --
    // six- and five-wave workgroups do not spread evenly over four SIMDs; 128 threads 8 % slower, the ranking over 128 cells
    // is too coarse)
    __shared__ double sh_flux[N * 3][BS];
    // Round 5: (almost) NOTHING of the integrator's state is live across the Simpson passes of a stage (before: u, u_prev and
    // the flux divergence, 18 doubles for two 3-moment modes -> 69 registers in scratch at the 4-wave occupancy target, 7.7 x
    // the algorithmic HBM traffic; now none, 1.01 x).  (i) The sedimentation flux and its exchange come AFTER the
    // passes, from (n, theta, k) read back from the ranking's LDS slots -- whose 3N rows, read by nobody but their owner once
    // the passes are done, carry the exchange; (ii) of u and u_prev the update needs one combination per stage -- w = u_prev,
    // 3 u_prev + u, u_prev + 2 u, formed BEFORE the passes exactly as the update formula forms it (same bits) -- and u_prev
    // itself in stage 1 (for stage 2); (iii) w waits in sh_flux, as many rows of u_prev as fit at 16 waves per CU in sh_up
    // (4 of 6 for N = 2), the empty-cell flag in a byte; (iv) the lane's place in its column is re-derived, not kept; (v) the
    // lane's own flux is read back from the exchange rows after the barrier (kept, it was the last thing the allocator spilled).
    constexpr bool kPark = (MODE != MODE_ALLINF);   // (also decides the re-derivation of the lane's place after the passes)
    constexpr bool kParkState = kPark && !RHS_ONLY;
    constexpr int kRowBytes = BS * 8;
    constexpr int kLdsUsed = (3 * N + 3 * N + 3 * (N > 1 ? N - 1 : 1)) * kRowBytes + BS * 6;  // flux, sh_par, sh_T, cnt, perm
    constexpr int kLdsBudget = 40960 * (BS / 256);  // 160 KB per CU, 16 waves: 4 / 2 / 1 workgroups of 256 / 512 / 1024 threads
    constexpr int kLdsFreeRows = (kLdsBudget - kLdsUsed) / kRowBytes;
    constexpr int NUP = !kParkState ? 0 : kLdsFreeRows < 0 ? 0 : kLdsFreeRows > 3 * N ? 3 * N : kLdsFreeRows;
    __shared__ double sh_up[NUP > 0 ? NUP : 1][BS];
    __shared__ unsigned char sh_small[kPark ? BS : 1];  // the empty-cell flag of the stage (:67-72) waits here as well
    const KArgs<N, P> &A = *Ag;
    const int cpb = BS / nz;       // whole columns per workgroup
    const int pos = threadIdx.x;   // cell slot of the workgroup this lane integrates: column pos / nz, level pos % nz
    // (With a finite threshold the Simpson passes of every stage run on cells re-ranked on that stage's state, per
    // thresholded mode: coal_ints_ranked.  Round 1 permuted the lanes once, on the initial state.)
    const int cl = pos / nz, iz = pos - cl * nz;
    const size_t col = (size_t)blockIdx.x * cpb + cl;
    const bool active = (cl < cpb) && (col < n_columns);
    const size_t i = col * (size_t)nz + iz;
    double u[N][3], up[N][3];
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            u[m][q] = (active && q < A.np[m]) ? (double)u_in[(size_t)(A.off[m] + q) * ld + i] : 0.0;
            up[m][q] = 0.0;
        }
    const int n_steps_run = RHS_ONLY ? 1 : n_steps;
#pragma unroll 1
    for (int step = 0; step < n_steps_run; ++step) {
#pragma unroll 1
        for (int stage = 0; stage < (RHS_ONLY ? 1 : 3); ++stage) {
            // The plan constants are re-derived through opaque zero offsets (see ssprk33_kernel), three times per stage: the
            // closure inversion sees the norms, the coalescence integrals the tensors, the sedimentation flux the velocity
            // block -- otherwise all of them are live in SGPRs at once and spill into VGPR lanes.
            size_t opaque_zero = 0;
            if (!SPEC) asm volatile("" : "+s"(opaque_zero));
            const KArgs<N, P> &As = *(Ag + opaque_zero);
            double nn[N], th[N], kk[N], w[N][3];
            int ps = threadIdx.x;
            if (kPark) asm volatile("" : "+v"(ps));
            const int cl1 = ps / nz;
            const bool active1 = kPark ? (cl1 < cpb) && ((size_t)blockIdx.x * cpb + cl1 < n_columns) : active;
            bool all_small = true;
            // the clamp of the RHS (rainshaft_helpers.jl:52, in place), then OrdinaryDiffEq's SSPRK33 update, first half:
            // w = u_prev (clamped) | 3 u_prev + u | u_prev + 2 u
#pragma unroll
            for (int m = 0; m < N; ++m)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    u[m][q] = u[m][q] < 0.0 ? 0.0 : u[m][q];
                    if (stage == 0) up[m][q] = u[m][q];
                    w[m][q] = stage == 0 ? u[m][q] : stage == 1 ? 3.0 * up[m][q] + u[m][q] : up[m][q] + 2.0 * u[m][q];
                }
            if (kParkState) {
#pragma unroll
                for (int r = 0; r < 3 * N; ++r) {
                    sh_flux[r][ps] = w[r / 3][r % 3];
                    if (r < NUP) sh_up[r][ps] = up[r / 3][r % 3];
                }
            }
#pragma unroll
            for (int m = 0; m < N; ++m) {
                nn[m] = 0.0;
                th[m] = 1.0;
                kk[m] = 1.0;
            }
            if (active1) {
#pragma unroll
                for (int m = 0; m < N; ++m) {
                    const double m0 = div_by_const(u[m][0], As.norm[3 * m + 0], As.inv_norm[3 * m + 0]);
                    const double m1 = div_by_const(u[m][1], As.norm[3 * m + 1], As.inv_norm[3 * m + 1]);
                    const double m2 = div_by_const(u[m][2], As.norm[3 * m + 2], As.inv_norm[3 * m + 2]);
                    all_small = all_small && (m0 < kEps) && (m1 < kEps) && (As.np[m] != 3 || m2 < kEps);
                    invert_closure(As.dist_type[m], m0, m1, m2, As.kmin, As.kmax, nn[m], th[m], kk[m]);
                }
            }
            if (kPark) sh_small[ps] = all_small ? 1 : 0;
            size_t opaque_zero2 = 0;
            if (!SPEC) asm volatile("" : "+s"(opaque_zero2));
            const KArgs<N, P> &Ac = *(Ag + opaque_zero2);
            double acc[N][3];
            double (*fx)[BS] = sh_flux;  // the flux-exchange rows of this stage
            if (MODE == MODE_ALLINF) {
                if (active) coal_ints_parcel<N, P, MODE, false, SPEC>(Ac, nodes, nn, th, kk, acc);
            } else {
                // every lane of the workgroup: barriers inside.  (n, theta, k) come back from the lane's LDS slots, and those
                // 3N rows -- read by nobody but their owner once the passes are done -- then carry the flux exchange
                coal_ints_ranked_impl<N, P, MODE, SPEC, BS, true>(Ac, nodes, active1, nn, th, kk, acc, fx);
            }
            // (the lane's place in its column is re-derived after the passes, not kept across them)
            int pp = threadIdx.x;
            if (kPark) asm volatile("" : "+v"(pp));
            const int cl2 = pp / nz, iz2 = pp - cl2 * nz;
            const bool active2 = kPark ? (cl2 < cpb) && ((size_t)blockIdx.x * cpb + cl2 < n_columns) : active;
            const bool small2 = kPark ? sh_small[pp] != 0 : all_small;
            size_t opaque_zero3 = 0;
            if (!SPEC) asm volatile("" : "+s"(opaque_zero3));
            const KArgs<N, P> &Af = *(Ag + opaque_zero3);
            const SediArgs &S = *(Sg + opaque_zero3);
            if (active2) {
                double fl[N][3];
                sedi_flux_parcel<N>(Af.dist_type, S, nn, th, kk, fl);
#pragma unroll
                for (int m = 0; m < N; ++m)
#pragma unroll
                    for (int q = 0; q < 3; ++q)  // (the lane reads its own flux back after the barrier as well: not kept)
                        fx[3 * m + q][pp] = (q < Af.np[m]) ? fl[m][q] * Af.out_scale[3 * m + q] : 0.0;
            }
            CLOUDY_STAGE_BARRIER();
            if (kParkState) {
#pragma unroll
                for (int r = 0; r < 3 * N; ++r) {
                    w[r / 3][r % 3] = sh_flux[r][pp];
                    if (r < NUP) up[r / 3][r % 3] = sh_up[r][pp];
                }
            }
#pragma unroll
            for (int m = 0; m < N; ++m)  // (a lane without a cell: so that no lane's u is live across the passes)
                u[m][0] = u[m][1] = u[m][2] = 0.0;
            if (active2) {
                const bool top = (iz2 == nz - 1);  // zero flux above the top cell (:80-81)
                const double c = stage == 2 ? 2.0 * dt : dt;
#pragma unroll
                for (int m = 0; m < N; ++m)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const double f_up = top ? 0.0 : fx[3 * m + q][pp + 1];
                        const double fd = -(f_up - fx[3 * m + q][pp]) / dz;  // :83-85
                        // coal_source .+ sedi_source (:88), empty cells skip coalescence (:67-72)
                        const double ft = ((q < Af.np[m] && !small2) ? acc[m][q] * Af.out_scale[3 * m + q] : 0.0) + fd;
                        if (RHS_ONLY) {
                            u[m][q] = ft;
                        } else {
                            const double r = w[m][q] + c * ft;  // u_prev + dt f | 3 u_prev + u + dt f | u_prev + 2 u + 2 dt f
                            u[m][q] = stage == 0 ? r : stage == 1 ? r * 0.25 : div_by_const(r, 3.0, 1.0 / 3.0);
                        }
                    }
                if (RHS_ONLY && flux_out != nullptr) {   // cloudy_rainshaft_rhs's flux_work_dev: the cell fluxes
                    const size_t ie2 = ((size_t)blockIdx.x * cpb + cl2) * (size_t)nz + iz2;
#pragma unroll
                    for (int m = 0; m < N; ++m)
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            if (q < Af.np[m]) flux_out[(size_t)(Af.off[m] + q) * ld + ie2] = (TIO)fx[3 * m + q][pp];
                }
            }
            CLOUDY_STAGE_BARRIER();
        }
    }
    int pe = threadIdx.x;
    if (kPark) asm volatile("" : "+v"(pe));
    const int cle = pe / nz;
    const size_t cole = (size_t)blockIdx.x * cpb + cle;
    if ((cle < cpb) && (cole < n_columns)) {
        const size_t ie = cole * (size_t)nz + (pe - cle * nz);
#pragma unroll
        for (int m = 0; m < N; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < A.np[m]) {
                    const double v = (!RHS_ONLY && n_steps > 0 && u[m][q] < 0.0) ? 0.0 : u[m][q];  // the FSAL evaluation's clamp
                    u_out[(size_t)(A.off[m] + q) * ld + ie] = (TIO)v;
                }
    }
}

template <int N, int P, int MODE, typename TIO>
__global__ void __launch_bounds__(kRainshaftBlock)
    rainshaft_ssprk33_kernel(const KArgs<N, P> *__restrict__ Ag, const double *__restrict__ nodes, int nz,
                             size_t n_columns, size_t ld, const TIO *u_in, TIO *u_out, double dt, double dz,
                             int n_steps) {
    // SediArgs sits directly behind KArgs in the plan's constant block (launch_impl.hpp, OP_PREPARE)
    rainshaft_ssprk33_body<N, P, MODE, TIO, false>(Ag, reinterpret_cast<const SediArgs *>(Ag + 1), nodes, nz, n_columns, ld,
                                                   u_in, u_out, dt, dz, n_steps);
}

// get_cond_evap, src/Sources/Condensation.jl:22-37, behind rhs_condensation! (box_model_helpers.jl:55-67):
//   d(mom_j)/dt = 3 xi s j M_{j - 2/3} (4 pi/3)^(2/3) / rho_l^(1/3)   (0-based order j; zero for j = 0)
// `coef` = 3 xi_normalised (4 pi/3)^(2/3) / rho_l^(1/3) is folded on the host; s is a scalar or one value per parcel.
template <int N, int P, typename TIO>
__device__ __forceinline__ void cond_evap_body(const KArgs<N, P> &A, double coef, double s_scalar, const double *__restrict__ s_dev, size_t n,
                     size_t ld, const TIO *__restrict__ in, TIO *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        double nn[N], th[N], kk[N];
        load_parcel<N, P, TIO>(A, i, ld, in, nn, th, kk);
        const double sv = s_dev ? s_dev[i] : s_scalar;
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const int off = A.off[m];
            const int dtp = A.dist_type[m];
            const bool gam = (dtp == DIST_GAMMA || dtp == DIST_EXP);
            const double lnth = (dtp == DIST_LOGNORMAL) ? th[m] : log(th[m]);
            out[(size_t)off * ld + i] = (TIO)0.0;
            // M_{1/3} from the log-gamma ratio, M_{4/3} = M_{1/3} theta (k + 1/3) for the Gamma family
            double mom = 0.0;
            for (int j = 1; j < A.np[m]; ++j) {
                const double q = double(j) - 2.0 / 3.0;
                if (gam)
                    mom = (j == 1) ? nn[m] * exp(fma(q, lnth, log_gamma_ratio(kk[m], q)))
                                   : mom * (th[m] * (kk[m] + (q - 1.0)));
                else
                    mom = nn[m] * exp((dtp == DIST_MONO) ? q * lnth : fma(q, lnth, 0.5 * q * q * (kk[m] * kk[m])));
                out[(size_t)(off + j) * ld + i] = (TIO)(coef * sv * double(j) * mom * A.out_scale[3 * m + j]);
            }
        }
    }
}
template <int N, int P, typename TIO>
__global__ void __launch_bounds__(kBlock)
    cond_evap_kernel(const KArgs<N, P> A, double coef, double s_scalar, const double *__restrict__ s_dev, size_t n,
                     size_t ld, const TIO *__restrict__ in, TIO *__restrict__ out) {
    cond_evap_body<N, P, TIO>(A, coef, s_scalar, s_dev, n, ld, in, out);
}

// get_standard_N_q, ParticleDistributions.jl:634-687 (cloud / rain diagnostics by a size cutoff): 4 planes
// (N_liq, N_rai, M_liq, M_rai) in physical units from partial_moment (:226-285):
//   Gamma / Exponential: n theta^q Gamma(q+k)/Gamma(k) P(q+k, x_c/theta) = M_q P(q+k, x_c/theta)
//   Monodisperse:        M_q if x_c >= theta else 0
template <int N, int P, typename TIO>
__device__ __forceinline__ void standard_nq_body(const KArgs<N, P> &A, double cutoff_n, double n0, double m0, size_t n, size_t ld,
                       const TIO *__restrict__ in, TIO *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        double nn[N], th[N], kk[N];
        load_parcel<N, P, TIO>(A, i, ld, in, nn, th, kk);
        double Nl = 0.0, Nr = 0.0, Ml = 0.0, Mr = 0.0;
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const int dtp = A.dist_type[m];
            const double M0 = nn[m];
            double M1, p0, p1;
            if (dtp == DIST_MONO) {
                M1 = nn[m] * th[m];
                p0 = (cutoff_n < th[m]) ? 0.0 : 1.0;
                p1 = p0;
            } else if (dtp == DIST_LOGNORMAL) {
                // partial_moment (:255-269, quadgk in the reference) in closed form:
                // int_0^xc x^q n f = n exp(q mu + q^2 sigma^2 / 2) Phi((ln xc - mu - q sigma^2) / sigma)
                M1 = nn[m] * exp(th[m] + 0.5 * kk[m] * kk[m]);
                const double w = (log(cutoff_n) - th[m]) / kk[m];
                p0 = norm_cdf(w);
                p1 = norm_cdf(w - kk[m]);
            } else {
                M1 = nn[m] * th[m] * kk[m];
                const double z = cutoff_n / th[m];
                const double a1 = kk[m] + 1.0;  // P(k+1, z), then P(k, z) = P(k+1, z) + z^k e^-z / Gamma(k+1)
                const double lg = lgamma(a1 + 1.0);
                const double E1 = exp(a1 * log(z) - z - lg);
                p1 = (z > 0.0) ? inc_gamma_p_from_E(a1, z, E1, nullptr) : 0.0;
                p0 = (z > 0.0) ? p1 + E1 * a1 / z : 0.0;
                p0 = p0 > 1.0 ? 1.0 : p0;
            }
            Nl += M0 * p0;
            Nr += M0 - M0 * p0;
            Ml += M1 * p1;
            Mr += M1 - M1 * p1;
        }
        out[(size_t)0 * ld + i] = (TIO)(Nl * n0);
        out[(size_t)1 * ld + i] = (TIO)(Nr * n0);
        out[(size_t)2 * ld + i] = (TIO)(Ml * n0 * m0);
        out[(size_t)3 * ld + i] = (TIO)(Mr * n0 * m0);
    }
}
template <int N, int P, typename TIO>
__global__ void __launch_bounds__(kBlock)
    standard_nq_kernel(const KArgs<N, P> A, double cutoff_n, double n0, double m0, size_t n, size_t ld,
                       const TIO *__restrict__ in, TIO *__restrict__ out) {
    standard_nq_body<N, P, TIO>(A, cutoff_n, n0, m0, n, ld, in, out);
}

}  // namespace cloudy
