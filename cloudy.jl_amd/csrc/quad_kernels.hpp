// quad_kernels.hpp -- the kernel behind cloudy_coal_rhs / cloudy_get_coal_ints of a NumericalCoalStyle plan
// (quad.hpp states the rule; reference: src/Sources/Coalescence.jl:470-708).
//
// One lane owns one parcel, as everywhere (kernels.hpp): load -> normalise -> closure inversion -> the Gauss rule of
// every mode (nodes and weights in registers when the kernel is compiled for its plan) -> pair sums -> store.
//
// Algebra.  With Kab = K(x^j_a, x^k_b) v^j_a v^k_b (v = n W) and, per unordered pair of modes,
//     s0 = sum Kab,  sa = sum Kab x^j_a,  saa = sum Kab (x^j_a)^2,  sab = sum Kab x^j_a x^k_b,
// the reference's  sum_j Q[m][j,k] - sum_j R[m][j,k]  (:478-487) is, order by order (m = 0, 1, 2),
//     pair j < k:   mode k gains (0, sa, saa + 2 sab)   [Q_jk - R_jk: the terms in x^k_b alone cancel],
//                   mode j loses (s0, sa, saa)          [R_kj],
//     pair j = k:   S_1 + S_2 - R_kk = (-s0/2, 0, sab)  [mass conservation is exact in the discrete rule],
// and the self collisions that weighting_fn (:624-642) assigns to the next mode,
//     T_m = S_2k^(m) = 1/2 sum_ab Kab (x_a + x_b)^m (1 - w(x_a + x_b, k)),
// move from mode k to mode k + 1 (for the last mode w == 1).  The products common to Q and R are never formed, as in
// the tensor kernels (kernels.hpp, pair_terms).  1 - w is evaluated from density RATIOS against the mode's own density,
//     1 - w = sum_{j > k} rho_j / (1 + sum_{j != k} rho_j),   rho_j = g_j(x) / g_k(x) = exp(ln g_j - ln g_k),
// which needs N - 1 exponentials per point instead of N and cannot underflow to 0 / 0 at a mode's own nodes.
#pragma once
#include "kernels.hpp"
#include "quad.hpp"
#include "quad_conv.hpp"

namespace cloudy {

// ln of the normed density (ParticleDistributions.jl:363-388) at x given lx = ln x:
//   Gamma / Exponential: (k - 1) lx - x / theta - [ln Gamma(k) + k ln theta];  Lognormal (theta = mu, k = sigma):
//   -(lx - mu)^2 / (2 sigma^2) - lx - ln(sigma sqrt(2 pi))
struct LogDensity {
    double a, b, c;   // Gamma: a = k - 1, b = 1 / theta, c = lnGamma(k) + k ln theta;  Lognormal: a = mu, b = 1/(2 sigma^2), c = ln(sigma sqrt(2 pi))
    bool lognormal;
    __device__ __forceinline__ double operator()(double x, double lx) const {
        if (lognormal) {
            const double d = lx - a;
            return -(d * d) * b - lx - c;
        }
        return fma(a, lx, -(x * b)) - c;
    }
};

#ifndef CLOUDY_QUAD_SELF_UNROLL
#define CLOUDY_QUAD_SELF_UNROLL 2  // inner loop of the self-collision points: two points in flight per trip (8.44 -> 8.22 ms on the cfg4q batch; 3: the same)
#endif
// get_coal_ints(::NumericalCoalStyle, ...) for the parcel of this lane (Coalescence.jl:470-489): acc[k][m] in normalised
// units, WITHOUT the kernel function's constant factor kf_scale<KIND>(Q) (the caller folds it into its output scale).
template <int N, int KIND, int NQ>
__device__ __forceinline__ void quad_coal_ints(const KArgs<N, 1> &A, const QArgs &Q, const double *__restrict__ tab,
                                               const double (&nn)[N], const double (&th)[N], const double (&kk)[N],
                                               double (&acc)[N][3]) {
    constexpr int NQA = NQ ? NQ : kQuadMax;
    constexpr int QB = NQ ? quad_block(NQ) : kBlock;
    const int nq = NQ ? NQ : Q.nq;
    const int t = threadIdx.x;
    // ---- the rule of every mode: aux (x or x^(1/3)) and v = n W
    double X[N][NQA], V[N][NQA];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        if (A.dist_type[m] == DIST_LOGNORMAL) {  // wave-uniform
            const double *gh = tab + nq * (Q.deg + 1);
            const double s2 = 1.4142135623730951 * kk[m];
#pragma unroll
            for (int a = 0; a < nq; ++a) {
                X[m][a] = kf_aux<KIND>(exp_fin(fma(s2, gh[a], th[m])));
                V[m][a] = nn[m] * gh[nq + a];
            }
        } else {
            double u[NQA], W[NQA];
            gamma_rule<NQ>(Q, tab, kk[m], u, W);
#pragma unroll
            for (int a = 0; a < nq; ++a) {
                X[m][a] = kf_aux<KIND>(th[m] * u[a]);
                V[m][a] = nn[m] * W[a];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0;
    LogDensity lg[N];
    if (N > 1) {
#pragma unroll
        for (int m = 0; m < N; ++m) {
            lg[m].lognormal = A.dist_type[m] == DIST_LOGNORMAL;
            if (lg[m].lognormal) {
                lg[m].a = th[m];
                lg[m].b = 0.5 / (kk[m] * kk[m]);
                lg[m].c = log_pos(kk[m] * 2.5066282746310002);
            } else {
                lg[m].a = kk[m] - 1.0;
                lg[m].b = 1.0 / th[m];
                lg[m].c = fma(kk[m], log_pos(th[m]), lgamma_pos(kk[m]));
            }
        }
    }
    // Mode j in turn is the OUTER mode: its nodes are parked in the lane's own LDS slots (sh_x[a][t]: conflict-free, no
    // barrier -- a lane only reads what it wrote) and walked by ROLLED loops, first against itself (the self collisions:
    // a log and N - 1 exponentials per point -- unrolled that is ~150 KB of code), then against every mode k > j, whose
    // nodes are register operands of the unrolled inner loop.  Rolled outer loops keep the code small and stop the
    // compiler from hoisting per-node subexpressions of ALL modes at once (measured: 350-500 VGPRs, up to 1400 spills
    // for the Long family, when everything was unrolled).  After its turn a mode's registers are dead.  The
    // ahead-of-time kernel (run-time point count) indexes its scratch-resident arrays instead of LDS.
    __shared__ double sh_x[NQ ? NQ : 1][QB], sh_v[NQ ? NQ : 1][QB];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (NQ) {
#pragma unroll
            for (int a = 0; a < nq; ++a) {
                sh_x[a][t] = X[j][a];
                sh_v[a][t] = V[j][a];
            }
        }
        // ---- self collisions of mode j
        {
            LogDensity dlg[N];  // parameters of ln g_m - ln g_j (Gamma family)
#pragma unroll
            for (int m = 0; m < N; ++m) {
                dlg[m].a = lg[m].a - lg[j].a;
                dlg[m].b = lg[m].b - lg[j].b;
                dlg[m].c = lg[m].c - lg[j].c;
                dlg[m].lognormal = false;
            }
            double s0 = 0.0, sab = 0.0, T0 = 0.0, T1 = 0.0, T2 = 0.0;
#pragma unroll 1
            for (int a = 0; a < nq; ++a) {
                const double Xa = NQ ? sh_x[a][t] : X[j][a], va = NQ ? sh_v[a][t] : V[j][a];
                const double xa = kf_x<KIND>(Xa);
#pragma unroll CLOUDY_QUAD_SELF_UNROLL
                for (int b = a; b < nq; ++b) {
                    const double Xb = NQ ? sh_x[b][t] : X[j][b], vb = NQ ? sh_v[b][t] : V[j][b];
                    const double xb = kf_x<KIND>(Xb);
                    // Kab + Kba = 2 Kab for a < b (K is symmetric)
                    const double Kab = kf_eval<KIND>(Q, Xa, Xb) * (va * vb) * (a == b ? 1.0 : 2.0);
                    s0 += Kab;
                    sab = fma(Kab, xa * xb, sab);
                    if (j < N - 1) {
                        const double xs = xa + xb, lx = log_pos(xs);
                        const double own = lg[j].lognormal ? lg[j](xs, lx) : 0.0;
                        double up = 0.0, den = 1.0;
#pragma unroll
                        for (int m = 0; m < N; ++m) {
                            if (m == j) continue;
                            // ln g_m - ln g_j; between two Gamma-family modes with the parameter differences formed
                            // once per pair of modes (two FMAs per point, and no cancellation between the two logs)
                            const double e = (lg[m].lognormal || lg[j].lognormal)
                                                 ? lg[m](xs, lx) - (lg[j].lognormal ? own : lg[j](xs, lx))
                                                 : fma(dlg[m].a, lx, fma(-dlg[m].b, xs, -dlg[m].c));
                            const double rho = exp_fin(fmin(e, 700.0));
                            den += rho;
                            if (m > j) up += rho;
                        }
                        const double h = 0.5 * Kab * (up * quad_recip(den));  // 1/2 Kab (1 - w)
                        T0 += h;
                        T1 = fma(h, xs, T1);
                        T2 = fma(h * xs, xs, T2);
                    }
                }
            }
            acc[j][0] -= fma(0.5, s0, T0);
            acc[j][1] -= T1;
            acc[j][2] += sab - T2;
            if (j < N - 1) {
                acc[j + 1][0] += T0;
                acc[j + 1][1] += T1;
                acc[j + 1][2] += T2;
            }
        }
        // ---- pairs (j, k > j)
#pragma unroll
        for (int k = j + 1; k < N; ++k) {
            double s0 = 0.0, sa = 0.0, saa = 0.0, sab = 0.0;
#pragma unroll 1
            for (int a = 0; a < nq; ++a) {
                const double Xa = NQ ? sh_x[a][t] : X[j][a], va = NQ ? sh_v[a][t] : V[j][a];
                const double xa = kf_x<KIND>(Xa);
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int b = 0; b < nq; ++b) {
                    const double Kv = kf_eval<KIND>(Q, Xa, X[k][b]) * V[k][b];
                    t0 += Kv;
                    t1 = fma(Kv, kf_x<KIND>(X[k][b]), t1);
                }
                const double w0 = va * t0, w1 = w0 * xa;
                s0 += w0;
                sa += w1;
                saa = fma(w1, xa, saa);
                sab = fma(va * xa, t1, sab);
            }
            acc[j][0] -= s0;
            acc[j][1] -= sa;
            acc[j][2] -= saa;
            acc[k][1] += sa;
            acc[k][2] += fma(2.0, sab, saa);
        }
    }
}

// CONV: the converged mode of quad_conv.hpp (closed forms + one 1-D rule per mode; kernel constants included in acc)
//
// hint (converged mode, kernels compiled for the plan; may be null): ONE BYTE PER PARCEL of the plan's scratch -- the number of
// panel evaluations the parcel's adaptive rules took the last time this plan evaluated parcel i (0: never).  The adaptive
// walk makes a wave run as long as its lane with the most evaluations: 0.79 active lanes on the cfg4q batch (round 4 PMC), and
// no cheap a-priori predictor of a parcel's cost exists (the number of initial panels correlates 0.51 with it;
// tools/conv_lab/experiments.py sim_sort).  The cost a parcel HAD is one: an ODE solver calls the operator on the same
// parcels again and again (three times per SSPRK33 step) and the state moves by a fraction of a per cent between calls.  So
// the workgroup ranks its 256 parcels by their hints (the counting sort of the threshold kernels) and lane t takes the parcel
// of rank t -- waves of parcels of EQUAL cost: ranking by the true cost the simulation gives 0.93 lanes for workgroups of
// 256.  Loads and stores are a permutation inside the workgroup's 2-KB window of each plane.  Which lane computes a parcel
// changes nothing in its result (parcels are independent; the wave-level votes of the walk only skip code that is a no-op for
// every lane), so results are bit-identical with and without hints, and a stale or foreign hint costs balance, never accuracy.
// the parcel this lane takes: the natural one, or -- hints given -- the parcel of rank threadIdx.x among the workgroup's parcels
// ordered by their hint bytes (every lane of the workgroup must call it: barriers)
// hint2 (Long plans; round 6): a second byte per parcel.  The Long kernel's walk has two loops (quad_conv.hpp, PHASE) whose trip
// counts are nearly independent: byte 1 holds phase 1's count, byte 2 phase 2's.  The kernel behind cloudy_coal_rhs ranks its
// workgroup once per loop (ConvSplit); the fused integrators, which rank once per call, use round 5's key: 4 bits of phase 2's
// count first (it decides whether a wave enters the second loop at all), 4 bits of phase 1's (in units of four) second.
template <int QB>
__device__ __forceinline__ size_t quad_ranked_parcel(size_t n, const unsigned char *__restrict__ hint, unsigned int *sh_cnt,
                                                     unsigned short *sh_perm, const unsigned char *__restrict__ hint2 = nullptr,
                                                     bool key44 = false) {
    size_t i = (size_t)blockIdx.x * QB + threadIdx.x;
    if (hint != nullptr) {   // (a kernel argument: the same for every lane)
        const bool valid = i < n;
        int hb = valid ? (int)hint[i] : 0;
        if (key44 && hint2 != nullptr) {
            const int p2 = valid ? (int)hint2[i] : 0, p1 = hb >> 2;
            hb = ((p2 > 15 ? 15 : p2) << 4) | (p1 > 15 ? 15 : p1);
        }
        regime_rank<QB>(valid, hb > QB - 2 ? QB - 2 : hb, sh_cnt, sh_perm);
        i = (size_t)blockIdx.x * QB + sh_perm[threadIdx.x];   // (lanes without a parcel rank last)
    }
    return i;
}
template <int QB>
__device__ __forceinline__ size_t quad_ranked_parcel(size_t n, const unsigned char *__restrict__ hint,
                                                     const unsigned char *__restrict__ hint2 = nullptr, bool key44 = false) {
    __shared__ unsigned int sh_cnt[QB];
    __shared__ unsigned short sh_perm[QB];
    return quad_ranked_parcel<QB>(n, hint, sh_cnt, sh_perm, hint2, key44);
}
// what a parcel leaves behind for the next call from the `cost` of conv_coal_ints (Long: phase 2's count in the upper half)
template <int KIND>
__device__ __forceinline__ void quad_store_hints(unsigned char *__restrict__ hint, unsigned char *__restrict__ hint2, size_t i,
                                                 int cost) {
    if (KIND == KF_LONG && hint2 != nullptr) {
        const int p1 = cost & 0xffff, p2 = cost >> 16;
        hint[i] = (unsigned char)(p1 > 255 ? 255 : p1);
        hint2[i] = (unsigned char)(p2 > 255 ? 255 : p2);
    } else {
        hint[i] = conv_hint_byte<KIND>(cost);
    }
}

// ROUNDS (round 6, an experiment switch: CLOUDY_HIP_CONV_ROUNDS): a workgroup of QB threads takes ROUNDS x QB parcels, ranks ALL of
// them by their hints, and wave w walks the groups of 64 in the order w, 2W - 1 - w, 2W + w, ... (W waves): the cheapest group
// with the dearest one, so that the waves of a workgroup finish together.  A workgroup keeps its LDS and its wave slots until its
// LAST wave ends, and the ranking makes the waves of a workgroup as different as it can: with one group per wave the slots of the
// cheap waves idle (VALU busy 0.93 at 256 threads, 0.78 at 512, 28 ms at 1024 where ranking alone predicts the opposite order).
// No barrier but the ranking's; results do not depend on which lane or round computes a parcel.
template <int QB, int ROUNDS>
__device__ __forceinline__ void quad_rank_rounds(size_t n, size_t base, const unsigned char *__restrict__ hint,
                                                 const unsigned char *__restrict__ hint2, bool key44, unsigned int *sh_cnt /*[QB]*/,
                                                 unsigned short *sh_perm /*[ROUNDS * QB]*/) {
    static_assert(QB >= 256, "one counter per hint value");
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    sh_cnt[t] = 0u;
    __syncthreads();
    unsigned int pos[ROUNDS];
    int bucket[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const size_t i = base + (size_t)r * QB + t;
        const bool valid = i < n;
        int hb = (valid && hint != nullptr) ? (int)hint[i] : 0;
        if (key44 && hint2 != nullptr) {
            const int p2 = valid ? (int)hint2[i] : 0, p1 = hb >> 2;
            hb = ((p2 > 15 ? 15 : p2) << 4) | (p1 > 15 ? 15 : p1);
        }
        bucket[r] = valid ? (hb > QB - 2 ? QB - 2 : hb) : QB - 1;
        pos[r] = atomicAdd(&sh_cnt[bucket[r]], 1u);
    }
    __syncthreads();
    const unsigned int c = sh_cnt[t];
    unsigned int incl = c;
    const int lane = t & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int up = (unsigned int)__builtin_amdgcn_ds_bpermute(((lane - d) & 63) << 2, (int)incl);
        if (lane >= d) incl += up;
    }
    __shared__ unsigned int sh_wave_tot[QB / 64];
    if (lane == 63) sh_wave_tot[t >> 6] = incl;
    __syncthreads();
    unsigned int offs = 0;
#pragma unroll
    for (int w = 0; w < QB / 64; ++w)
        if (w < (t >> 6)) offs += sh_wave_tot[w];
    sh_cnt[t] = offs + incl - c;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) sh_perm[sh_cnt[bucket[r]] + pos[r]] = (unsigned short)(r * QB + t);
    __syncthreads();
}

template <int N, int KIND, typename TIO, int ROUNDS>
__device__ __forceinline__ void coal_rhs_quad_rounds(const KArgs<N, 1> &A, const QArgs &Q, const double *__restrict__ tab, size_t n,
                                                     size_t ld, const TIO *__restrict__ in, TIO *__restrict__ out,
                                                     unsigned char *__restrict__ hint, unsigned char *__restrict__ hint2) {
    constexpr int QB = kConvBlock, W = QB / 64;
    __shared__ unsigned int sh_cnt[QB];
    __shared__ unsigned short sh_perm[ROUNDS * QB];
    const size_t base = (size_t)blockIdx.x * (QB * ROUNDS);
    quad_rank_rounds<QB, ROUNDS>(n, base, hint, hint2, true, sh_cnt, sh_perm);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll 1
    for (int r = 0; r < ROUNDS; ++r) {
        const int group = (r & 1) ? (r + 1) * W - 1 - w : r * W + w;   // w, 2W - 1 - w, 2W + w, 4W - 1 - w, ...
        const size_t i = base + sh_perm[group * 64 + l];
        if (i < n) {
            double nn[N], th[N], kk[N], acc[N][3];
            load_parcel<N, 1, TIO>(A, i, ld, in, nn, th, kk);
            int cost = 0;
            conv_coal_ints<N, KIND>(A, Q, tab, nn, th, kk, acc, cost);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int off = A.off[k];
                out[(size_t)(off + 0) * ld + i] = (TIO)(acc[k][0] * A.out_scale[3 * k + 0]);
                out[(size_t)(off + 1) * ld + i] = (TIO)(acc[k][1] * A.out_scale[3 * k + 1]);
                if (A.np[k] == 3) out[(size_t)(off + 2) * ld + i] = (TIO)(acc[k][2] * A.out_scale[3 * k + 2]);
            }
            if (hint != nullptr) quad_store_hints<KIND>(hint, hint2, i, cost);
        }
    }
}

template <int N, int KIND, int NQ, typename TIO, bool CONV = false, bool SPLIT = false, int ROUNDS = 1>
__device__ __forceinline__ void coal_rhs_quad_body(const KArgs<N, 1> &A, const QArgs &Q, const double *__restrict__ tab,
                                                   size_t n, size_t ld, const TIO *__restrict__ in,
                                                   TIO *__restrict__ out, unsigned char *__restrict__ hint = nullptr,
                                                   unsigned char *__restrict__ hint2 = nullptr) {
    constexpr int QB = NQ ? quad_block(NQ) : (CONV ? kConvBlock : kBlock);
    static_assert(!SPLIT || (CONV && KIND == KF_LONG), "SPLIT: the Long kernel's converged mode");
    static_assert(ROUNDS == 1 || (CONV && !SPLIT), "ROUNDS: the barrier-free converged kernels");
    if constexpr (ROUNDS > 1) {
        coal_rhs_quad_rounds<N, KIND, TIO, ROUNDS>(A, Q, tab, n, ld, in, out, hint, hint2);
        return;
    }
    __shared__ unsigned int sh_cnt[CONV ? QB : 1];
    __shared__ unsigned short sh_perm[CONV ? QB : 1];
    const size_t i = quad_ranked_parcel<QB>(n, CONV ? hint : nullptr, sh_cnt, sh_perm, hint2, !SPLIT);
    // (SPLIT: the workgroup meets again between the two loops of the walk -- lanes without a parcel stay, with an empty one)
    const bool valid = i < n;
    if (!SPLIT && !valid) return;
    double nn[N], th[N], kk[N], acc[N][3];
    load_parcel<N, 1, TIO>(A, valid ? i : 0, ld, in, nn, th, kk);
    int cost = 0;
    if (SPLIT) {
        if (!valid) {
#pragma unroll
            for (int m = 0; m < N; ++m) nn[m] = 0.0;
        }
        ConvSplit sp;
        sp.valid = valid;
        sp.p2_hint = (valid && hint2 != nullptr) ? (int)hint2[i] : 0;
        sp.sh_cnt = sh_cnt;
        sp.sh_perm = sh_perm;
        sp.cost2 = 0;
        conv_coal_ints<N, KIND, SPLIT>(A, Q, tab, nn, th, kk, acc, cost, &sp);
        cost = (cost & 0xffff) | (sp.cost2 << 16);
        if (!valid) return;
    } else if (CONV) {
        conv_coal_ints<N, KIND>(A, Q, tab, nn, th, kk, acc, cost);
    } else {
        quad_coal_ints<N, KIND, NQ>(A, Q, tab, nn, th, kk, acc);
    }
    const double ksc = CONV ? 1.0 : kf_scale<KIND>(Q);
    // A ranked workgroup stores a PERMUTATION of its 2-KB window of each plane: a wave's 64 values land in 64 different sectors,
    // and the four waves of the workgroup finish milliseconds apart.  As nontemporal stores (round 5) every 8-B value went to
    // HBM on its own: WRITE_SIZE 3.3 x the output (round 6 PMC: 945 MB for 288 MB of tendencies).  Plain stores let the L2
    // assemble the lines before they leave.
    const auto store = [&](TIO *p, double v) {
        if (CONV)
            *p = (TIO)v;
        else
            st_stream(p, v);
    };
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int off = A.off[k];
        store(out + (size_t)(off + 0) * ld + i, acc[k][0] * (ksc * A.out_scale[3 * k + 0]));
        store(out + (size_t)(off + 1) * ld + i, acc[k][1] * (ksc * A.out_scale[3 * k + 1]));
        if (A.np[k] == 3) store(out + (size_t)(off + 2) * ld + i, acc[k][2] * (ksc * A.out_scale[3 * k + 2]));
    }
    if (CONV && hint != nullptr) quad_store_hints<KIND>(hint, hint2, i, cost);
}

// solve(ODEProblem(make_box_model_rhs(NumericalCoalStyle()), m, tspan, p), SSPRK33(), dt) -- the Numerical drivers
// (test/examples/Numerical/n_particles_gamma.jl:39-40, single_particle_gamma.jl:41-42) -- with the state in registers
// over all stages and steps, as ssprk33_body (kernels.hpp) does for the tensor plans: one read and one write of the state
// per call, OrdinaryDiffEq's SSPRK33 update formulas ("/4" exact, "/3" correctly rounded).
template <int N, int KIND, int NQ, typename TIO, bool CONV = false>
__device__ __forceinline__ void quad_ssprk33_body(const KArgs<N, 1> &A, const QArgs &Q, const double *__restrict__ tab,
                                                  size_t n, size_t ld, const TIO *u_in, TIO *u_out, double dt,
                                                  int n_steps, unsigned char *__restrict__ hint = nullptr,
                                                  unsigned char *__restrict__ hint2 = nullptr) {
    // (converged mode: the lane keeps, for the whole call, the parcel its hint ranks it to -- the cost of a parcel moves little
    // over a few stages -- and leaves the cost of its last evaluation behind; coal_rhs_quad_body shares the bytes)
    constexpr int QB = NQ ? quad_block(NQ) : (CONV ? kConvBlock : kBlock);
    const size_t i = quad_ranked_parcel<QB>(n, CONV ? hint : nullptr, hint2, true);
    if (i >= n) return;
    int cost = 0;
    double u[N][3], up[N][3];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const int off = A.off[m];
        u[m][0] = (double)u_in[(size_t)(off + 0) * ld + i];
        u[m][1] = (double)u_in[(size_t)(off + 1) * ld + i];
        u[m][2] = (A.np[m] == 3) ? (double)u_in[(size_t)(off + 2) * ld + i] : 0.0;
    }
    const double ksc = CONV ? 1.0 : kf_scale<KIND>(Q);
#pragma unroll 1
    for (int step = 0; step < n_steps; ++step) {
#pragma unroll
        for (int m = 0; m < N; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) up[m][q] = u[m][q];
#pragma unroll 1
        for (int stage = 0; stage < 3; ++stage) {
            double nn[N], th[N], kk[N], acc[N][3], f[N][3];
#pragma unroll
            for (int m = 0; m < N; ++m) {  // mom ./ mom_norms, update_dist_from_moments (as load_parcel)
                const double m0 = div_by_const(u[m][0], A.norm[3 * m + 0], A.inv_norm[3 * m + 0]);
                const double m1 = div_by_const(u[m][1], A.norm[3 * m + 1], A.inv_norm[3 * m + 1]);
                const double m2 = div_by_const(u[m][2], A.norm[3 * m + 2], A.inv_norm[3 * m + 2]);
                invert_closure(A.dist_type[m], m0, m1, m2, A.kmin, A.kmax, nn[m], th[m], kk[m]);
            }
            if (CONV) {
                // the state and the step's first value wait in the lane's own LDS slots across the walk (conflict-free, no
                // barrier: a lane reads what it wrote) instead of in 36 registers the walk has no room for -- round 5: the fused
                // converged integrator had 251 spilled registers and was 1.16 x SLOWER than three cloudy_coal_rhs launches
                // (not under the Long kernel: its incomplete-beta tables already take 48 KB per workgroup, and 110 KB would leave
                // one workgroup per CU)
                constexpr bool kPark = CONV && KIND != KF_LONG;
                __shared__ double sh_state[kPark ? 6 * N : 1][QB];
                if (kPark) {
                    const int t0 = threadIdx.x;
#pragma unroll
                    for (int m = 0; m < N; ++m)
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            sh_state[kPark ? 3 * m + q : 0][t0] = u[m][q];
                            sh_state[kPark ? 3 * N + 3 * m + q : 0][t0] = up[m][q];
                        }
                }
                conv_coal_ints<N, KIND>(A, Q, tab, nn, th, kk, acc, cost);
                if (kPark) {
                    int t1 = threadIdx.x;
                    asm volatile("" : "+v"(t1));
#pragma unroll
                    for (int m = 0; m < N; ++m)
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            u[m][q] = sh_state[kPark ? 3 * m + q : 0][t1];
                            up[m][q] = sh_state[kPark ? 3 * N + 3 * m + q : 0][t1];
                        }
                }
            } else {
                quad_coal_ints<N, KIND, NQ>(A, Q, tab, nn, th, kk, acc);
            }
#pragma unroll
            for (int m = 0; m < N; ++m) {
                f[m][0] = acc[m][0] * (ksc * A.out_scale[3 * m + 0]);
                f[m][1] = acc[m][1] * (ksc * A.out_scale[3 * m + 1]);
                f[m][2] = (A.np[m] == 3) ? acc[m][2] * (ksc * A.out_scale[3 * m + 2]) : 0.0;
            }
            if (stage == 0) {
#pragma unroll
                for (int m = 0; m < N; ++m)
#pragma unroll
                    for (int q = 0; q < 3; ++q) u[m][q] = up[m][q] + dt * f[m][q];
            } else if (stage == 1) {
#pragma unroll
                for (int m = 0; m < N; ++m)
#pragma unroll
                    for (int q = 0; q < 3; ++q) u[m][q] = (3.0 * up[m][q] + u[m][q] + dt * f[m][q]) * 0.25;
            } else {
#pragma unroll
                for (int m = 0; m < N; ++m)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        u[m][q] = div_by_const(up[m][q] + 2.0 * u[m][q] + 2.0 * dt * f[m][q], 3.0, 1.0 / 3.0);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const int off = A.off[m];
        u_out[(size_t)(off + 0) * ld + i] = (TIO)u[m][0];
        u_out[(size_t)(off + 1) * ld + i] = (TIO)u[m][1];
        if (A.np[m] == 3) u_out[(size_t)(off + 2) * ld + i] = (TIO)u[m][2];
    }
    if (CONV && hint != nullptr && n_steps > 0) quad_store_hints<KIND>(hint, hint2, i, cost);
}

// cloudy_tsit5_steps of a NumericalCoalStyle plan (round 4): the tableau of tsit5_advance (kernels.hpp) around the same
// right-hand side as quad_ssprk33_body.  Plan-time compiled only (jit.hpp part 4).
template <int N, int KIND, int NQ, typename TIO, bool CONV = false>
__device__ __forceinline__ void quad_tsit5_body(const KArgs<N, 1> &A, const QArgs &Q, const double *__restrict__ tab,
                                                size_t n, size_t ld, const TIO *u_in, TIO *u_out, double dt, int n_steps,
                                                unsigned char *__restrict__ hint = nullptr,
                                                unsigned char *__restrict__ hint2 = nullptr) {
    constexpr int QB = NQ ? quad_block(NQ) : (CONV ? kConvBlock : kBlock);
    const size_t i = quad_ranked_parcel<QB>(n, CONV ? hint : nullptr, hint2, true);   // (as quad_ssprk33_body)
    if (i >= n) return;
    int cost = 0;
    double u[N][3];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const int off = A.off[m];
        u[m][0] = (double)u_in[(size_t)(off + 0) * ld + i];
        u[m][1] = (double)u_in[(size_t)(off + 1) * ld + i];
        u[m][2] = (A.np[m] == 3) ? (double)u_in[(size_t)(off + 2) * ld + i] : 0.0;
    }
    const double ksc = CONV ? 1.0 : kf_scale<KIND>(Q);
    tsit5_advance<N>(u, dt, n_steps, [&](const double (&state)[N][3], double (&f)[N][3]) {
        double nn[N], th[N], kk[N], acc[N][3];
#pragma unroll
        for (int m = 0; m < N; ++m) {  // mom ./ mom_norms, update_dist_from_moments (as load_parcel)
            const double m0 = div_by_const(state[m][0], A.norm[3 * m + 0], A.inv_norm[3 * m + 0]);
            const double m1 = div_by_const(state[m][1], A.norm[3 * m + 1], A.inv_norm[3 * m + 1]);
            const double m2 = div_by_const(state[m][2], A.norm[3 * m + 2], A.inv_norm[3 * m + 2]);
            invert_closure(A.dist_type[m], m0, m1, m2, A.kmin, A.kmax, nn[m], th[m], kk[m]);
        }
        if (CONV)
            conv_coal_ints<N, KIND>(A, Q, tab, nn, th, kk, acc, cost);
        else
            quad_coal_ints<N, KIND, NQ>(A, Q, tab, nn, th, kk, acc);
#pragma unroll
        for (int m = 0; m < N; ++m) {
            f[m][0] = acc[m][0] * (ksc * A.out_scale[3 * m + 0]);
            f[m][1] = acc[m][1] * (ksc * A.out_scale[3 * m + 1]);
            f[m][2] = (A.np[m] == 3) ? acc[m][2] * (ksc * A.out_scale[3 * m + 2]) : 0.0;
        }
    });
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const int off = A.off[m];
        u_out[(size_t)(off + 0) * ld + i] = (TIO)u[m][0];
        u_out[(size_t)(off + 1) * ld + i] = (TIO)u[m][1];
        if (A.np[m] == 3) u_out[(size_t)(off + 2) * ld + i] = (TIO)u[m][2];
    }
    if (CONV && hint != nullptr && n_steps > 0) quad_store_hints<KIND>(hint, hint2, i, cost);
}

// ahead-of-time instance: run-time point count, the rule arrays in scratch (the plan-time compiled kernel of jit.hpp
// has them in registers)
// (converged mode: two waves per SIMD asked for, as the plan-time compiled kernels do -- 256 registers and scratch instead of 512
// with the accumulation registers as spill space.  Round 6: the 512-register build of <4, KF_LINEAR> with exp_node was not
// reproducible from run to run on identical input -- 1e-12 ... 1e-11 of scale on ~7 % of the parcels, alternative walks of the
// adaptive rule, tools/jit_aot_diff.py -- while every perturbation of its code generation was: zero- or pattern-initialised locals
// (so no uninitialised value is consumed: the pattern is NaN), exp_fin in place of exp_node, this attribute.  Not understood;
// the fuzzers and tests/test_gpu_numerical.py now assert that a second call returns the first call's bits.)
template <int N, int KIND, typename TIO, bool CONV = false>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(CONV ? 2 : 1)))
    coal_rhs_quad_kernel(const KArgs<N, 1> A, const QArgs Q, const double *__restrict__ tab, size_t n, size_t ld,
                         const TIO *__restrict__ in, TIO *__restrict__ out) {
    coal_rhs_quad_body<N, KIND, 0, TIO, CONV>(A, Q, tab, n, ld, in, out);
}

template <int N, int KIND, typename TIO, bool CONV = false>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(CONV ? 2 : 1)))
    quad_ssprk33_kernel(const KArgs<N, 1> A, const QArgs Q, const double *__restrict__ tab, size_t n, size_t ld,
                        const TIO *u_in, TIO *u_out, double dt, int n_steps) {
    quad_ssprk33_body<N, KIND, 0, TIO, CONV>(A, Q, tab, n, ld, u_in, u_out, dt, n_steps);
}

}  // namespace cloudy
