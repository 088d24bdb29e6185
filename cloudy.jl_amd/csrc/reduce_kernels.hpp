// reduce_kernels.hpp -- plane sums for the moment / mass-conservation diagnostic
// (test/examples/utils/plotting_helpers.jl:240-252).  Included by cloudy_hip.hip only.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace cloudy {

// per-block partial sums of `planes` planes; second pass adds the partials (deterministic order)
template <typename TIO>
__global__ void __launch_bounds__(kBlock)
    plane_partial_sums_kernel(size_t n, size_t ld, int planes, const TIO *__restrict__ arr,
                              double *__restrict__ partial) {
    __shared__ double red[kBlock / 64];
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (int q = 0; q < planes; ++q) {
        // four independent accumulators: four loads in flight per lane instead of one dependent add chain
        const TIO *__restrict__ a = arr + (size_t)q * ld;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
        for (; i + 3 * stride < n; i += 4 * stride) {
            s0 += (double)__builtin_nontemporal_load(a + i);
            s1 += (double)__builtin_nontemporal_load(a + i + stride);
            s2 += (double)__builtin_nontemporal_load(a + i + 2 * stride);
            s3 += (double)__builtin_nontemporal_load(a + i + 3 * stride);
        }
        for (; i < n; i += stride) s0 += (double)a[i];
        double s = (s0 + s1) + (s2 + s3);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < kBlock / 64; ++w) t += red[w];
            partial[(size_t)q * gridDim.x + blockIdx.x] = t;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kBlock)
    plane_final_sums_kernel(int nblocks, int planes, const double *__restrict__ partial, double *__restrict__ sums) {
    __shared__ double red[kBlock / 64];
    for (int q = 0; q < planes; ++q) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += kBlock) s += partial[(size_t)q * nblocks + b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < kBlock / 64; ++w) t += red[w];
            sums[q] = t;
        }
        __syncthreads();
    }
}

// cloudy_closure_stats: how many parcels of a batch had their closure inversion clamped or replaced, per mode (round 6).
// update_dist_from_moments (ParticleDistributions.jl:456-541) never fails per parcel: moments at or below eps give the fallback
// distribution (0, 1, 1), a Gamma shape outside k_range is clamped (:459-475), and nothing reports it -- with 1e7 parcels per
// call a caller could only find out by pulling the (n, theta, k) planes to the host.  Four counters per mode:
//   [0] fallback: the parcel took the (0, 1, 1) branch;   [1] shape at its lower clamp (Gamma: k = k_min; Lognormal: sigma = eps);
//   [2] shape at its upper clamp (Gamma: k = k_max);      [3] check_moment_consistency (:437-449) would throw for the mode's
//   normalised moments: a negative moment, or (three moments) a negative second central moment, summed in the reference's order.
// One pass; a lane counts its parcels in registers, a workgroup adds its lanes and stores one row of 4N counters, the host adds the rows.
struct ClosureStatsArgs {
    int N;
    int dist_type[CLOUDY_MAX_MODES], np[CLOUDY_MAX_MODES], off[CLOUDY_MAX_MODES];
    double norm[3 * CLOUDY_MAX_MODES], inv_norm[3 * CLOUDY_MAX_MODES];
    double kmin, kmax;
};
// the reference's check on one mode's (normalised) moments; NaN comparisons are false there as here
__device__ __forceinline__ bool moments_inconsistent(int np, double m0, double m1, double m2) {
    // (every product and sum rounded on its own, as Julia rounds them: the library is built with -ffp-contract=fast, and a parcel of
    // zero variance -- M2 = M1^2 / M0 to the last bit -- has a central moment of rounding size whose SIGN a fused multiply-add
    // changes: 29 of 2 500 such parcels of the bench batch, first GPU run of the test.  Neither __dmul_rn (a plain product in HIP)
    // nor `#pragma clang fp contract(off)` stops the fusion under that option -- it happens in the instruction selector -- so the
    // products pass through opaque copies before they are added.)
    if (m0 < 0.0 || m1 < 0.0 || (np == 3 && m2 < 0.0)) return true;
    if (np < 3) return false;
    // order 2: sum_i binomial(2, i) (-1)^i (m[2]/m[1])^i (m[3-i]/m[1]), i = 0, 1, 2, left to right
    const double r = m1 / m0;
    const double t0 = m2 / m0;
    double t1 = (-2.0 * r) * (m1 / m0);
    double rr = r * r;
    asm volatile("" : "+v"(t1));
    asm volatile("" : "+v"(rr));
    double t2 = rr * (m0 / m0);
    asm volatile("" : "+v"(t2));
    const double s01 = t0 + t1;
    const double cm = s01 + t2;
    return cm < 0.0;
}
template <typename TIO>
__global__ void __launch_bounds__(kBlock)
    closure_stats_kernel(const ClosureStatsArgs a, size_t n, size_t ld, const TIO *__restrict__ mom,
                         unsigned long long *__restrict__ counts) {
    unsigned int c[4 * CLOUDY_MAX_MODES];
#pragma unroll
    for (int q = 0; q < 4 * CLOUDY_MAX_MODES; ++q) c[q] = 0u;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int m = 0; m < CLOUDY_MAX_MODES; ++m) {
            if (m >= a.N) break;
            const int off = a.off[m], np = a.np[m], dt = a.dist_type[m];
            double m0 = (double)mom[(size_t)(off + 0) * ld + i], m1 = (double)mom[(size_t)(off + 1) * ld + i];
            double m2 = np == 3 ? (double)mom[(size_t)(off + 2) * ld + i] : 0.0;
            m0 = div_by_const(m0, a.norm[3 * m + 0], a.inv_norm[3 * m + 0]);   // mom ./ mom_norms, as load_parcel
            m1 = div_by_const(m1, a.norm[3 * m + 1], a.inv_norm[3 * m + 1]);
            m2 = div_by_const(m2, a.norm[3 * m + 2], a.inv_norm[3 * m + 2]);
            double nn, th, kk;
            invert_closure(dt, m0, m1, m2, a.kmin, a.kmax, nn, th, kk);
            const bool fb = dt == DIST_LOGNORMAL ? !(m0 > kEps && m1 > kEps && m2 > kEps) : !(m0 > kEps && m1 > kEps);
            c[4 * m + 0] += fb ? 1u : 0u;
            c[4 * m + 1] += (!fb && ((dt == DIST_GAMMA && kk == a.kmin) || (dt == DIST_LOGNORMAL && kk == kEps))) ? 1u : 0u;
            c[4 * m + 2] += (!fb && dt == DIST_GAMMA && kk == a.kmax) ? 1u : 0u;
            c[4 * m + 3] += moments_inconsistent(np, m0, m1, m2) ? 1u : 0u;
        }
    }
    // a wave adds its lanes, the workgroup its waves, and every workgroup STORES its own row of `counts` (the host adds the rows).
    // No counter is shared between workgroups: the first version zeroed 4N counters with hipMemsetAsync and added to them with one
    // atomic per wave -- on one box of the pool the counters of the first two modes came back short (105 of 469), those of the
    // last two whole: the zeros of the fill arrived in the middle of the kernel's atomics (round 6, last day).
    __shared__ unsigned int sh_c[kBlock / 64][4 * CLOUDY_MAX_MODES];
#pragma unroll
    for (int q = 0; q < 4 * CLOUDY_MAX_MODES; ++q) {
        if (q >= 4 * a.N) break;
        unsigned int v = c[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) sh_c[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < 4 * a.N) {
        unsigned long long t = 0ull;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) t += sh_c[w][threadIdx.x];
        counts[(size_t)blockIdx.x * (4 * a.N) + threadIdx.x] = t;
    }
}

// Upwind flux divergence of make_rainshaft_rhs (test/examples/utils/rainshaft_helpers.jl:80-86), columns of nz cells
// stored contiguously (cell index fastest): rhs[q][i] = coal[q][i] - (flux[q][i+1] - flux[q][i]) / dz, flux above the
// top cell of a column = 0.  `rhs` holds the coalescence source on entry.
template <typename TIO>
__global__ void __launch_bounds__(kBlock)
    rainshaft_divergence_kernel(size_t n, size_t ld, int planes, size_t nz, double dz, const TIO *__restrict__ flux,
                                TIO *__restrict__ rhs) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const bool top = ((i % nz) == nz - 1);
    for (int q = 0; q < planes; ++q) {
        const double f0 = (double)flux[(size_t)q * ld + i];
        const double f1 = top ? 0.0 : (double)flux[(size_t)q * ld + i + 1];
        rhs[(size_t)q * ld + i] = (TIO)((double)rhs[(size_t)q * ld + i] + (-(f1 - f0) / dz));  // rainshaft_helpers.jl:83-85
    }
}

// One SSPRK33 stage update of the staged column stepping (cloudy_rainshaft_ssprk33_steps for columns too tall for the fused
// kernel): OrdinaryDiffEq's formulas on the CLAMPED stage argument -- the reference's rhs clamps negative moments of its argument
// in place before it evaluates (rainshaft_helpers.jl:52), and f = rhs(u) was computed on the clamped values.
//   stage 0: uc = max(u, 0); up = uc; u = uc + dt f         stage 1: u = (3 up + uc + dt f) / 4
//   stage 2: u = (up + 2 uc + 2 dt f) / 3                   stage 3: u = max(u, 0)  (the FSAL evaluation on the final state)
template <typename TIO>
__global__ void __launch_bounds__(kBlock)
    rainshaft_stage_kernel(int stage, size_t n, size_t ld, int planes, TIO *__restrict__ up, TIO *__restrict__ u,
                           const TIO *__restrict__ f, double dt) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    for (int q = 0; q < planes; ++q) {
        const size_t e = (size_t)q * ld + i;
        const double uq = (double)u[e], uc = uq < 0.0 ? 0.0 : uq;
        if (stage == 3) {
            u[e] = (TIO)uc;
            continue;
        }
        const double fq = (double)f[e];
        if (stage == 0) {
            up[e] = (TIO)uc;
            u[e] = (TIO)(uc + dt * fq);
        } else if (stage == 1) {
            u[e] = (TIO)((3.0 * (double)up[e] + uc + dt * fq) * 0.25);
        } else {
            u[e] = (TIO)(((double)up[e] + 2.0 * uc + 2.0 * dt * fq) / 3.0);
        }
    }
}

}  // namespace cloudy
