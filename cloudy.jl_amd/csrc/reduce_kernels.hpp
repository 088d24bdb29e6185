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
