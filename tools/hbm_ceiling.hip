// hbm_ceiling.hip -- what a do-nothing kernel with the coalescence RHS's memory shape sustains on this GPU:
// reads `planes` planes of n doubles and writes `planes` planes (8 B/lane and 16 B/lane variants), launched
// back-to-back like bench.py launches the RHS.  Prints per-launch GB/s for the first launches and the sustained
// median.  Build: hipcc -O3 --offload-arch=gfx950 tools/hbm_ceiling.hip -o tools/hbm_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int PLANES>
__global__ void __launch_bounds__(256) copy8(size_t n, size_t ld, const double* __restrict__ in, double* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        double v[PLANES];
#pragma unroll
        for (int q = 0; q < PLANES; ++q) v[q] = in[q * ld + i];
#pragma unroll
        for (int q = 0; q < PLANES; ++q) out[q * ld + i] = v[q] * 1.0000001;
    }
}
template <int PLANES, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) copy8nt(size_t n, size_t ld, const double* __restrict__ in, double* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        double v[PLANES];
#pragma unroll
        for (int q = 0; q < PLANES; ++q) v[q] = NTL ? __builtin_nontemporal_load(in + q * ld + i) : in[q * ld + i];
#pragma unroll
        for (int q = 0; q < PLANES; ++q) {
            if (NTS) __builtin_nontemporal_store(v[q] * 1.0000001, out + q * ld + i);
            else out[q * ld + i] = v[q] * 1.0000001;
        }
    }
}
template <int PLANES>
__global__ void __launch_bounds__(256) copy16(size_t n, size_t ld, const double* __restrict__ in, double* __restrict__ out) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < n) {
        double2 v[PLANES];
#pragma unroll
        for (int q = 0; q < PLANES; ++q) v[q] = *reinterpret_cast<const double2*>(in + q * ld + i);
#pragma unroll
        for (int q = 0; q < PLANES; ++q) *reinterpret_cast<double2*>(out + q * ld + i) = make_double2(v[q].x * 1.0000001, v[q].y * 1.0000001);
    }
}
typedef double dv2 __attribute__((ext_vector_type(2)));
template <int PLANES>
__global__ void __launch_bounds__(256) copy16nt(size_t n, size_t ld, const double* __restrict__ in, double* __restrict__ out) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < n) {
        dv2 v[PLANES];
#pragma unroll
        for (int q = 0; q < PLANES; ++q) v[q] = __builtin_nontemporal_load(reinterpret_cast<const dv2*>(in + q * ld + i));
#pragma unroll
        for (int q = 0; q < PLANES; ++q) __builtin_nontemporal_store(v[q] * 1.0000001, reinterpret_cast<dv2*>(out + q * ld + i));
    }
}

int main(int argc, char** argv) {
    size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000;
    int iters = argc > 2 ? atoi(argv[2]) : 60;
    const int PL = 6;
    double *in, *out;
    CK(hipMalloc(&in, PL * n * 8));
    CK(hipMalloc(&out, PL * n * 8));
    CK(hipMemset(in, 0x11, PL * n * 8));
    std::vector<hipEvent_t> ev(iters + 1);
    for (auto& evt : ev) CK(hipEventCreate(&evt));
    for (int variant = 0; variant < 6; ++variant) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(ev[0], 0));
        for (int it = 0; it < iters; ++it) {
            if (variant == 0) hipLaunchKernelGGL(copy8<PL>, dim3((n + 255) / 256), dim3(256), 0, 0, n, n, in, out);
            else if (variant == 1) hipLaunchKernelGGL(copy16<PL>, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, n, n, in, out);
            else if (variant == 2) hipLaunchKernelGGL((copy8nt<PL, true, false>), dim3((n + 255) / 256), dim3(256), 0, 0, n, n, in, out);
            else if (variant == 3) hipLaunchKernelGGL((copy8nt<PL, false, true>), dim3((n + 255) / 256), dim3(256), 0, 0, n, n, in, out);
            else if (variant == 4) hipLaunchKernelGGL((copy8nt<PL, true, true>), dim3((n + 255) / 256), dim3(256), 0, 0, n, n, in, out);
            else hipLaunchKernelGGL(copy16nt<PL>, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, n, n, in, out);
            CK(hipEventRecord(ev[it + 1], 0));
        }
        CK(hipDeviceSynchronize());
        std::vector<float> us(iters);
        for (int it = 0; it < iters; ++it) { float ms; CK(hipEventElapsedTime(&ms, ev[it], ev[it + 1])); us[it] = ms * 1e3f; }
        double bytes = 2.0 * PL * n * 8;
        printf("%s: first 8 launches us:", (const char*[]){"copy 8B/lane      ", "copy 16B/lane     ", "8B nt-load        ", "8B nt-store       ", "8B nt-load+store  ", "16B nt-load+store "}[variant]);
        for (int it = 0; it < 8 && it < iters; ++it) printf(" %.0f", us[it]);
        std::vector<float> tail(us.begin() + iters / 2, us.end());
        std::sort(tail.begin(), tail.end());
        float med = tail[tail.size() / 2];
        printf("  | sustained median %.1f us = %.0f GB/s, best %.1f us = %.0f GB/s\n", med, bytes / med * 1e-3,
               *std::min_element(us.begin(), us.end()), bytes / *std::min_element(us.begin(), us.end()) * 1e-3);
    }
    return 0;
}
