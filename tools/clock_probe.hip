// clock_probe: what shader clock does the part sustain under a pure fp64-VALU load?  (round 4)
// bench.py's valu_issue_frac prices issue slots at 2.4 GHz.  A kernel that keeps every SIMD issuing fp64 FMAs with all 64
// lanes active may be held below that clock by the power limit; then "idle issue slots" at the nominal clock are not idle
// at all.  Each workgroup reads s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around ITERS rounds of
// eight independent FMA chains; reported: shader MHz = d(clock64) / d(wall_clock64) x 100, for all lanes active, for
// 3 of 4 lanes masked off, and for one wave per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/clock_probe tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) burn(double *out, long long *clk, int iters, int lane_mod) {
    const int t = threadIdx.x;
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0 + 1e-9 * (t + k);
    const double b = 1.0000001, c = 1e-12;
    const long long c0 = clock64(), w0 = wall_clock64();
    if (t % lane_mod == 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = fma(a[k], b, c);
            }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0.0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[(size_t)blockIdx.x * blockDim.x + t] = s;
    if (t == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    int dev = 0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, dev);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, nominal %d MHz\n", p.name, cus, p.clockRate / 1000);
    struct Cfg { const char *name; int wg_per_cu; int lane_mod; } cfgs[] = {
        {"8 waves/SIMD-pair (2 WG/CU... 256 thr), all lanes", 8, 1},
        {"same, 1 of 4 lanes active", 8, 4},
        {"1 WG/CU (1 wave/SIMD), all lanes", 1, 1},
        {"8 WG/CU again, all lanes (after warm-up)", 8, 1},
    };
    for (auto &c : cfgs) {
        const int blocks = cus * c.wg_per_cu;
        double *out;
        long long *clk;
        hipMalloc(&out, sizeof(double) * blocks * 256);
        hipMalloc(&clk, sizeof(long long) * 2 * blocks);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        burn<<<blocks, 256>>>(out, clk, iters / 10, c.lane_mod);  // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        burn<<<blocks, 256>>>(out, clk, iters, c.lane_mod);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(2 * blocks);
        hipMemcpy(h.data(), clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        for (int b = 0; b < blocks; ++b) mhz.push_back(100.0 * (double)h[2 * b] / (double)h[2 * b + 1]);
        std::sort(mhz.begin(), mhz.end());
        const double fmas = (double)blocks * 256 / c.lane_mod * (double)iters * 128;
        printf("%-52s %8.3f ms  shader clock median %7.1f MHz (min %7.1f, max %7.1f)  %6.2f TFLOP/s fp64\n", c.name, ms,
               mhz[mhz.size() / 2], mhz.front(), mhz.back(), 2.0 * fmas / (ms * 1e-3) / 1e12);
        hipFree(out);
        hipFree(clk);
    }
    return 0;
}
