// rcp_probe: how good is v_rcp_f64 on gfx950, and what do one / two Newton steps (and the cubic single pass) leave?  (round 6)
// recip_fast (device_math.hpp) takes two Newton steps after the hardware estimate; the converged-mode walk has 22 reciprocals per trip.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/rcp_probe tools/rcp_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void probe(const double *x, double *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double r0 = __builtin_amdgcn_rcp(v);
    const double e0 = fma(-v, r0, 1.0);
    const double r1 = fma(e0, r0, r0);
    const double r2 = fma(fma(-v, r1, 1.0), r1, r1);
    const double rc = fma(fma(e0, e0, e0), r0, r0);
    out[4 * i] = r0, out[4 * i + 1] = r1, out[4 * i + 2] = r2, out[4 * i + 3] = rc;
}

int main() {
    const int n = 1 << 22;
    std::vector<double> h(n), o(4 * (size_t)n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
        h[i] = ldexp(1.0 + u, (int)(s % 41) - 20);
    }
    double *dx, *dout;
    (void)hipMalloc(&dx, n * sizeof(double));
    (void)hipMalloc(&dout, 4 * (size_t)n * sizeof(double));
    (void)hipMemcpy(dx, h.data(), n * sizeof(double), hipMemcpyHostToDevice);
    probe<<<(n + 255) / 256, 256>>>(dx, dout, n);
    (void)hipMemcpy(o.data(), dout, 4 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    const char *nm[4] = {"v_rcp_f64", "one Newton step", "two Newton steps (recip_fast)", "cubic single pass r (1 + e + e^2)"};
    for (int k = 0; k < 4; ++k) {
        long double w = 0;
        for (int i = 0; i < n; ++i) {
            const long double e = fabsl((long double)o[4 * (size_t)i + k] * (long double)h[i] - 1.0L);
            if (e > w) w = e;
        }
        printf("%-36s max |r x - 1| = %.3Le\n", nm[k], w);
    }
    return 0;
}
