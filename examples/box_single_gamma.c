/*
 * box_single_gamma.c -- the reference driver test/examples/Analytical/box_single_gamma.jl:15-36 through the plain-C
 * ABI of libcloudy_hip.so (include/cloudy_hip.h), with no Python and no HIP headers: one 0-D box (replicated over
 * `n` lanes), Gamma(1e8, 1e-10, 1), Golovin kernel b = 5 as an order-1 tensor, norms (1e6, 1e-9), tspan (0, 120),
 * SSPRK33 with dt = 10 -- (i) stage by stage through cloudy_coal_rhs, the way OrdinaryDiffEq calls rhs!(dm, m, p, t),
 * and (ii) with the fused device integrator cloudy_ssprk33_steps.
 *
 * Build:  gcc -O2 -Iinclude examples/box_single_gamma.c -o box_single_gamma -Lcloudy.jl_amd -lcloudy_hip \
 *             -Wl,-rpath,$PWD/cloudy.jl_amd -lm
 * Exit code 0 and "OK" on success; 3 if no GPU is visible (CLOUDY_ENODEVICE).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <stdint.h>
#include "cloudy_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != CLOUDY_OK) {                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, cloudy_last_error()); \
            return rc_ == CLOUDY_ENODEVICE ? 3 : 1;                              \
        }                                                                        \
    } while (0)

int main(void) {
    const size_t n = 256, nmom = 3;
    const double eps = 2.220446049250313e-16;
    /* CoalescenceTensor(LinearKernelFunction(5.0), 1, 1e-6): c = [[eps/n0, b], [b, 0]] */
    const double c[4] = {eps / 1e6, 5.0, 5.0, 0.0};

    cloudy_plan_desc d;
    cloudy_plan_desc_init(&d);
    d.n_modes = 1;
    d.dist_type[0] = CLOUDY_DIST_GAMMA;
    d.tensor_p = 2;
    d.kernel_layout = CLOUDY_KERNEL_SINGLE;
    d.kernel_c = c;
    d.dist_thresholds[0] = INFINITY;
    d.threshold_style = CLOUDY_FIXED_THRESHOLD;
    d.norms[0] = 1e6;
    d.norms[1] = 1e-9;
    cloudy_plan *plan = NULL;
    CHECK(cloudy_plan_create(&d, &plan));
    if (cloudy_plan_nmom(plan) != (int)nmom) return 1;

    const double m0[3] = {1e8, 1e-2, 2e-12};
    double *host = (double *)malloc(sizeof(double) * nmom * n), *out = (double *)malloc(sizeof(double) * nmom * n);
    for (size_t q = 0; q < nmom; ++q)
        for (size_t i = 0; i < n; ++i) host[q * n + i] = m0[q];

    void *u, *up, *k;
    const size_t bytes = sizeof(double) * nmom * n;
    CHECK(cloudy_malloc(&u, bytes));
    CHECK(cloudy_malloc(&up, bytes));
    CHECK(cloudy_malloc(&k, bytes));

    /* (i) SSPRK33 stage by stage: three rhs!(k, u, p, t) per step, the axpys on the host for brevity */
    const double dt = 10.0;
    const int n_steps = 12;
    double *U = (double *)malloc(bytes), *UP = (double *)malloc(bytes), *K = (double *)malloc(bytes);
    memcpy(U, host, bytes);
    for (int s = 0; s < n_steps; ++s) {
        memcpy(UP, U, bytes);
        for (int stage = 0; stage < 3; ++stage) {
            CHECK(cloudy_memcpy_h2d(u, U, bytes, NULL));
            CHECK(cloudy_coal_rhs(plan, n, n, u, k, NULL));
            CHECK(cloudy_memcpy_d2h(K, k, bytes, NULL));
            CHECK(cloudy_stream_synchronize(NULL));
            for (size_t j = 0; j < nmom * n; ++j) {
                if (stage == 0)
                    U[j] = UP[j] + dt * K[j];
                else if (stage == 1)
                    U[j] = (3.0 * UP[j] + U[j] + dt * K[j]) / 4.0;
                else
                    U[j] = (UP[j] + 2.0 * U[j] + 2.0 * dt * K[j]) / 3.0;
            }
        }
    }

    /* (ii) the fused device integrator */
    CHECK(cloudy_memcpy_h2d(u, host, bytes, NULL));
    CHECK(cloudy_ssprk33_steps(plan, n, n, u, u, dt, n_steps, NULL));
    CHECK(cloudy_memcpy_d2h(out, u, bytes, NULL));
    CHECK(cloudy_stream_synchronize(NULL));

    double worst = 0.0;
    for (size_t j = 0; j < nmom * n; ++j) {
        const double r = fabs(out[j] - U[j]) / fabs(U[j]);
        if (r > worst) worst = r;
    }
    /* Golovin: M1 conserved, M0(t) = M0 exp(-b M1 t); SSPRK33 with dt b M1 = 0.5 is ~5 % off the exact decay */
    const double m0_exact = 1e8 * exp(-5.0 * 1e-2 * 120.0);
    printf("M(t=120) = [%.10e, %.10e, %.10e]  (exact M0 %.6e)\n", out[0], out[n], out[2 * n], m0_exact);
    printf("fused vs staged SSPRK33: max rel diff %.2e\n", worst);
    int ok = worst < 1e-13 && fabs(out[n] - 1e-2) < 1e-15 && fabs(out[0] / m0_exact - 1.0) < 6e-2;

    /* (iii) the batch form of the reference's silent clamps (cloudy_closure_stats): every box of the final state inverts to a
     * regular Gamma closure -- no fallback, no clamp, consistent moments; one box with M2 below M1^2 / M0 is counted */
    uint64_t counts[4];
    CHECK(cloudy_closure_stats(plan, n, n, u, counts, NULL));
    printf("closure stats of the final state: fallback %llu, k_min %llu, k_max %llu, inconsistent %llu\n",
           (unsigned long long)counts[0], (unsigned long long)counts[1], (unsigned long long)counts[2], (unsigned long long)counts[3]);
    ok = ok && counts[0] == 0 && counts[1] == 0 && counts[2] == 0 && counts[3] == 0;
    out[2 * n] = 0.5 * out[n] * out[n] / out[0];   /* parcel 0: negative variance */
    CHECK(cloudy_memcpy_h2d(u, out, bytes, NULL));
    CHECK(cloudy_closure_stats(plan, n, n, u, counts, NULL));
    ok = ok && counts[0] == 0 && counts[1] == 1 && counts[2] == 0 && counts[3] == 1;

    CHECK(cloudy_free(u));
    CHECK(cloudy_free(up));
    CHECK(cloudy_free(k));
    cloudy_plan_destroy(plan);
    free(host); free(out); free(U); free(UP); free(K);
    puts(ok ? "OK" : "MISMATCH");
    return ok ? 0 : 2;
}
