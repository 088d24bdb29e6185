// allinf_f32.hpp -- CLOUDY_F32_FAST on a plan WITHOUT thresholds: float planes AND single-precision arithmetic (round 4).
//
// A CLOUDY_F32 plan halves the HBM traffic of the headline kernel (48 B per parcel of the 2-mode, order-2 workload) but
// keeps its ~260 fp64 instructions per parcel, and at half the bytes those instructions bound it (0.107 ms per 1e7
// parcels = 0.56 of the 8 TB/s roofline on 480 MB, where the fp64-plane kernel sits at 0.75).  fp32 FMAs issue at the
// same rate as fp64 ones on this part -- unless they are PACKED: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do two
// single-precision operations per lane.  So a lane takes FOUR parcels, as two packed pairs (one 16-B nontemporal access
// per plane, as the fp64 kernel), and every operation of the fused body -- mom ./ norms, closure inversion, moments by
// recurrence, the Q/R/S algebra of pair_terms -- is written on 2-vectors, which the compiler lowers to the packed
// instructions.  Gamma / Exponential / Monodisperse closures (a Lognormal mode keeps the fp64-arithmetic kernel).
// The eps rule (F[p,q] = 0 where M_p M_q < eps, Coalescence.jl:213) fires only when a moment is below 2^-26; such parcels
// (and only they) go through the fp64 routines of kernels.hpp.  Reference: rhs_coal!, box_model_helpers.jl:29-53.
// Error against the fp64 oracle: tests/test_gpu_parity.py reports it (single precision in the ill-conditioned shape
// inversion k = mean / (M2/M1 - mean): ~1e-6 of scale, more where the variance is a small difference).
#pragma once
#include "kernels.hpp"

namespace cloudy {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2f v2(float a) { return v2f{a, a}; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// 1/x: hardware estimate (1 ulp) + one Newton step, per component (v_rcp_f32 has no packed form)
__device__ __forceinline__ v2f rcp2(v2f x) {
    v2f r{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
    return fma2(fma2(-x, r, v2(1.0f)), r, r);
}
__device__ __forceinline__ v2f sel2(bool cx, bool cy, v2f a, v2f b) { return v2f{cx ? a.x : b.x, cy ? a.y : b.y}; }

// update_dist_from_moments (ParticleDistributions.jl:456-476, :512-523, :530-541) on a packed pair, single precision
__device__ __forceinline__ void invert_closure_f32(int dist_type, v2f m0, v2f m1, v2f m2, float kmin, float kmax, v2f &n,
                                                    v2f &th, v2f &k) {
    const float eps = 2.220446049250313e-16f;  // the reference's eps(Float64) guard, in normalised units
    const bool okx = m0.x > eps && m1.x > eps, oky = m0.y > eps && m1.y > eps;
    const v2f mean = m1 * rcp2(m0);
    v2f kk = v2(1.0f), tt = mean;
    if (dist_type == DIST_GAMMA) {
        const v2f d = fma2(m2, rcp2(m1), -mean);
        // (IEEE quotient: the divisor may be zero -> +-Inf -> the clamps; 0/0 -> NaN, which Julia's min / max propagate)
        const v2f q{mean.x / d.x, mean.y / d.y};
        const v2f cl{fmaxf(fminf(q.x, kmax), kmin), fmaxf(fminf(q.y, kmax), kmin)};
        kk = sel2(q.x != q.x, q.y != q.y, q, cl);
        tt = mean * rcp2(kk);
    }
    n = sel2(okx, oky, m0, v2(0.0f));
    th = sel2(okx, oky, tt, v2(1.0f));
    k = sel2(okx, oky, kk, v2(1.0f));
}

template <int N, int P, bool SPEC>
__device__ __forceinline__ void coal_ints_pair_f32(const KArgs<N, P> &A, const v2f (&nn)[N], const v2f (&th)[N],
                                                   const v2f (&kk)[N], v2f (&acc)[N][3]) {
    constexpr int M = P + 2;
    v2f Mm[N][M];
    bool rare = false;  // a moment below 2^-26: the eps rule may fire (finite_2d_and_promoted)
#pragma unroll
    for (int i = 0; i < N; ++i) {
        Mm[i][0] = nn[i];
        const bool mono = A.dist_type[i] == DIST_MONO;
#pragma unroll
        for (int q = 1; q < M; ++q) Mm[i][q] = Mm[i][q - 1] * (mono ? th[i] : th[i] * (kk[i] + v2(float(q - 1))));
        if (A.n_mom_max < M) {
#pragma unroll
            for (int q = 0; q < M; ++q)
                if (q >= A.n_mom_max) Mm[i][q] = v2(0.0f);
        }
#pragma unroll
        for (int q = 0; q < M; ++q)
            if (q < A.n_mom_max)
                rare = rare || (nn[i].x > 0.0f && Mm[i][q].x < (float)kSqrtEps) || (nn[i].y > 0.0f && Mm[i][q].y < (float)kSqrtEps);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k][0] = acc[k][1] = acc[k][2] = v2(0.0f);
    // ---- pair terms (kernels.hpp, pair_terms): Q - R for j < k, -R for j > k, S_1 - R for j == k, common products cancelled
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j < k) {
                v2f s1 = v2(0.0f), s2a = v2(0.0f), s2b = v2(0.0f);
#pragma unroll
                for (int b = 0; b < P; ++b) {
                    v2f v1 = v2(0.0f), vv2 = v2(0.0f);
#pragma unroll
                    for (int a = 0; a < P; ++a)
                        if (!SPEC || A.c[j][k][a][b] != 0.0) {
                            v1 = fma2(v2((float)A.c[j][k][a][b]), Mm[j][a + 1], v1);
                            vv2 = fma2(v2((float)A.c[j][k][a][b]), Mm[j][a + 2], vv2);
                        }
                    s1 = fma2(v1, Mm[k][b], s1);
                    s2a = fma2(v1, Mm[k][b + 1], s2a);
                    s2b = fma2(vv2, Mm[k][b], s2b);
                }
                acc[k][1] += s1;
                acc[k][2] += fma2(v2(2.0f), s2a, s2b);
            } else if (j > k) {
                v2f r0 = v2(0.0f), r1 = v2(0.0f), r2 = v2(0.0f);
#pragma unroll
                for (int b = 0; b < P; ++b) {
                    v2f v = v2(0.0f);
#pragma unroll
                    for (int a = 0; a < P; ++a)
                        if (!SPEC || A.c[j][k][a][b] != 0.0) v = fma2(v2((float)A.c[j][k][a][b]), Mm[j][a], v);
                    r0 = fma2(v, Mm[k][b], r0);
                    r1 = fma2(v, Mm[k][b + 1], r1);
                    r2 = fma2(v, Mm[k][b + 2], r2);
                }
                acc[k][0] -= r0;
                acc[k][1] -= r1;
                acc[k][2] -= r2;
            } else {
                v2f r0 = v2(0.0f), s2 = v2(0.0f);
#pragma unroll
                for (int b = 0; b < P; ++b) {
                    v2f v = v2(0.0f), v1 = v2(0.0f);
#pragma unroll
                    for (int a = 0; a < P; ++a)
                        if (!SPEC || A.c[k][k][a][b] != 0.0) {
                            v = fma2(v2((float)A.c[k][k][a][b]), Mm[k][a], v);
                            v1 = fma2(v2((float)A.c[k][k][a][b]), Mm[k][a + 1], v1);
                        }
                    r0 = fma2(v, Mm[k][b], r0);
                    s2 = fma2(v1, Mm[k][b + 1], s2);
                }
                acc[k][0] -= v2(0.5f) * r0;
                acc[k][2] += s2;
            }
        }
    }
    // ---- the eps rule: the promoted part D = M_p M_q - F of the self collisions, through the fp64 routines (rare)
    if (rare) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const float nk = e ? nn[k].y : nn[k].x;
                double Mk[M];
                double mn = 1e300;
#pragma unroll
                for (int q = 0; q < M; ++q) {
                    Mk[q] = (double)(e ? Mm[k][q].y : Mm[k][q].x);
                    if (q < A.n_mom_max) mn = fmin(mn, Mk[q]);
                }
                if (!(nk > 0.0f) || !(mn < kSqrtEps)) continue;
                double F[M * (M + 1) / 2], D[M * (M + 1) / 2], msh[M * (M + 1) / 2], T0, T1, T2;
#pragma unroll
                for (int t = 0; t < M * (M + 1) / 2; ++t) msh[t] = 0.0;
                finite_2d_and_promoted<P>(Mk, false, msh, F, D);
                contract_promoted<P>(A.c[k][k], D, T0, T1, T2);
                const float t0 = (float)T0, t1 = (float)T1, t2 = (float)T2;
                if (e) {
                    acc[k][0].y -= t0;
                    acc[k][1].y -= t1;
                    acc[k][2].y -= t2;
                    if (k + 1 < N) {
                        acc[k + 1][0].y += t0;
                        acc[k + 1][1].y += t1;
                        acc[k + 1][2].y += t2;
                    }
                } else {
                    acc[k][0].x -= t0;
                    acc[k][1].x -= t1;
                    acc[k][2].x -= t2;
                    if (k + 1 < N) {
                        acc[k + 1][0].x += t0;
                        acc[k + 1][1].x += t1;
                        acc[k + 1][2].x += t2;
                    }
                }
            }
        }
    }
}

// cloudy_coal_rhs of an all-Inf CLOUDY_F32_FAST plan: four parcels per lane.  aligned16 (the same for every lane: 16-B aligned
// planes and ld % 4 == 0): one 16-B nontemporal access per plane and lane; otherwise four scalar accesses -- the SAME
// arithmetic either way (round 5, ADVICE r4: until then an unaligned batch or an odd stride took the fp64-arithmetic kernel, so
// the precision of the plan depended on the layout of the batch -- e.g. on the offset of a shard)
template <int N, int P, bool SPEC>
__device__ __forceinline__ void coal_rhs_allinf4_f32_body(const KArgs<N, P> &A, size_t n, size_t ld, const float *__restrict__ in,
                                                          float *__restrict__ out, int aligned16 = 1) {
    const size_t i = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (i >= n) return;
    const bool whole = i + 3 < n, full = whole && aligned16 != 0;
    v2f nn[2][N], th[2][N], kk[2][N], acc[2][N][3];
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const int off = A.off[m];
        const bool three = A.np[m] == 3;
        v4f q[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            q[o] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
            if (o == 2 && !three) continue;
            const float *p = in + (size_t)(off + o) * ld + i;
            if (full) {
                q[o] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
            } else {
                q[o].x = p[0];
                if (i + 1 < n) q[o].y = p[1];
                if (i + 2 < n) q[o].z = p[2];
                if (whole) q[o].w = p[3];
            }
        }
        const float r0 = (float)A.inv_norm[3 * m + 0], r1 = (float)A.inv_norm[3 * m + 1], r2 = (float)A.inv_norm[3 * m + 2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const v2f m0 = (e ? v2f{q[0].z, q[0].w} : v2f{q[0].x, q[0].y}) * v2(r0);
            const v2f m1 = (e ? v2f{q[1].z, q[1].w} : v2f{q[1].x, q[1].y}) * v2(r1);
            const v2f m2 = (e ? v2f{q[2].z, q[2].w} : v2f{q[2].x, q[2].y}) * v2(r2);
            invert_closure_f32(A.dist_type[m], m0, m1, m2, (float)A.kmin, (float)A.kmax, nn[e][m], th[e][m], kk[e][m]);
        }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) coal_ints_pair_f32<N, P, SPEC>(A, nn[e], th[e], kk[e], acc[e]);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int off = A.off[k];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            if (m == 2 && A.np[k] != 3) continue;
            const v2f s = v2((float)A.out_scale[3 * k + m]);
            const v2f a = acc[0][k][m] * s, b = acc[1][k][m] * s;
            float *dst = out + (size_t)(off + m) * ld + i;
            if (full) {
                __builtin_nontemporal_store(v4f{a.x, a.y, b.x, b.y}, reinterpret_cast<v4f *>(dst));
            } else {
                dst[0] = a.x;
                if (i + 1 < n) dst[1] = a.y;
                if (i + 2 < n) dst[2] = b.x;
                if (whole) dst[3] = b.y;
            }
        }
    }
}

}  // namespace cloudy
