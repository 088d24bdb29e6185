// launch_impl.hpp -- fills KArgs<N,P> from the host plan and launches the kernel for one N (all P, all modes).
// Included by inst_n{1,2,3,4}.hip so that the instantiations compile in parallel.
#pragma once
#include "host_plan.hpp"
#include "kernels.hpp"

namespace cloudy {

inline unsigned grid_for(size_t n, bool /*heavy*/) {
    size_t blocks = (n + kBlock - 1) / kBlock;  // one parcel per lane (see kernels.hpp)
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

template <int N, int P>
void fill_args(const HostPlan &h, const LaunchReq &r, KArgs<N, P> &A) {
    for (int i = 0; i < N; ++i) {
        A.dist_type[i] = h.dist_type[i];
        A.np[i] = h.np[i];
        A.off[i] = h.off[i];
        A.finite[i] = h.finite[i];
        A.node_off[i] = h.node_off[i];
        A.n_bins[i] = h.n_bins[i];
        A.n_2d[i] = h.n_2d[i];
        A.thr[i] = h.thr[i];
        for (int m = 0; m < 3; ++m) {
            A.norm[3 * i + m] = h.mom_norm[i][m];
            A.inv_norm[3 * i + m] = 1.0 / h.mom_norm[i][m];
            A.out_scale[3 * i + m] = r.physical_out ? h.mom_norm[i][m] : 1.0;
        }
    }
    A.n_mom_max = h.n_mom_max;
    A.input_kind = r.input_kind;
    A.rainshaft = r.rainshaft;
    A.nbpl = h.nbpl;
    A.kmin = h.kmin;
    A.kmax = h.kmax;
    for (int j = 0; j < N; ++j)
        for (int k = 0; k < N; ++k)
            for (int a = 0; a < P; ++a)
                for (int b = 0; b < P; ++b) A.c[j][k][a][b] = h.c[j][k][a][b];
    static_assert(kInvTerms == 16, "HostPlan::inv_tab holds 16 coefficients per mode");
    A.inv_map[0] = h.inv_map[0];
    A.inv_map[1] = h.inv_map[1];
    A.inv_klo = h.inv_klo;
    for (int i = 0; i < N; ++i)
        for (int d = 0; d < kInvTerms; ++d) A.inv_tab[i][d] = h.inv_tab[i][d];
}

// workgroup size of the threshold kernel for a plan (see coal_rhs_sorted_kernel in kernels.hpp)
inline int sorted_block_size(const HostPlan &h) {
    if (h.mode != MODE_FIXED || h.N > 2) return kBlock;
    int passes = 0;
    for (int i = 0; i < h.N - 1; ++i) passes += h.finite[i] ? 1 : 0;
    return passes == 1 ? 512 : kBlock;
}

inline void fill_sedi(const HostPlan &h, SediArgs &S) {
    S.n_vel = h.n_vel;
    S.pad = 0;
    for (int v = 0; v < 4; ++v) {
        S.vel[v][0] = v < h.n_vel ? h.vel_n[v][0] : 0.0;
        S.vel[v][1] = v < h.n_vel ? h.vel_n[v][1] : 0.0;
    }
}

// kernels that exist for both plane types (double / float storage)
template <int N, int P, typename TIO>
hipError_t launch_io(const HostPlan &h, const LaunchReq &r, const KArgs<N, P> &A) {
    const bool heavy = h.mode != MODE_ALLINF;
    const TIO *in = static_cast<const TIO *>(r.in);
    TIO *out = static_cast<TIO *>(r.out);
    switch (r.op) {
    case OP_COAL: {
        const unsigned g = grid_for(r.n, heavy);
        const uintptr_t amask = 2 * sizeof(TIO) - 1;  // two parcels per lane need 2-element aligned planes
        const bool aligned2 = ((reinterpret_cast<uintptr_t>(r.in) | reinterpret_cast<uintptr_t>(r.out)) & amask) == 0 &&
                              (r.ld % 2 == 0);
        if (h.mode == MODE_ALLINF && r.input_kind == IN_MOMENTS && !r.rainshaft && aligned2 && !h.force_ppl1)
            hipLaunchKernelGGL((coal_rhs_allinf2_kernel<N, P, TIO>), dim3(grid_for((r.n + 1) / 2, false)), dim3(kBlock),
                               0, r.stream, A, r.n, r.ld, in, out);
        else if (h.mode == MODE_ALLINF)
            hipLaunchKernelGGL((coal_rhs_kernel<N, P, MODE_ALLINF, TIO>), dim3(g), dim3(kBlock), 0, r.stream, A,
                               h.nodes_dev, r.n, r.ld, in, out);
        else {
            // FixedThreshold plans with N <= 2 and one thresholded mode rank 512 parcels per workgroup (kernels.hpp)
            const bool fast = h.dtype == CLOUDY_F32_FAST && sizeof(TIO) == 4 && r.input_kind == IN_MOMENTS;
            if (N <= 2 && sorted_block_size(h) == 512) {
                if constexpr (N <= 2) {
                    const unsigned g5 = (unsigned)((r.n + 511) / 512);
                    if (fast)
                        hipLaunchKernelGGL((coal_rhs_sorted_kernel<N, P, MODE_FIXED, TIO, true, 512>), dim3(g5), dim3(512), 0,
                                           r.stream, A, h.nodes_dev, r.n, r.ld, in, out);
                    else
                        hipLaunchKernelGGL((coal_rhs_sorted_kernel<N, P, MODE_FIXED, TIO, false, 512>), dim3(g5), dim3(512), 0,
                                           r.stream, A, h.nodes_dev, r.n, r.ld, in, out);
                }
            } else if (fast) {
                if (h.mode == MODE_FIXED)
                    hipLaunchKernelGGL((coal_rhs_sorted_kernel<N, P, MODE_FIXED, TIO, true>), dim3(g), dim3(kBlock), 0,
                                       r.stream, A, h.nodes_dev, r.n, r.ld, in, out);
                else
                    hipLaunchKernelGGL((coal_rhs_sorted_kernel<N, P, MODE_MOVING, TIO, true>), dim3(g), dim3(kBlock), 0,
                                       r.stream, A, h.nodes_dev, r.n, r.ld, in, out);
            } else if (h.mode == MODE_FIXED)
                hipLaunchKernelGGL((coal_rhs_sorted_kernel<N, P, MODE_FIXED, TIO>), dim3(g), dim3(kBlock), 0, r.stream, A,
                                   h.nodes_dev, r.n, r.ld, in, out);
            else
                hipLaunchKernelGGL((coal_rhs_sorted_kernel<N, P, MODE_MOVING, TIO>), dim3(g), dim3(kBlock), 0, r.stream, A,
                                   h.nodes_dev, r.n, r.ld, in, out);
        }
        break;
    }
    case OP_SEDI: {
        SediArgs S;
        fill_sedi(h, S);
        hipLaunchKernelGGL((sedi_flux_kernel<N, P, TIO>), dim3(grid_for(r.n, false)), dim3(kBlock), 0, r.stream, A, S,
                           r.n, r.ld, in, out);
        break;
    }
    case OP_NQ:  // r.s_scalar = size cutoff in physical mass units
        hipLaunchKernelGGL((standard_nq_kernel<N, P, TIO>), dim3(grid_for(r.n, false)), dim3(kBlock), 0, r.stream, A,
                           r.s_scalar / h.norms[1], h.norms[0], h.norms[1], r.n, r.ld, in, out);
        break;
    case OP_COND:
        hipLaunchKernelGGL((cond_evap_kernel<N, P, TIO>), dim3(grid_for(r.n, false)), dim3(kBlock), 0, r.stream, A,
                           r.coef, r.s_scalar, r.s_dev, r.n, r.ld, in, out);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int N, int P>
hipError_t launch_np(const HostPlan &h, const LaunchReq &r) {
    KArgs<N, P> A;
    fill_args<N, P>(h, r, A);
    const bool heavy = h.mode != MODE_ALLINF;
    switch (r.op) {
    case OP_PREPARE: {  // constant block in device memory for the fused integrators (moments in, physical units out)
        struct Block {  // SediArgs directly behind KArgs: rainshaft_ssprk33_kernel reads it at (Ag + 1)
            KArgs<N, P> A;
            SediArgs S;
        } blk;
        static_assert(sizeof(KArgs<N, P>) % 8 == 0 && offsetof(Block, S) == sizeof(KArgs<N, P>), "block layout");
        blk.A = A;
        fill_sedi(h, blk.S);
        void *dev = nullptr;
        hipError_t e = hipMalloc(&dev, sizeof(blk));
        if (e != hipSuccess) return e;
        e = hipMemcpy(dev, &blk, sizeof(blk), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(dev);
            return e;
        }
        *static_cast<void **>(r.out) = dev;
        return hipSuccess;
    }
    case OP_SSPRK33:
    case OP_TSIT5:
    case OP_RAINSHAFT_SSPRK33:
        return launch_int<N, P>(h, r);  // int_n<N>_p<P>.hip
    case OP_COAL:
    case OP_SEDI:
    case OP_COND:
    case OP_NQ:
        // get_coal_ints on (n, theta, k) planes is an fp64 interface for every plan
        if (h.dtype != CLOUDY_F64 && r.input_kind == IN_MOMENTS) return launch_io<N, P, float>(h, r, A);
        return launch_io<N, P, double>(h, r, A);
    case OP_UPDATE_DIST:
        hipLaunchKernelGGL((update_dist_kernel<N, P>), dim3(grid_for(r.n, false)), dim3(kBlock), 0, r.stream, A, r.n,
                           r.ld, static_cast<const double *>(r.in), static_cast<double *>(r.out));
        break;
    case OP_FINITE_2D: {
        const unsigned g = grid_for(r.n, heavy);
        const double *in = static_cast<const double *>(r.in);
        double *out = static_cast<double *>(r.out), *out2 = static_cast<double *>(r.out2);
        if (h.mode == MODE_ALLINF)
            hipLaunchKernelGGL((finite_2d_kernel<N, P, MODE_ALLINF>), dim3(g), dim3(kBlock), 0, r.stream, A,
                               h.nodes_dev, r.n, r.ld, in, out, out2);
        else if (h.mode == MODE_FIXED)
            hipLaunchKernelGGL((finite_2d_kernel<N, P, MODE_FIXED>), dim3(g), dim3(kBlock), 0, r.stream, A,
                               h.nodes_dev, r.n, r.ld, in, out, out2);
        else
            hipLaunchKernelGGL((finite_2d_kernel<N, P, MODE_MOVING>), dim3(g), dim3(kBlock), 0, r.stream, A,
                               h.nodes_dev, r.n, r.ld, in, out, out2);
        break;
    }
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace cloudy
