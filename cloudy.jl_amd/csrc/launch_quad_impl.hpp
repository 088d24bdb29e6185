// launch_quad_impl.hpp -- ahead-of-time kernels of the NumericalCoalStyle plans for one N (quad_n<N>.hip): every kernel
// function family x both plane types, run-time rule order.  The plan-time compiled kernels (jit.hpp) are the fast path.
#pragma once
#include "launch_impl.hpp"
#include "quad_kernels.hpp"

namespace cloudy {
static_assert(kConvUnrolledModes == CLOUDY_AOT_MAX_MODES, "quad_conv.hpp: the rolled form serves the plans beyond the ahead-of-time families");

template <int N, int KIND, typename TIO, bool CONV>
hipError_t launch_quad_io2(const HostPlan &h, const LaunchReq &r, const KArgs<N, 1> &A) {
    if (r.op == OP_SSPRK33) {  // cloudy_ssprk33_steps of a NumericalCoalStyle plan (quad_ssprk33_body)
        hipLaunchKernelGGL((quad_ssprk33_kernel<N, KIND, TIO, CONV>), dim3(grid_for(r.n, true)), dim3(kBlock), 0, r.stream, A,
                           h.q, h.qtab_dev, r.n, r.ld, static_cast<const TIO *>(r.in), static_cast<TIO *>(r.out), r.dt,
                           r.n_steps);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((coal_rhs_quad_kernel<N, KIND, TIO, CONV>), dim3(grid_for(r.n, true)), dim3(kBlock), 0, r.stream, A, h.q,
                       h.qtab_dev, r.n, r.ld, static_cast<const TIO *>(r.in), static_cast<TIO *>(r.out));
    return hipGetLastError();
}

template <int N, int KIND, typename TIO>
hipError_t launch_quad_io(const HostPlan &h, const LaunchReq &r, const KArgs<N, 1> &A) {
    if (h.q.mode == QUAD_CONVERGED) return launch_quad_io2<N, KIND, TIO, true>(h, r, A);
    return launch_quad_io2<N, KIND, TIO, false>(h, r, A);
}

template <int N>
hipError_t launch_quad(const HostPlan &h, const LaunchReq &r) {
    if ((r.op != OP_COAL && r.op != OP_SSPRK33) || h.P != 1 || !h.qtab_dev) return hipErrorInvalidValue;
    KArgs<N, 1> A;
    fill_args<N, 1>(h, r, A);
    const bool f32 = h.dtype != CLOUDY_F64 && r.input_kind == IN_MOMENTS;  // (n, theta, k) planes are always fp64
    switch (h.q.kind) {
#define CLOUDY_QUAD_CASE(K) \
    case K: return f32 ? launch_quad_io<N, K, float>(h, r, A) : launch_quad_io<N, K, double>(h, r, A);
        CLOUDY_QUAD_CASE(KF_CONSTANT)
        CLOUDY_QUAD_CASE(KF_LINEAR)
        CLOUDY_QUAD_CASE(KF_HYDRODYNAMIC)
        CLOUDY_QUAD_CASE(KF_LONG)
#undef CLOUDY_QUAD_CASE
    default: return hipErrorInvalidValue;
    }
}

}  // namespace cloudy
