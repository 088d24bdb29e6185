// launch_int_impl.hpp -- the fused on-device integrators of one (N, P) family.  Included only by int_n<N>_p<P>.hip,
// which the Makefile compiles with `-mllvm -disable-machine-licm`: the step/stage loops of these kernels wrap the whole
// right-hand side, and machine LICM hoists every literal of the libm polynomials (exp, log, lgamma) and of the series
// out of them into registers for the life of the kernel -- 256 VGPRs, occupancy 1 and spills, against 108-170 VGPRs
// without (measured on MI355X: cloudy_ssprk33_steps on the cfg3b batch 1.12 -> 0.53 ms per RHS evaluation per 1e6
// parcels, the rainshaft column integrator 1.53 -> 0.78).  The single-pass kernels of inst_n<N>_p<P>.hip keep LICM:
// their Simpson node loops gain from the hoisting (cfg3b, at the time: 4.85 ms with, 4.96 ms without).
#pragma once
#include "launch_impl.hpp"

namespace cloudy {

template <int N, int P, typename TIO>
hipError_t launch_int_io(const HostPlan &h, const LaunchReq &r) {
    const bool heavy = h.mode != MODE_ALLINF;
    const TIO *in = static_cast<const TIO *>(r.in);
    TIO *out = static_cast<TIO *>(r.out);
    switch (r.op) {
    case OP_SSPRK33: {
        const unsigned g = grid_for(r.n, heavy);
        if (!h.kargs_dev) return hipErrorNotInitialized;  // uploaded by OP_PREPARE at plan creation
        const KArgs<N, P> *Ad = static_cast<const KArgs<N, P> *>(h.kargs_dev);
        if (h.mode == MODE_ALLINF)
            hipLaunchKernelGGL((ssprk33_kernel<N, P, MODE_ALLINF, TIO>), dim3(g), dim3(kBlock), 0, r.stream, Ad,
                               h.nodes_dev, r.n, r.ld, in, out, r.dt, r.n_steps);
        else if (h.mode == MODE_FIXED && N <= 2 && sorted_block_size(h) == 512) {
            if constexpr (N <= 2)
                hipLaunchKernelGGL((ssprk33_kernel<N, P, MODE_FIXED, TIO, 512>), dim3((unsigned)((r.n + 511) / 512)),
                                   dim3(512), 0, r.stream, Ad, h.nodes_dev, r.n, r.ld, in, out, r.dt, r.n_steps);
        } else if (h.mode == MODE_FIXED)
            hipLaunchKernelGGL((ssprk33_kernel<N, P, MODE_FIXED, TIO>), dim3(g), dim3(kBlock), 0, r.stream, Ad,
                               h.nodes_dev, r.n, r.ld, in, out, r.dt, r.n_steps);
        else
            hipLaunchKernelGGL((ssprk33_kernel<N, P, MODE_MOVING, TIO>), dim3(g), dim3(kBlock), 0, r.stream, Ad,
                               h.nodes_dev, r.n, r.ld, in, out, r.dt, r.n_steps);
        break;
    }
    case OP_TSIT5: {  // cloudy_tsit5_steps: thresholds Inf, fixed or moving; fp64 or float planes
        if (!h.kargs_dev) return hipErrorNotInitialized;
        const KArgs<N, P> *Ad = static_cast<const KArgs<N, P> *>(h.kargs_dev);
        const unsigned g = grid_for(r.n, heavy);
        if (h.mode == MODE_ALLINF)
            hipLaunchKernelGGL((tsit5_kernel<N, P, MODE_ALLINF, TIO>), dim3(g), dim3(kBlock), 0, r.stream, Ad, h.nodes_dev,
                               r.n, r.ld, in, out, r.dt, r.n_steps);
        else if (h.mode == MODE_FIXED)
            hipLaunchKernelGGL((tsit5_kernel<N, P, MODE_FIXED, TIO>), dim3(g), dim3(kBlock), 0, r.stream, Ad, h.nodes_dev,
                               r.n, r.ld, in, out, r.dt, r.n_steps);
        else
            hipLaunchKernelGGL((tsit5_kernel<N, P, MODE_MOVING, TIO>), dim3(g), dim3(kBlock), 0, r.stream, Ad, h.nodes_dev,
                               r.n, r.ld, in, out, r.dt, r.n_steps);
        break;
    }
    case OP_RAINSHAFT_SSPRK33: {
        if (!h.kargs_dev) return hipErrorNotInitialized;
        if (r.nz < 1 || r.nz > (size_t)kBlock) return hipErrorInvalidValue;
        const KArgs<N, P> *Ad = static_cast<const KArgs<N, P> *>(h.kargs_dev);
        const size_t cpb = kRainshaftBlock / r.nz, n_columns = r.n / r.nz;
        const unsigned g = (unsigned)((n_columns + cpb - 1) / cpb);
        if (h.mode == MODE_ALLINF)
            hipLaunchKernelGGL((rainshaft_ssprk33_kernel<N, P, MODE_ALLINF, TIO>), dim3(g), dim3(kRainshaftBlock), 0, r.stream,
                               Ad, h.nodes_dev, (int)r.nz, n_columns, r.ld, in, out, r.dt, r.dz, r.n_steps);
        else if (h.mode == MODE_FIXED)
            hipLaunchKernelGGL((rainshaft_ssprk33_kernel<N, P, MODE_FIXED, TIO>), dim3(g), dim3(kRainshaftBlock), 0, r.stream,
                               Ad, h.nodes_dev, (int)r.nz, n_columns, r.ld, in, out, r.dt, r.dz, r.n_steps);
        else
            return hipErrorInvalidValue;  // make_rainshaft_rhs is FixedThreshold only (rainshaft_helpers.jl:70)
        break;
    }
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int N, int P>
hipError_t launch_int(const HostPlan &h, const LaunchReq &r) {
    if (h.dtype != CLOUDY_F64) return launch_int_io<N, P, float>(h, r);
    return launch_int_io<N, P, double>(h, r);
}

}  // namespace cloudy
