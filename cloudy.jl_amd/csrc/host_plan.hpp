// host_plan.hpp -- host-side plan (the CoalescenceData + ODE_parameters of the reference, flattened)
// and the launch request passed to the per-N instantiation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/cloudy_hip.h"
#include "quad.hpp"

namespace cloudy {

struct HostPlan {
    int N = 0, P = 0, nmom = 0;
    int dist_type[CLOUDY_MAX_MODES] = {0}, np[CLOUDY_MAX_MODES] = {0}, off[CLOUDY_MAX_MODES] = {0};
    int finite[CLOUDY_MAX_MODES] = {0}, node_off[CLOUDY_MAX_MODES] = {0}, n_bins[CLOUDY_MAX_MODES] = {0};
    int n_2d[CLOUDY_MAX_MODES] = {0};
    int n_mom_max = 0, threshold_style = 0, nbpl = 15, mode = 0, dtype = 0;  // dtype: the PLANE type (F64 / F32 / F32_FAST)
    bool relaxed = false;  // desc.dtype == CLOUDY_F64_RELAXED: fp64 planes, series / continued fraction stopped at 1e-11
    double thr[CLOUDY_MAX_MODES] = {0};                 // Coalescence.jl:78-84
    double mom_norm[CLOUDY_MAX_MODES][3] = {{0}};       // helper_functions.jl:40-53, per (mode, order)
    double norms[2] = {1, 1};
    double kmin = 0, kmax = 10;
    double c[CLOUDY_MAX_MODES][CLOUDY_MAX_MODES][CLOUDY_MAX_P][CLOUDY_MAX_P] = {{{{0}}}};
    int n_vel = 0;
    double vel[CLOUDY_MAX_VEL][2] = {{0}};              // physical
    double vel_n[CLOUDY_MAX_VEL][2] = {{0}};            // rainshaft_helpers.jl:74-76
    // MovingThreshold: start-value fit of the percentile inversion (kernels.hpp, KArgs::inv_tab)
    double inv_map[2] = {0.0, 0.0}, inv_klo = 0.0;
    double inv_tab[CLOUDY_MAX_MODES][16] = {{0}};
    int device = 0;
    int jit_conv_rounds = 1;  // parcels per lane of the plan-time compiled converged-mode RHS kernel (quad_kernels.hpp: ROUNDS)
    int jit_conv_bs = 0;    // ... of the plan-time compiled converged-mode kernels (NumericalCoalStyle), likewise
    int jit_sorted_bs = 0;  // workgroup size of the plan-time compiled threshold kernels, fixed at plan creation (jit.hpp)
    int force_ppl1 = 0;  // CLOUDY_HIP_PPL1=1: always the one-parcel-per-lane ALLINF kernel (A/B timing)
    double *nodes_dev = nullptr;                        // [n_nodes][kNodeStride]
    int n_nodes = 0;
    double *partial_dev = nullptr;                      // moment_sums workspace
    void *kargs_dev = nullptr;                          // KArgs<N,P> of (moments in, physical out) followed by SediArgs, uploaded at plan creation
    // NumericalCoalStyle plans (quad.hpp): P = 1, no tensors, no thresholds
    int coal_style = 0;                                 // CLOUDY_ANALYTICAL_COAL / CLOUDY_NUMERICAL_COAL
    QArgs q = {};                                       // kernel function (normalised), rule order, start-value table shape
    std::vector<double> qtab;                           // quad_tab_size(q.nq, q.deg) doubles (host copy: the plan-time compiled kernel embeds it)
    double *qtab_dev = nullptr;
};

enum Op { OP_COAL = 0, OP_UPDATE_DIST = 1, OP_FINITE_2D = 2, OP_SEDI = 3, OP_SSPRK33 = 4, OP_COND = 5, OP_PREPARE = 6 /* plan creation: upload the constant block */, OP_NQ = 7, OP_RAINSHAFT_SSPRK33 = 8, OP_TSIT5 = 9 };

struct LaunchReq {
    int op;
    int input_kind;   // IN_MOMENTS / IN_PARAMS
    int physical_out; // 1: multiply by mom_norms (rhs_coal!), 0: normalised units (get_coal_ints)
    int rainshaft;    // clamp negatives + skip empty cells
    size_t n, ld;
    const void *in;   // planes of the plan's dtype (double, or float for CLOUDY_F32 plans)
    void *out;        // OP_COAL: dmom; OP_UPDATE_DIST: params; OP_FINITE_2D: F (may be null); OP_SEDI: flux
    void *out2;       // OP_FINITE_2D: thresholds (may be null)
    hipStream_t stream;
    double dt = 0.0;  // OP_SSPRK33
    int n_steps = 0;  // OP_SSPRK33
    double coef = 0.0, s_scalar = 0.0;  // OP_COND
    const double *s_dev = nullptr;      // OP_COND (optional per-parcel supersaturation)
    size_t nz = 0;    // OP_RAINSHAFT_SSPRK33: cells per column (n = nz * n_columns)
    double dz = 0.0;  // OP_RAINSHAFT_SSPRK33
};

// the fused integrators (cloudy_ssprk33_steps, cloudy_rainshaft_ssprk33_steps) live in their own units
// (int_n<N>_p<P>.hip), compiled with machine LICM off: see launch_int_impl.hpp
template <int N, int P>
hipError_t launch_int(const HostPlan &h, const LaunchReq &r);

// one per instantiation unit (inst_n<N>_p<P>.hip), so that the kernel families compile in parallel
hipError_t launch_n1_p1(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n1_p2(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n1_p3(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n1_p4(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n1_p5(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n2_p1(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n2_p2(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n2_p3(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n2_p4(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n2_p5(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n3_p1(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n3_p2(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n3_p3(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n3_p4(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n3_p5(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n4_p1(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n4_p2(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n4_p3(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n4_p4(const HostPlan &h, const LaunchReq &r);
hipError_t launch_n4_p5(const HostPlan &h, const LaunchReq &r);
// cloudy_coal_rhs / cloudy_get_coal_ints of a NumericalCoalStyle plan, ahead-of-time kernels (quad_n<N>.hip)
hipError_t launch_quad_n1(const HostPlan &h, const LaunchReq &r);
hipError_t launch_quad_n2(const HostPlan &h, const LaunchReq &r);
hipError_t launch_quad_n3(const HostPlan &h, const LaunchReq &r);
hipError_t launch_quad_n4(const HostPlan &h, const LaunchReq &r);

}  // namespace cloudy
