// cloudy_hip.hip -- the C ABI of libcloudy_hip.so (include/cloudy_hip.h): plan construction on the host,
// argument checking, and dispatch to the gfx950 kernels.  No CPU compute path exists here: every batched
// entry point launches HIP kernels or fails with CLOUDY_ENODEVICE / CLOUDY_EHIP.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <map>
#include <mutex>
#include <mutex>
#include <string>
#include <utility>

#include "host_gamma.hpp"
#include "host_plan.hpp"
#include "host_quad.hpp"
#include "kernels.hpp"
#include "jit.hpp"
#include "reduce_kernels.hpp"

using namespace cloudy;

struct cloudy_plan {
    HostPlan h;
    bool jit_on = false;   // plan-time specialised kernels (jit.hpp) serve cloudy_coal_rhs / cloudy_ssprk33_steps
    JitKernels jit;
    std::string jit_log;   // why not, when jit_on is false
    // thresholded plans compile their fused integrator on the first cloudy_ssprk33_steps call
    mutable std::once_flag int_once, rs_once, rsint_once, tsit5_once, rsint512_once, rsint1024_once, rsint320_once;
    mutable hipFunction_t int_ssprk33 = nullptr, rs_coal = nullptr, rs_int = nullptr, int_tsit5 = nullptr;
    mutable hipFunction_t rs_int512 = nullptr, rs_int1024 = nullptr;  // the column integrator for 256 < nz <= 512 / 1024
    mutable hipFunction_t rs_int320 = nullptr;                        // ... with 320-thread workgroups (jit_rainshaft_part)
    mutable hipFunction_t rs_rhs = nullptr, rs_rhs512 = nullptr, rs_rhs1024 = nullptr;  // one RHS evaluation, same modules
    // one log per once-flag (a single string written from seven independent call_once lambdas raced when two host
    // threads made first calls to different entry points of one plan, and a later successful compile overwrote the log an
    // earlier failure's message refers to); each is written once, inside its call_once, and read only after it
    mutable std::string int_log, rs_log, rsint_log, rsint512_log, rsint1024_log, rsint320_log, tsit5_log, diag_log;
    // plans beyond the ahead-of-time families: diagnostics and parameter-plane entry points compiled on first use (jit.hpp part 7)
    mutable std::once_flag diag_once;
    mutable JitDiag diag;
    // NumericalCoalStyle plans in converged mode: the cost hints of coal_rhs_quad_body (quad_kernels.hpp) -- one byte per parcel,
    // two for a Long-kernel plan (second plane at hint + cap).  Allocated on the first cloudy_coal_rhs call, grown (x 2 at least)
    // when a larger batch arrives, never per call otherwise; CLOUDY_HIP_CONV_HINTS=0 turns the mechanism off.  Kernels of
    // concurrent calls may race on the bytes: a hint only orders lanes.  ADVICE r5 (medium): a buffer a launched kernel may still
    // read is NEVER freed before cloudy_plan_destroy -- a superseded buffer waits in `hint_old` (geometric growth bounds the list
    // and its bytes by the final size); a stream that is being captured gets no hints (hipMalloc is not capturable).
    mutable unsigned char *hint_dev = nullptr;
    mutable size_t hint_cap = 0;
    mutable std::vector<unsigned char *> hint_old;
    mutable std::mutex hint_mu;
};

namespace {
struct ConvHints {
    unsigned char *p1 = nullptr, *p2 = nullptr;   // nullptr: off, or the allocation failed -- the kernel then keeps the natural order
};
ConvHints conv_hints(const cloudy_plan *plan, size_t n, hipStream_t stream) {
    static const bool on = [] {
        const char *e = std::getenv("CLOUDY_HIP_CONV_HINTS");
        return !(e && e[0] == '0');
    }();
    ConvHints r;
    if (!on || n == 0) return r;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return r;
    }
    if (cs != hipStreamCaptureStatusNone) return r;   // (a captured launch would bake this call's pointers into the graph)
    const bool two = plan->h.q.kind == KF_LONG;
    std::lock_guard<std::mutex> lock(plan->hint_mu);
    if (plan->hint_cap < n) {
        const size_t cap = std::max(n, 2 * plan->hint_cap), bytes = (two ? 2 : 1) * cap;
        unsigned char *p = nullptr;
        if (hipMalloc((void **)&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return r;
        }
        if (hipMemsetAsync(p, 0, bytes, stream) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(p);
            return r;
        }
        if (plan->hint_dev) {
            // the costs the plan has learnt so far move over (stream-ordered after the memset): a batch that grew keeps the ranking
            // of the parcels it already had
            (void)hipMemcpyAsync(p, plan->hint_dev, plan->hint_cap, hipMemcpyDeviceToDevice, stream);
            if (two) (void)hipMemcpyAsync(p + cap, plan->hint_dev + plan->hint_cap, plan->hint_cap, hipMemcpyDeviceToDevice, stream);
            (void)hipGetLastError();
            plan->hint_old.push_back(plan->hint_dev);   // kernels of earlier calls may still use it
        }
        plan->hint_dev = p;
        plan->hint_cap = cap;
    }
    r.p1 = plan->hint_dev;
    r.p2 = two ? plan->hint_dev + plan->hint_cap : nullptr;
    return r;
}
}  // namespace

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int fail_hip(hipError_t e, const char *what) {
    const int code = (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? CLOUDY_ENODEVICE
                     : (e == hipErrorOutOfMemory)                             ? CLOUDY_ENOMEM
                                                                              : CLOUDY_EHIP;
    return fail(code, "%s: %s", what, hipGetErrorString(e));
}

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return fail_hip(_e, #expr); \
    } while (0)

constexpr int kSumBlocks = 1024;

inline bool plan_beyond_aot(const HostPlan &h) { return h.N > CLOUDY_AOT_MAX_MODES || h.P > CLOUDY_AOT_MAX_P; }

hipError_t dispatch(const HostPlan &h, const LaunchReq &r) {
    using Fn = hipError_t (*)(const HostPlan &, const LaunchReq &);
    static const Fn table[CLOUDY_AOT_MAX_MODES][CLOUDY_AOT_MAX_P] = {
        {launch_n1_p1, launch_n1_p2, launch_n1_p3, launch_n1_p4, launch_n1_p5},
        {launch_n2_p1, launch_n2_p2, launch_n2_p3, launch_n2_p4, launch_n2_p5},
        {launch_n3_p1, launch_n3_p2, launch_n3_p3, launch_n3_p4, launch_n3_p5},
        {launch_n4_p1, launch_n4_p2, launch_n4_p3, launch_n4_p4, launch_n4_p5}};
    if (h.N < 1 || h.N > CLOUDY_AOT_MAX_MODES || h.P < 1 || h.P > CLOUDY_AOT_MAX_P) return hipErrorInvalidValue;
    if (h.coal_style == CLOUDY_NUMERICAL_COAL && (r.op == OP_COAL || r.op == OP_SSPRK33)) {
        static const Fn quad[CLOUDY_AOT_MAX_MODES] = {launch_quad_n1, launch_quad_n2, launch_quad_n3, launch_quad_n4};
        return quad[h.N - 1](h, r);
    }
    return table[h.N - 1][h.P - 1](h, r);
}

// A plan belongs to one device (its constant block, node tables and code objects live there): every entry point that
// launches makes that device current for the call and restores the caller's device on return.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

int check_batch(const cloudy_plan *plan, size_t n, size_t ld, const void *a, const void *b) {
    if (!plan) return fail(CLOUDY_EINVAL, "plan is NULL");
    if (ld < n) return fail(CLOUDY_EINVAL, "ld (%zu) must be >= n_parcels (%zu)", ld, n);
    if (n > 0 && (!a || !b)) return fail(CLOUDY_EINVAL, "device buffer is NULL");
    return CLOUDY_OK;
}

// The column kernel compiled for the plan that serves columns of nz cells: the workgroup size jit_rainshaft_part prefers, then
//  every smaller one that still holds a column and fits the LDS -- a kernel that does not compile or load is not the
// end of the call while another can serve it.  Compiles on first use.  fn == nullptr: none (the caller steps stage by stage, or
// runs the ahead-of-time integrator).  rhs_only: the one-evaluation kernel of the same translation unit (cloudy_rainshaft_rhs).
struct RsPick {
    hipFunction_t fn = nullptr;
    int bs = 0;
};
RsPick pick_rainshaft(const cloudy_plan *plan, size_t nz, size_t n_columns, bool rhs_only) {
    RsPick r;
    if (!plan->jit_on || nz < 1 || nz > 1024) return r;
    const HostPlan &h = plan->h;
    const auto get = [&](int part) -> hipFunction_t {
        switch (part) {
        case 3:
            std::call_once(plan->rsint_once, [&] { (void)jit_get_rainshaft_integrator(h, plan->rs_int, plan->rsint_log, 3, &plan->rs_rhs); });
            return rhs_only ? plan->rs_rhs : plan->rs_int;
        case 5:
            std::call_once(plan->rsint512_once, [&] { (void)jit_get_rainshaft_integrator(h, plan->rs_int512, plan->rsint512_log, 5, &plan->rs_rhs512); });
            return rhs_only ? plan->rs_rhs512 : plan->rs_int512;
        case 6:
            std::call_once(plan->rsint1024_once, [&] { (void)jit_get_rainshaft_integrator(h, plan->rs_int1024, plan->rsint1024_log, 6, &plan->rs_rhs1024); });
            return rhs_only ? plan->rs_rhs1024 : plan->rs_int1024;
        case 8:   // (CLOUDY_HIP_RS_BLOCK=320: an experiment switch of the integrator only)
            if (rhs_only) return nullptr;
            std::call_once(plan->rsint320_once, [&] { (void)jit_get_rainshaft_integrator(h, plan->rs_int320, plan->rsint320_log, 8); });
            return plan->rs_int320;
        default: return nullptr;
        }
    };
    int part = jit_rainshaft_part(h, nz, n_columns);
    if (part == 8) {
        if ((r.fn = get(8)) != nullptr) {
            r.bs = jit_rainshaft_block(8);
            return r;
        }
        part = jit_rainshaft_part(h, nz, n_columns, false);   // (the normal pick for this height, not unconditionally 256 threads)
    }
    const int order[3] = {6, 5, 3};
    bool reached = false;
    for (int i = 0; i < 3; ++i) {
        if (order[i] == part) reached = true;
        if (!reached || !jit_rainshaft_fits(h, order[i], nz)) continue;
        if ((r.fn = get(order[i])) != nullptr) {
            r.bs = jit_rainshaft_block(order[i]);
            return r;
        }
    }
    return r;
}

// cloudy_coal_rhs / cloudy_ssprk33_steps of a specialised plan: same grids and the same two-parcels-per-lane
// alignment rule as launch_io() in launch_impl.hpp.
hipError_t launch_jit(const cloudy_plan *plan, const LaunchReq &r) {
    const HostPlan &h = plan->h;
    size_t n = r.n, ld = r.ld;
    const void *in = r.in;
    void *out = r.out;
    const unsigned g1 = (unsigned)((n + kBlock - 1) / kBlock);
    if (r.op == OP_RAINSHAFT_SSPRK33) {  // as launch_int_io() in launch_int_impl.hpp: whole columns per workgroup
        const double *nodes = h.nodes_dev;
        int nz = (int)r.nz, n_steps = r.n_steps;
        size_t n_columns = r.n / r.nz;
        double dt = r.dt, dz = r.dz;
        const RsPick pk = pick_rainshaft(plan, r.nz, r.n / r.nz, false);   // (as run() decided: compiled by then)
        if (pk.fn == nullptr) return hipErrorInvalidValue;
        const unsigned bs = (unsigned)pk.bs;
        hipFunction_t fn = pk.fn;
        const size_t cpb = bs / r.nz;
        void *args[] = {&nodes, &nz, &n_columns, &ld, &in, &out, &dt, &dz, &n_steps};
        return hipModuleLaunchKernel(fn, (unsigned)((n_columns + cpb - 1) / cpb), 1, 1, bs, 1, 1, 0, r.stream, args, nullptr);
    }
    if (r.op == OP_TSIT5) {
        double dt = r.dt;
        int n_steps = r.n_steps;
        if (h.coal_style == CLOUDY_NUMERICAL_COAL) {
            const unsigned qb = h.q.mode == QUAD_CONVERGED ? (unsigned)jit_conv_block_size(h) : (unsigned)quad_block(h.q.nq);
            ConvHints hints = h.q.mode == QUAD_CONVERGED ? conv_hints(plan, n, r.stream) : ConvHints{};
            void *args[] = {&n, &ld, &in, &out, &dt, &n_steps, &hints.p1, &hints.p2};
            return hipModuleLaunchKernel(plan->int_tsit5, (unsigned)((n + qb - 1) / qb), 1, 1, qb, 1, 1, 0, r.stream, args, nullptr);
        }
        const double *nodes = h.nodes_dev;
        void *args[] = {&nodes, &n, &ld, &in, &out, &dt, &n_steps};
        return hipModuleLaunchKernel(plan->int_tsit5, g1, 1, 1, kBlock, 1, 1, 0, r.stream, args, nullptr);
    }
    if (h.coal_style == CLOUDY_NUMERICAL_COAL) {
        const unsigned qb = h.q.mode == QUAD_CONVERGED ? (unsigned)jit_conv_block_size(h) : (unsigned)quad_block(h.q.nq);
        if (r.op == OP_SSPRK33) {
            double dt = r.dt;
            int n_steps = r.n_steps;
            ConvHints hints = h.q.mode == QUAD_CONVERGED ? conv_hints(plan, n, r.stream) : ConvHints{};
            void *args[] = {&n, &ld, &in, &out, &dt, &n_steps, &hints.p1, &hints.p2};
            return hipModuleLaunchKernel(plan->int_ssprk33, (unsigned)((n + qb - 1) / qb), 1, 1, qb, 1, 1, 0, r.stream, args,
                                         nullptr);
        }
        ConvHints hints = h.q.mode == QUAD_CONVERGED ? conv_hints(plan, n, r.stream) : ConvHints{};
        void *args[] = {&n, &ld, &in, &out, &hints.p1, &hints.p2};
        const size_t per_wg = (size_t)qb * (size_t)(h.q.mode == QUAD_CONVERGED ? h.jit_conv_rounds : 1);
        return hipModuleLaunchKernel(plan->jit.quad, (unsigned)((n + per_wg - 1) / per_wg), 1, 1, qb, 1, 1, 0, r.stream, args,
                                     nullptr);
    }
    if (r.op == OP_SSPRK33) {
        double dt = r.dt;
        int n_steps = r.n_steps;
        if (h.mode != MODE_ALLINF) {
            const double *nodes = h.nodes_dev;
            void *args[] = {&nodes, &n, &ld, &in, &out, &dt, &n_steps};
            const unsigned bs = (unsigned)jit_sorted_block_size(h);
            return hipModuleLaunchKernel(plan->int_ssprk33, (unsigned)((n + bs - 1) / bs), 1, 1, bs, 1, 1, 0, r.stream, args,
                                         nullptr);
        }
        void *args[] = {&n, &ld, &in, &out, &dt, &n_steps};
        return hipModuleLaunchKernel(plan->jit.ssprk33, g1, 1, 1, kBlock, 1, 1, 0, r.stream, args, nullptr);
    }
    void *out2 = r.out2;  // rainshaft cell body: the sedimentation flux planes (cloudy_rainshaft_sources), or null
    if (h.mode != MODE_ALLINF) {
        const double *nodes = h.nodes_dev;
        void *args[] = {&nodes, &n, &ld, &in, &out, &out2};  // (the plain kernel takes the first five)
        const unsigned bs = (unsigned)jit_sorted_block_size(h);
        return hipModuleLaunchKernel(r.rainshaft ? plan->rs_coal : plan->jit.sorted, (unsigned)((n + bs - 1) / bs), 1, 1, bs, 1,
                                     1, 0, r.stream, args, nullptr);
    }
    if (r.rainshaft) {
        void *args[] = {&n, &ld, &in, &out, &out2};
        return hipModuleLaunchKernel(plan->rs_coal, g1, 1, 1, kBlock, 1, 1, 0, r.stream, args, nullptr);
    }
    const size_t esz = h.dtype != CLOUDY_F64 ? sizeof(float) : sizeof(double);
    const uintptr_t amask = 2 * esz - 1;
    const bool aligned2 =
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & amask) == 0 && (ld % 2 == 0);
    void *args[] = {&n, &ld, &in, &out};
    if (plan->jit.allinf4_f32 && !h.force_ppl1) {
        // CLOUDY_F32_FAST without thresholds: the packed single-precision kernel for EVERY layout (16-B accesses where the
        // planes allow them, scalar ones otherwise): the arithmetic is a property of the plan, not of the batch's alignment
        int aligned16 = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
        void *args4[] = {&n, &ld, &in, &out, &aligned16};
        const unsigned g4 = (unsigned)(((n + 3) / 4 + kBlock - 1) / kBlock);
        return hipModuleLaunchKernel(plan->jit.allinf4_f32, g4, 1, 1, kBlock, 1, 1, 0, r.stream, args4, nullptr);
    }
    if (aligned2 && !h.force_ppl1) {
        const unsigned g2 = (unsigned)(((n + 1) / 2 + kBlock - 1) / kBlock);
        return hipModuleLaunchKernel(plan->jit.allinf2, g2, 1, 1, kBlock, 1, 1, 0, r.stream, args, nullptr);
    }
    return hipModuleLaunchKernel(plan->jit.ppl1, g1, 1, 1, kBlock, 1, 1, 0, r.stream, args, nullptr);
}

// the diagnostics of a plan beyond the ahead-of-time families: argument lists as the kernels of launch_impl.hpp take them,
// minus the constant block (a compile-time object of the module)
hipError_t launch_diag(const cloudy_plan *plan, const LaunchReq &r) {
    const HostPlan &h = plan->h;
    size_t n = r.n, ld = r.ld;
    const void *in = r.in;
    void *out = r.out, *out2 = r.out2;
    const double *nodes = h.nodes_dev;
    const unsigned g = (unsigned)((n + kBlock - 1) / kBlock);
    hipFunction_t fn = nullptr;
    void *args[8];
    int na = 0;
    double d0 = 0.0, d1 = 0.0, d2 = 0.0;
    const double *s_dev = r.s_dev;
    switch (r.op) {
    case OP_UPDATE_DIST:
        fn = plan->diag.update_dist;
        args[na++] = &n, args[na++] = &ld, args[na++] = &in, args[na++] = &out;
        break;
    case OP_FINITE_2D:
        fn = plan->diag.finite_2d;
        args[na++] = &nodes, args[na++] = &n, args[na++] = &ld, args[na++] = &in, args[na++] = &out, args[na++] = &out2;
        break;
    case OP_SEDI:
        fn = plan->diag.sedi_flux;
        args[na++] = &n, args[na++] = &ld, args[na++] = &in, args[na++] = &out;
        break;
    case OP_COND:
        fn = plan->diag.cond_evap;
        d0 = r.coef, d1 = r.s_scalar;
        args[na++] = &d0, args[na++] = &d1, args[na++] = &s_dev, args[na++] = &n, args[na++] = &ld, args[na++] = &in, args[na++] = &out;
        break;
    case OP_NQ:
        fn = plan->diag.standard_nq;
        d0 = r.s_scalar / h.norms[1], d1 = h.norms[0], d2 = h.norms[1];
        args[na++] = &d0, args[na++] = &d1, args[na++] = &d2, args[na++] = &n, args[na++] = &ld, args[na++] = &in, args[na++] = &out;
        break;
    case OP_COAL:
        fn = plan->diag.coal_ints;
        args[na++] = &nodes, args[na++] = &n, args[na++] = &ld, args[na++] = &in, args[na++] = &out;
        break;
    default: return hipErrorInvalidValue;
    }
    return hipModuleLaunchKernel(fn, g, 1, 1, kBlock, 1, 1, 0, r.stream, args, nullptr);
}

int run(const cloudy_plan *plan, const LaunchReq &r) {
    if (plan->h.coal_style == CLOUDY_NUMERICAL_COAL &&
        (r.op == OP_FINITE_2D || r.op == OP_RAINSHAFT_SSPRK33 || r.rainshaft))
        return fail(CLOUDY_EUNSUPPORTED,
                    "NumericalCoalStyle plans serve cloudy_coal_rhs / cloudy_get_coal_ints / cloudy_ssprk33_steps and the "
                    "per-mode diagnostics; thresholds and the rainshaft body belong to AnalyticalCoalStyle plans");
    if (r.n == 0) return CLOUDY_OK;
    DeviceGuard guard(plan->h.device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "selecting the plan's device");
    bool use_jit = plan->jit_on && r.input_kind == IN_MOMENTS && r.physical_out &&
                   (r.op == OP_COAL || (r.op == OP_SSPRK33 && !r.rainshaft) || r.op == OP_RAINSHAFT_SSPRK33 || r.op == OP_TSIT5);
    if (use_jit && r.op == OP_TSIT5) {
        std::call_once(plan->tsit5_once, [&] { (void)jit_get_tsit5(plan->h, plan->int_tsit5, plan->tsit5_log); });
        use_jit = plan->int_tsit5 != nullptr;  // otherwise the ahead-of-time integrator (tensor plans)
    }
    if (!use_jit && r.op == OP_TSIT5 && plan->h.coal_style == CLOUDY_NUMERICAL_COAL)
        return fail(CLOUDY_EUNSUPPORTED, "cloudy_tsit5_steps of a NumericalCoalStyle plan runs the kernel compiled for the plan "
                                         "(hiprtc); plan-time compilation is off or failed: %s", plan->tsit5_log.c_str());
    if (use_jit && r.op == OP_RAINSHAFT_SSPRK33)   // otherwise the ahead-of-time integrator (nz <= 256, the ahead-of-time families)
        use_jit = plan->h.mode != MODE_MOVING && pick_rainshaft(plan, r.nz, r.n / r.nz, false).fn != nullptr;
    if (!use_jit && r.op == OP_RAINSHAFT_SSPRK33 && r.nz > (size_t)kRainshaftBlock)
        return fail(CLOUDY_EUNSUPPORTED, "columns of %zu cells run the column integrator compiled for the plan (hiprtc, up to 1024 "
                                         "cells); plan-time compilation is off or failed: %s", r.nz,
                    (r.nz <= 512 ? plan->rsint512_log : plan->rsint1024_log).c_str());
    if (use_jit && r.op == OP_COAL && r.rainshaft) {
        std::call_once(plan->rs_once, [&] { (void)jit_get_rainshaft(plan->h, plan->rs_coal, plan->rs_log); });
        use_jit = plan->rs_coal != nullptr;
    }
    if (use_jit && r.op == OP_SSPRK33 && (plan->h.mode != MODE_ALLINF || plan->h.coal_style == CLOUDY_NUMERICAL_COAL)) {
        std::call_once(plan->int_once,
                       [&] { (void)jit_get_integrator(plan->h, plan->int_ssprk33, plan->int_log); });
        use_jit = plan->int_ssprk33 != nullptr;  // otherwise the ahead-of-time integrator
    }
    if (use_jit) {
        hipError_t e = launch_jit(plan, r);
        if (e != hipSuccess) return fail_hip(e, "specialised kernel launch");
        return CLOUDY_OK;
    }
    if (plan_beyond_aot(plan->h)) {
        // no ahead-of-time kernel exists for this (N, P): the diagnostics and the parameter-plane entry points run the
        // bodies of their ahead-of-time kernels compiled for the plan (jit.hpp part 7, on first use)
        const bool diag_op = !r.rainshaft && (r.op == OP_UPDATE_DIST || r.op == OP_FINITE_2D || r.op == OP_SEDI || r.op == OP_COND ||
                                              r.op == OP_NQ || (r.op == OP_COAL && r.input_kind == IN_PARAMS));
        if (diag_op && plan->jit_on) {
            std::call_once(plan->diag_once, [&] { (void)jit_get_diag(plan->h, plan->diag, plan->diag_log); });
            if (plan->diag.ok) {
                hipError_t e = launch_diag(plan, r);
                if (e != hipSuccess) return fail_hip(e, "diagnostic kernel launch");
                return CLOUDY_OK;
            }
        }
        return fail(CLOUDY_EUNSUPPORTED, "plans of more than %d modes or order > %d are served by the kernels compiled for the plan "
                                         "(this call has none, or its compilation failed: %s)",
                    CLOUDY_AOT_MAX_MODES, CLOUDY_AOT_MAX_P - 1, plan->diag_log.c_str());
    }
    hipError_t e = dispatch(plan->h, r);
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return CLOUDY_OK;
}

}  // namespace

extern "C" {

// (internal: cloudy_comm.hip reports its failures through the same thread-local message)
void cloudy_set_last_error_(const char *msg) { std::snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

const char *cloudy_last_error(void) { return g_err; }
int cloudy_version(void) { return CLOUDY_HIP_VERSION; }
unsigned long long cloudy_source_hash(int which) {
    // FNV-1a over the kernel sources that travel inside the library (embedded_src.inc): bench.py prints a committed PMC figure
    // of a kernel only while the sources it was collected on are the ones in the library
    const auto fnv = [](unsigned long long h, const char *s) {
        for (; *s; ++s) h = (h ^ (unsigned char)*s) * 1099511628211ull;
        return h;
    };
    using namespace cloudy::jit_detail;
    unsigned long long h = 14695981039346656037ull;
    h = fnv(fnv(h, kSrcKernels), kSrcDeviceMath);
    if (which == 1) return fnv(h, kSrcBodyAllinf2);   // what the all-Inf kernels (the headline) are compiled from
    for (const char *s : {kSrcBodyPpl1, kSrcBodyAllinf2, kSrcBodySorted, kSrcQuad, kSrcQuadKernels, kSrcQuadConv, kSrcAllinfF32}) h = fnv(h, s);
    return h;
}

void cloudy_plan_desc_init(cloudy_plan_desc *d) {
    if (!d) return;
    std::memset(d, 0, sizeof(*d));
    d->struct_size = (uint32_t)sizeof(*d);
    d->norms[0] = d->norms[1] = 1.0;
    d->k_range[0] = DBL_EPSILON;
    d->k_range[1] = 10.0;  // ParticleDistributions.jl:459
    d->n_bins_per_log_unit = 15;  // ParticleDistributions.jl:594
    for (int i = 0; i < CLOUDY_MAX_MODES; ++i) d->dist_thresholds[i] = INFINITY;
    d->dtype = CLOUDY_F64;
    d->device = -1;
    d->coal_style = CLOUDY_ANALYTICAL_COAL;
    d->quad_order = 0;                     // the mode's default: 8 (CONVERGED) / 10 (FIXED)
    d->quad_mode = CLOUDY_QUAD_CONVERGED;  // the drop-in default meets the reference's quadgk(rtol = 1e-8) answer
}

}  // extern "C"

namespace {

// Host half of cloudy_plan_create: validation and every derived constant (no device involved).  On success *out owns
// a plan without device resources and `nodes` holds the Simpson node tables.
int build_host_plan(const cloudy_plan_desc *d, cloudy_plan **out, std::vector<double> &nodes) {
    if (!d || !out) return fail(CLOUDY_EINVAL, "desc/out is NULL");
    *out = nullptr;
    if (d->struct_size != sizeof(cloudy_plan_desc))
        return fail(CLOUDY_EINVAL, "struct_size %u != %zu (call cloudy_plan_desc_init)", d->struct_size,
                    sizeof(cloudy_plan_desc));
    if (d->coal_style != CLOUDY_ANALYTICAL_COAL && d->coal_style != CLOUDY_NUMERICAL_COAL)
        return fail(CLOUDY_EINVAL, "Invalid coal style!");  // box_model_helpers.jl:49-50
    const bool numerical = d->coal_style == CLOUDY_NUMERICAL_COAL;
    const int N = d->n_modes, P = numerical ? 1 : d->tensor_p;
    if (N < 1 || N > CLOUDY_MAX_MODES) return fail(CLOUDY_EUNSUPPORTED, "n_modes %d outside 1..%d", N, CLOUDY_MAX_MODES);
    if (P < 1 || P > CLOUDY_MAX_P) return fail(CLOUDY_EUNSUPPORTED, "tensor_p %d outside 1..%d", P, CLOUDY_MAX_P);
    if (!numerical && !d->kernel_c) return fail(CLOUDY_EINVAL, "kernel_c is NULL");
    if (numerical) {
        if (d->kernel_func < CLOUDY_KFUNC_CONSTANT || d->kernel_func > CLOUDY_KFUNC_LONG)
            return fail(CLOUDY_EINVAL, "kernel_func %d is not a CoalescenceKernelFunction family", d->kernel_func);
        if (d->quad_mode != CLOUDY_QUAD_FIXED && d->quad_mode != CLOUDY_QUAD_CONVERGED)
            return fail(CLOUDY_EINVAL, "quad_mode %d is neither CLOUDY_QUAD_FIXED nor CLOUDY_QUAD_CONVERGED", d->quad_mode);
        if (d->quad_order != 0 && (d->quad_order < 2 || d->quad_order > CLOUDY_MAX_QUAD))
            return fail(CLOUDY_EUNSUPPORTED, "quad_order %d outside 2..%d (0 = the mode's default)", d->quad_order, CLOUDY_MAX_QUAD);
        if (d->dtype == CLOUDY_F32_FAST)
            return fail(CLOUDY_EUNSUPPORTED, "CLOUDY_F32_FAST is the single-precision Simpson pass of threshold plans");
        for (int i = 0; i < N; ++i)
            if (d->dist_type[i] == CLOUDY_DIST_MONODISPERSE)
                return fail(CLOUDY_EINVAL, "no method normed_density_func for a Monodisperse distribution "
                                           "(weighting_fn, Coalescence.jl:624-642)");
    }
    if (d->dtype != CLOUDY_F64 && d->dtype != CLOUDY_F32 && d->dtype != CLOUDY_F32_FAST && d->dtype != CLOUDY_F64_RELAXED)
        return fail(CLOUDY_EINVAL, "bad dtype");
    if (!(d->norms[0] > 0) || !(d->norms[1] > 0))
        return fail(CLOUDY_EINVAL, "norms must be positive!");  // helper_functions.jl:44-46
    if (d->threshold_style != CLOUDY_FIXED_THRESHOLD && d->threshold_style != CLOUDY_MOVING_THRESHOLD)
        return fail(CLOUDY_EINVAL, "bad threshold_style");
    if (d->n_bins_per_log_unit < 1) return fail(CLOUDY_EINVAL, "n_bins_per_log_unit must be >= 1");
    if (!(d->k_range[0] > 0) || !(d->k_range[1] >= d->k_range[0])) return fail(CLOUDY_EINVAL, "bad k_range");
    if (d->n_vel < 0 || d->n_vel > CLOUDY_MAX_VEL) return fail(CLOUDY_EUNSUPPORTED, "n_vel outside 0..%d", CLOUDY_MAX_VEL);

    cloudy_plan *p = new (std::nothrow) cloudy_plan();
    if (!p) return fail(CLOUDY_ENOMEM, "out of host memory");
    HostPlan &h = p->h;
    h.N = N;
    h.P = P;
    h.relaxed = d->dtype == CLOUDY_F64_RELAXED;
    h.dtype = h.relaxed ? (int)CLOUDY_F64 : d->dtype;
    h.threshold_style = d->threshold_style;
    h.nbpl = d->n_bins_per_log_unit;
    h.kmin = d->k_range[0];
    h.kmax = d->k_range[1];
    h.norms[0] = d->norms[0];
    h.norms[1] = d->norms[1];
    int off = 0, np_max = 0;
    for (int i = 0; i < N; ++i) {
        if (d->dist_type[i] == CLOUDY_DIST_EXPONENTIAL || d->dist_type[i] == CLOUDY_DIST_MONODISPERSE)
            h.np[i] = 2;
        else if (d->dist_type[i] == CLOUDY_DIST_GAMMA || d->dist_type[i] == CLOUDY_DIST_LOGNORMAL)
            h.np[i] = 3;
        else {
            delete p;
            return fail(CLOUDY_EINVAL, "dist_type[%d] = %d is not a closure family", i, d->dist_type[i]);
        }
        h.dist_type[i] = d->dist_type[i];
        h.off[i] = off;
        off += h.np[i];
        if (h.np[i] > np_max) np_max = h.np[i];
        // get_moments_normalizing_factors, helper_functions.jl:40-53: norms[1] * norms[2]^(j-1)
        h.mom_norm[i][0] = d->norms[0];
        h.mom_norm[i][1] = d->norms[0] * d->norms[1];
        h.mom_norm[i][2] = d->norms[0] * (d->norms[1] * d->norms[1]);
    }
    h.nmom = off;

    if (numerical) {
        // get_normalized_kernel_func, KernelFunctions.jl:124-154
        h.coal_style = CLOUDY_NUMERICAL_COAL;
        h.q.kind = d->kernel_func;
        const double *kp = d->kernel_func_params;
        const double n0 = d->kernel_func_is_normalized ? 1.0 : d->norms[0], m0 = d->kernel_func_is_normalized ? 1.0 : d->norms[1];
        h.q.kf[0] = h.q.kf[1] = h.q.kf[2] = 0.0;
        switch (d->kernel_func) {
        case CLOUDY_KFUNC_CONSTANT: h.q.kf[0] = kp[0] * n0; break;
        case CLOUDY_KFUNC_LINEAR: h.q.kf[0] = kp[0] * n0 * m0; break;
        case CLOUDY_KFUNC_HYDRODYNAMIC: h.q.kf[0] = kp[0] * n0 * std::pow(m0, 4.0 / 3.0); break;
        default:
            h.q.kf[0] = kp[0] / m0;
            h.q.kf[1] = kp[1] * n0 * (m0 * m0);
            h.q.kf[2] = kp[2] * n0 * m0;
        }
        for (int k = 0; k < 3; ++k)
            if (std::isnan(h.q.kf[k])) {
                delete p;
                return fail(CLOUDY_EINVAL, "kernel_func_params[%d] is NaN", k);
            }
        const int quad_order = d->quad_order ? d->quad_order : (d->quad_mode == CLOUDY_QUAD_CONVERGED ? 8 : 10);
        if (d->quad_mode == CLOUDY_QUAD_CONVERGED) {
            // quad_conv.hpp: quad_order Gauss-Legendre points per panel of the inner rule of a Lognormal mode's T_m (the
            // adaptive rules carry their own Gauss-Kronrod nodes)
            h.q.mode = QUAD_CONVERGED;
            h.q.nq = quad_order;
            h.q.deg = 0;
            h.q.t_scale = 0.0;
            h.qtab.assign((size_t)2 * quad_order, 0.0);
            quad_host::legendre_rule(quad_order, h.qtab.data(), h.qtab.data() + quad_order);
        } else {
            // start-value table of the per-parcel Gauss-Laguerre rules over k in (0, max(k_range[1], 1)] (Exponential: k = 1)
            std::string msg;
            if (!quad_host::build_table(quad_order, std::fmax(d->k_range[1], 1.0), h.q, h.qtab, msg)) {
                delete p;
                return fail(CLOUDY_EUNSUPPORTED, "%s", msg.c_str());
            }
            h.q.mode = QUAD_FIXED;
        }
        h.n_mom_max = np_max;
        for (int i = 0; i < N; ++i) {
            h.n_2d[i] = 0;
            h.thr[i] = INFINITY;
        }
        h.mode = MODE_ALLINF;
        h.n_vel = d->n_vel;
        for (int v = 0; v < d->n_vel; ++v) {
            h.vel[v][0] = d->vel[v][0];
            h.vel[v][1] = d->vel[v][1];
            h.vel_n[v][0] = d->vel[v][0] * std::pow(d->norms[1], d->vel[v][1]);
            h.vel_n[v][1] = d->vel[v][1];
        }
        nodes.clear();
        h.jit_conv_bs = jit_conv_block_size_resolve(h);
        h.jit_conv_rounds = jit_conv_rounds_resolve(h);
        *out = p;
        return CLOUDY_OK;
    }
    // kernels: check_symmetry (KernelTensors.jl:157-171) + get_normalized_kernel_tensor (:189-199)
    for (int j = 0; j < N; ++j)
        for (int k = 0; k < N; ++k) {
            const double *c =
                d->kernel_layout == CLOUDY_KERNEL_MATRIX ? d->kernel_c + (size_t)(j * N + k) * P * P : d->kernel_c;
            for (int a = 0; a < P; ++a)
                for (int b = a + 1; b < P; ++b)
                    if (c[a * P + b] != c[b * P + a]) {
                        delete p;
                        return fail(CLOUDY_ENOTSYMMETRIC, "array not symmetric.");
                    }
            for (int a = 0; a < P; ++a)
                for (int b = 0; b < P; ++b)
                    h.c[j][k][a][b] = d->kernel_is_normalized
                                          ? c[a * P + b]
                                          : c[a * P + b] * (d->norms[0] * std::pow(d->norms[1], (double)(a + b)));
        }

    // CoalescenceData fields, Coalescence.jl:69-84
    h.n_mom_max = np_max + (P - 1);
    for (int i = 0; i < N; ++i) {
        const int nxt = (i < N - 1) ? (h.np[i] > h.np[i + 1] ? h.np[i] : h.np[i + 1]) : h.np[i];
        h.n_2d[i] = (P - 1) + nxt;
        h.thr[i] = (d->threshold_style == CLOUDY_FIXED_THRESHOLD && !d->thresholds_are_normalized)
                       ? d->dist_thresholds[i] / d->norms[1]
                       : d->dist_thresholds[i];
        if (std::isnan(h.thr[i])) {
            delete p;
            return fail(CLOUDY_EINVAL, "dist_thresholds[%d] is NaN", i);
        }
    }

    // Simpson node tables for FixedThreshold (ParticleDistributions.jl:604-610): the grid depends only on
    // the (normalised) threshold, so it is built once here, in fp64, exactly as the reference builds it.
    nodes.clear();
    bool any_finite = false;
    if (d->threshold_style == CLOUDY_FIXED_THRESHOLD) {
        for (int i = 0; i < N - 1; ++i) {
            const double xt = h.thr[i];
            if (std::isinf(xt) && xt > 0) continue;
            if (!(xt > 0)) {
                delete p;
                return fail(CLOUDY_EINVAL, "dist_thresholds[%d] must be positive or +Inf", i);
            }
            if (h.dist_type[i] == CLOUDY_DIST_MONODISPERSE ||  // analytic, no Simpson grid (:557-564)
                h.dist_type[i] == CLOUDY_DIST_LOGNORMAL) {      // :614-625 by msh_lognormal's own rule (kernels.hpp)
                h.finite[i] = 1;
                any_finite = true;
                continue;
            }
            const double x_lb = std::fmin(1e-5, 1e-5 * xt);
            const int n_bins = (int)std::floor(h.nbpl * std::log10(xt / x_lb));
            if (n_bins < 3) {
                delete p;
                return fail(CLOUDY_EINVAL, "n_bins must be at least 3");  // ParticleDistributions.jl:699
            }
            const double x_min = std::log(x_lb);
            const double dx = (std::log(xt) - std::log(x_lb)) / n_bins;
            h.finite[i] = 1;
            h.node_off[i] = (int)(nodes.size() / kNodeStride);
            h.n_bins[i] = n_bins;
            any_finite = true;
            for (int j = 1; j <= n_bins; ++j) {
                const double lx = x_min + (j - 1) * dx;  // logx, :566
                const double x = std::exp(lx);
                nodes.push_back(x);
                nodes.push_back(lx);
                nodes.push_back(xt - x);
                nodes.push_back(std::log(xt - x));
                nodes.push_back(simpson_weight(j, n_bins) * dx);
            }
        }
        h.mode = any_finite ? MODE_FIXED : MODE_ALLINF;
    } else {
        h.mode = (N > 1) ? MODE_MOVING : MODE_ALLINF;
        for (int i = 0; i < N - 1; ++i) {
            if (!(h.thr[i] >= 0.0 && h.thr[i] <= 1.0)) {
                delete p;
                return fail(CLOUDY_EINVAL, "MovingThreshold percentile %d outside [0, 1]", i);
            }
            if (h.dist_type[i] != CLOUDY_DIST_EXPONENTIAL && h.dist_type[i] != CLOUDY_DIST_GAMMA) {
                // compute_threshold has methods for Exponential and Gamma only (:747-761)
                // (round 6, found by the host sanitizer build: the message used to read h.dist_type[i] AFTER `delete p`)
                const int dti = h.dist_type[i];
                delete p;
                return fail(CLOUDY_EINVAL, "no method compute_threshold for dist_type[%d] = %d", i, dti);
            }
        }
        // start values of the per-parcel inversion gamma_inc_inv(k, p_i, 1 - p_i) (:760): ln x as a polynomial in ln k
        // over k in [0.03, kmax], one per Gamma mode; kept only if every fit is good to 1e-3 (two Halley steps then
        // reach machine precision); otherwise the kernels use the generic start value
        bool any_fit = false, all_good = true;
        const double k_lo = 0.03, k_hi = std::fmax(h.kmax, 1.0);
        for (int i = 0; i < N - 1 && k_hi > 2.0 * k_lo; ++i) {
            if (h.dist_type[i] != CLOUDY_DIST_GAMMA || !(h.thr[i] > 0.0 && h.thr[i] < 1.0)) continue;
            const double err = gamma_host::fit_inverse(h.thr[i], k_lo, k_hi, kInvTerms, h.inv_map[0], h.inv_map[1], h.inv_tab[i]);
            any_fit = true;
            all_good = all_good && err < 1e-3;
        }
        if (!any_fit || !all_good) h.inv_map[0] = h.inv_map[1] = 0.0;
        h.inv_klo = k_lo;
    }

    // sedimentation velocity, rescaled as the rainshaft caller does (rainshaft_helpers.jl:74-76)
    h.n_vel = d->n_vel;
    for (int v = 0; v < d->n_vel; ++v) {
        h.vel[v][0] = d->vel[v][0];
        h.vel[v][1] = d->vel[v][1];
        h.vel_n[v][0] = d->vel[v][0] * std::pow(d->norms[1], d->vel[v][1]);
        h.vel_n[v][1] = d->vel[v][1];
    }
    h.n_nodes = (int)(nodes.size() / kNodeStride);
    h.jit_sorted_bs = jit_sorted_block_size_resolve(h);
    *out = p;
    return CLOUDY_OK;
}

}  // namespace

extern "C" {

int cloudy_plan_create(const cloudy_plan_desc *d, cloudy_plan **out) {
    cloudy_plan *p = nullptr;
    std::vector<double> nodes;
    int rc0 = build_host_plan(d, &p, nodes);
    if (out) *out = nullptr;
    if (rc0 != CLOUDY_OK) return rc0;
    HostPlan &h = p->h;

    // device-side constants
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        delete p;
        return fail(CLOUDY_ENODEVICE, "no HIP device: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (d->device >= 0) {
        if (d->device >= ndev) {
            delete p;
            return fail(CLOUDY_EINVAL, "device %d out of range (%d devices)", d->device, ndev);
        }
        h.device = d->device;
    } else {
        (void)hipGetDevice(&h.device);
    }
    DeviceGuard guard(h.device);  // the caller's current device is restored on every return path
    if (guard.err != hipSuccess) {
        delete p;
        return fail_hip(guard.err, "hipSetDevice");
    }
    {
        const char *e = std::getenv("CLOUDY_HIP_PPL1");
        h.force_ppl1 = (e && e[0] == '1') ? 1 : 0;
    }
    if (h.n_nodes > 0) {
        e = hipMalloc((void **)&h.nodes_dev, nodes.size() * sizeof(double));
        if (e == hipSuccess)
            e = hipMemcpy(h.nodes_dev, nodes.data(), nodes.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            cloudy_plan_destroy(p);
            return fail_hip(e, "node table upload");
        }
    }
    if (!h.qtab.empty()) {
        e = hipMalloc((void **)&h.qtab_dev, h.qtab.size() * sizeof(double));
        if (e == hipSuccess)
            e = hipMemcpy(h.qtab_dev, h.qtab.data(), h.qtab.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            cloudy_plan_destroy(p);
            return fail_hip(e, "rule table upload");
        }
    }
    e = hipMalloc((void **)&h.partial_dev, sizeof(double) * kSumBlocks * (CLOUDY_MAX_MOMENTS + 64));
    if (e != hipSuccess) {
        cloudy_plan_destroy(p);
        return fail_hip(e, "workspace allocation");
    }
    if (!plan_beyond_aot(h)) {  // (the kernels compiled for the plan carry its constants as literals)
        LaunchReq prep{OP_PREPARE, IN_MOMENTS, 1, 0, 0, 0, nullptr, &h.kargs_dev, nullptr, nullptr};
        e = dispatch(h, prep);
        if (e != hipSuccess) {
            cloudy_plan_destroy(p);
            return fail_hip(e, "constant block upload");
        }
    }
    // plan-time specialisation (jit.hpp): 0 = when available, 1 = required, -1 = off
    const char *env = std::getenv("CLOUDY_HIP_JIT");
    const bool env_off = env && env[0] == '0';
    // every (threshold mode, plane type) combination of cloudy_coal_rhs has a specialised kernel
    if (d->specialize >= 0 && !(env_off && d->specialize == 0)) {
        p->jit_on = jit_get(h, p->jit, p->jit_log);
        if (!p->jit_on && (d->specialize > 0 || plan_beyond_aot(h))) {
            int rc = fail(CLOUDY_EUNSUPPORTED, "plan-time specialisation failed: %.400s", p->jit_log.c_str());
            cloudy_plan_destroy(p);
            return rc;
        }
    } else {
        p->jit_log = "switched off (desc.specialize < 0 or CLOUDY_HIP_JIT=0)";
        if (plan_beyond_aot(h)) {
            int rc = fail(CLOUDY_EUNSUPPORTED, "plans of more than %d modes or order > %d need plan-time compilation, which is %s",
                          CLOUDY_AOT_MAX_MODES, CLOUDY_AOT_MAX_P - 1, p->jit_log.c_str());
            cloudy_plan_destroy(p);
            return rc;
        }
    }
    *out = p;
    return CLOUDY_OK;
}

int cloudy_plan_specialized(const cloudy_plan *plan) {
    if (!plan) return fail(CLOUDY_EINVAL, "plan is NULL");
    return plan->jit_on ? 1 : 0;
}

const char *cloudy_plan_jit_log(const cloudy_plan *plan) { return plan ? plan->jit_log.c_str() : "plan is NULL"; }

int cloudy_jit_selfcheck(const cloudy_plan_desc *d, const char *arch) {
    cloudy_plan *p = nullptr;
    std::vector<double> nodes;
    int rc = build_host_plan(d, &p, nodes);
    if (rc != CLOUDY_OK) return rc;
    std::string log;
    std::vector<char> code;
    const std::string a = (arch && *arch) ? arch : "gfx950";
    const bool numerical = p->h.coal_style == CLOUDY_NUMERICAL_COAL;
    if (a == "source-only") {
        // (tests/test_host_sanitizers.py: the string assembly of EVERY translation-unit part of the plan, nothing compiled -- the
        // host code an N = 8 / P = 8 plan exercises, without the minutes of hiprtc its kernels take)
        size_t bytes = 0;
        for (int part : {0, 1, 2, 3, 4, 5, 6, 7, 8}) {
            if (numerical && (part == 2 || part == 3 || part == 5 || part == 6 || part == 8)) continue;   // no column kernels
            bytes += jit_source(p->h, part).size();
        }
        delete p;
        return bytes > 0 ? CLOUDY_OK : fail(CLOUDY_EUNSUPPORTED, "no source generated");
    }
    bool ok = jit_compile(jit_source(p->h, 0), a, numerical ? jit_quad_licm_off(p->h) : p->h.mode == MODE_ALLINF, code, log);
    if (ok && numerical) ok = jit_compile(jit_source(p->h, 1), a, true, code, log);  // fused SSPRK33 of the quadrature plan
    if (ok && !numerical && p->h.mode != MODE_ALLINF) ok = jit_compile(jit_source(p->h, 1), a, true, code, log);
    if (ok && !numerical) ok = jit_compile(jit_source(p->h, 2), a, false, code, log);  // rainshaft cell body
    if (ok && !numerical && p->h.n_vel > 0 && p->h.mode != MODE_MOVING)
    {
        const char *l = std::getenv("CLOUDY_HIP_JIT_RS_LICM");  // (the experiment switch of jit_get_rainshaft_integrator)
        // the fused column integrator + column RHS at every workgroup size whose LDS rows fit a workgroup (jit_rainshaft_fits: what
        // pick_rainshaft may choose); at least one must exist for columns of up to 256 cells unless the plan is beyond them all
        for (int part : {3, 5, 6})
            if (ok && jit_rainshaft_fits(p->h, part, 1)) ok = jit_compile(jit_source(p->h, part), a, !(l && l[0] == '1'), code, log);
    }
    if (ok && p->h.dtype != CLOUDY_F32_FAST) ok = jit_compile(jit_source(p->h, 4), a, true, code, log);  // cloudy_tsit5_steps
    if (ok && plan_beyond_aot(p->h)) ok = jit_compile(jit_source(p->h, 7), a, false, code, log);  // diagnostics
    delete p;
    if (!ok) return fail(CLOUDY_EUNSUPPORTED, "plan-time compilation failed: %.440s", log.c_str());
    return CLOUDY_OK;
}

void cloudy_plan_destroy(cloudy_plan *plan) {
    if (!plan) return;
    if (plan->h.nodes_dev) (void)hipFree(plan->h.nodes_dev);
    if (plan->h.partial_dev) (void)hipFree(plan->h.partial_dev);
    if (plan->h.kargs_dev) (void)hipFree(plan->h.kargs_dev);
    if (plan->h.qtab_dev) (void)hipFree(plan->h.qtab_dev);
    if (plan->hint_dev) (void)hipFree(plan->hint_dev);
    for (unsigned char *p : plan->hint_old) (void)hipFree(p);
    delete plan;
}

int cloudy_plan_desc_layout(const char **names, uint32_t *offsets, uint32_t *sizes, int cap) {
    struct Field {
        const char *name;
        size_t off, size;
    };
#define CLOUDY_FIELD(f) {#f, offsetof(cloudy_plan_desc, f), sizeof(((cloudy_plan_desc *)nullptr)->f)}
    static const Field fields[] = {
        CLOUDY_FIELD(struct_size), CLOUDY_FIELD(n_modes), CLOUDY_FIELD(dist_type), CLOUDY_FIELD(tensor_p),
        CLOUDY_FIELD(kernel_layout), CLOUDY_FIELD(kernel_is_normalized), CLOUDY_FIELD(kernel_c),
        CLOUDY_FIELD(dist_thresholds), CLOUDY_FIELD(threshold_style), CLOUDY_FIELD(norms), CLOUDY_FIELD(k_range),
        CLOUDY_FIELD(n_bins_per_log_unit), CLOUDY_FIELD(dtype), CLOUDY_FIELD(n_vel), CLOUDY_FIELD(vel),
        CLOUDY_FIELD(device), CLOUDY_FIELD(specialize), CLOUDY_FIELD(coal_style), CLOUDY_FIELD(kernel_func),
        CLOUDY_FIELD(kernel_func_is_normalized), CLOUDY_FIELD(quad_order), CLOUDY_FIELD(kernel_func_params),
        CLOUDY_FIELD(thresholds_are_normalized), CLOUDY_FIELD(quad_mode)};
#undef CLOUDY_FIELD
    const int nf = (int)(sizeof(fields) / sizeof(fields[0]));
    for (int i = 0; i < nf && i < cap; ++i) {
        if (names) names[i] = fields[i].name;
        if (offsets) offsets[i] = (uint32_t)fields[i].off;
        if (sizes) sizes[i] = (uint32_t)fields[i].size;
    }
    return nf;
}

int cloudy_quad_rule_host(int quad_order, double k_hi, double k, double *u, double *W) {
    if (quad_order < 2 || quad_order > CLOUDY_MAX_QUAD) return fail(CLOUDY_EUNSUPPORTED, "quad_order outside 2..%d", CLOUDY_MAX_QUAD);
    if (!u || !W) return fail(CLOUDY_EINVAL, "u / W is NULL");
    if (!(k_hi >= 1.0) || !(k > 0.0) || !(k <= k_hi)) return fail(CLOUDY_EINVAL, "need k_hi >= 1 and 0 < k <= k_hi");
    // one table per (order, k_hi) for the life of the process, as a plan would hold it
    static std::mutex mu;
    static std::map<std::pair<int, double>, std::pair<QArgs, std::vector<double>>> tables;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(quad_order, k_hi);
    auto it = tables.find(key);
    if (it == tables.end()) {
        QArgs q = {};
        std::vector<double> tab;
        std::string msg;
        if (!quad_host::build_table(quad_order, k_hi, q, tab, msg)) return fail(CLOUDY_EUNSUPPORTED, "%s", msg.c_str());
        it = tables.emplace(key, std::make_pair(q, std::move(tab))).first;
    }
    gamma_rule<0>(it->second.first, it->second.second.data(), k, u, W);
    return CLOUDY_OK;
}

int cloudy_plan_nmom(const cloudy_plan *plan) { return plan ? plan->h.nmom : fail(CLOUDY_EINVAL, "plan is NULL"); }
int cloudy_plan_device(const cloudy_plan *plan) { return plan ? plan->h.device : fail(CLOUDY_EINVAL, "plan is NULL"); }
int cloudy_plan_nparams(const cloudy_plan *plan) { return plan ? 3 * plan->h.N : fail(CLOUDY_EINVAL, "plan is NULL"); }

int cloudy_plan_get(const cloudy_plan *plan, int32_t *N_mom_max, int32_t *N_2d_ints, double *thresholds,
                    double *kernel_c_normalized, double *mom_norms) {
    if (!plan) return fail(CLOUDY_EINVAL, "plan is NULL");
    const HostPlan &h = plan->h;
    if (N_mom_max) *N_mom_max = h.n_mom_max;
    for (int i = 0; i < h.N; ++i) {
        if (N_2d_ints) N_2d_ints[i] = h.n_2d[i];
        if (thresholds) thresholds[i] = h.thr[i];
    }
    if (kernel_c_normalized)
        for (int j = 0; j < h.N; ++j)
            for (int k = 0; k < h.N; ++k)
                for (int a = 0; a < h.P; ++a)
                    for (int b = 0; b < h.P; ++b)
                        kernel_c_normalized[((size_t)(j * h.N + k) * h.P + a) * h.P + b] = h.c[j][k][a][b];
    if (mom_norms)
        for (int i = 0; i < h.N; ++i)
            for (int m = 0; m < h.np[i]; ++m) mom_norms[h.off[i] + m] = h.mom_norm[i][m];
    return CLOUDY_OK;
}

int cloudy_coal_rhs(const cloudy_plan *plan, size_t n, size_t ld, const void *mom_dev, void *dmom_dev, void *stream) {
    int rc = check_batch(plan, n, ld, mom_dev, dmom_dev);
    if (rc) return rc;
    LaunchReq r{OP_COAL, IN_MOMENTS, 1, 0, n, ld, mom_dev, dmom_dev, nullptr,
                (hipStream_t)stream};
    return run(plan, r);
}

int cloudy_get_coal_ints(const cloudy_plan *plan, size_t n, size_t ld, const void *params_dev, void *out_dev,
                         void *stream) {
    int rc = check_batch(plan, n, ld, params_dev, out_dev);
    if (rc) return rc;
    LaunchReq r{OP_COAL, IN_PARAMS, 0, 0, n, ld, params_dev, out_dev, nullptr,
                (hipStream_t)stream};
    return run(plan, r);
}

int cloudy_ssprk33_steps(const cloudy_plan *plan, size_t n, size_t ld, const void *u_in_dev, void *u_out_dev, double dt,
                         int n_steps, void *stream) {
    int rc = check_batch(plan, n, ld, u_in_dev, u_out_dev);
    if (rc) return rc;
    if (n_steps < 0 || !(dt == dt)) return fail(CLOUDY_EINVAL, "n_steps must be >= 0 and dt not NaN");
    LaunchReq r{OP_SSPRK33, IN_MOMENTS, 1, 0, n, ld, u_in_dev, u_out_dev, nullptr,
                (hipStream_t)stream};
    r.dt = dt;
    r.n_steps = n_steps;
    return run(plan, r);
}

int cloudy_tsit5_steps(const cloudy_plan *plan, size_t n, size_t ld, const void *u_in_dev, void *u_out_dev, double dt,
                       int n_steps, void *stream) {
    int rc = check_batch(plan, n, ld, u_in_dev, u_out_dev);
    if (rc) return rc;
    if (n_steps < 0 || !(dt == dt)) return fail(CLOUDY_EINVAL, "n_steps must be >= 0 and dt not NaN");
    if (plan->h.dtype == CLOUDY_F32_FAST)
        return fail(CLOUDY_EUNSUPPORTED, "cloudy_tsit5_steps serves CLOUDY_F64 and CLOUDY_F32 plans (the fused integrators keep "
                                         "the state in fp64 registers; CLOUDY_F32_FAST is the single-pass operator's mode)");
    LaunchReq r{OP_TSIT5, IN_MOMENTS, 1, 0, n, ld, u_in_dev, u_out_dev, nullptr, (hipStream_t)stream};
    r.dt = dt;
    r.n_steps = n_steps;
    return run(plan, r);
}

int cloudy_update_dist_from_moments(const cloudy_plan *plan, size_t n, size_t ld, const void *mom_dev,
                                    void *params_dev, void *stream) {
    int rc = check_batch(plan, n, ld, mom_dev, params_dev);
    if (rc) return rc;
    if (plan->h.dtype != CLOUDY_F64)
        return fail(CLOUDY_EUNSUPPORTED, "cloudy_update_dist_from_moments reads fp64 planes: use a CLOUDY_F64 plan");
    LaunchReq r{OP_UPDATE_DIST, IN_MOMENTS, 0, 0, n, ld, mom_dev, params_dev, nullptr,
                (hipStream_t)stream};
    return run(plan, r);
}

int cloudy_closure_stats(const cloudy_plan *plan, size_t n, size_t ld, const void *mom_dev, uint64_t *counts_host, void *stream) {
    if (!plan) return fail(CLOUDY_EINVAL, "plan is NULL");
    if (!counts_host) return fail(CLOUDY_EINVAL, "counts_host is NULL");
    if (ld < n) return fail(CLOUDY_EINVAL, "ld (%zu) must be >= n_parcels (%zu)", ld, n);
    if (n > 0 && !mom_dev) return fail(CLOUDY_EINVAL, "device buffer is NULL");
    const HostPlan &h = plan->h;
    const int nc = 4 * h.N;
    for (int q = 0; q < nc; ++q) counts_host[q] = 0;
    if (n == 0) return CLOUDY_OK;
    DeviceGuard guard(h.device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "selecting the plan's device");
    ClosureStatsArgs a;
    a.N = h.N;
    for (int m = 0; m < CLOUDY_MAX_MODES; ++m) {
        a.dist_type[m] = m < h.N ? h.dist_type[m] : 0;
        a.np[m] = m < h.N ? h.np[m] : 0;
        a.off[m] = m < h.N ? h.off[m] : 0;
        for (int q = 0; q < 3; ++q) {
            a.norm[3 * m + q] = m < h.N ? h.mom_norm[m][q] : 1.0;
            a.inv_norm[3 * m + q] = 1.0 / a.norm[3 * m + q];
        }
    }
    a.kmin = h.kmin;
    a.kmax = h.kmax;
    hipStream_t st = (hipStream_t)stream;
    size_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > (size_t)(4 * kSumBlocks)) blocks = 4 * kSumBlocks;
    // one row of 4N counters per workgroup, every one of them stored by its workgroup (nothing to zero, no atomics)
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "counter width");
    std::vector<uint64_t> rows(blocks * (size_t)nc);
    unsigned long long *dev = nullptr;
    HIP_TRY(hipMallocAsync((void **)&dev, sizeof(unsigned long long) * rows.size(), st));
    if (h.dtype != CLOUDY_F64)
        hipLaunchKernelGGL(closure_stats_kernel<float>, dim3((unsigned)blocks), dim3(kBlock), 0, st, a, n, ld, (const float *)mom_dev, dev);
    else
        hipLaunchKernelGGL(closure_stats_kernel<double>, dim3((unsigned)blocks), dim3(kBlock), 0, st, a, n, ld, (const double *)mom_dev, dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(rows.data(), dev, sizeof(uint64_t) * rows.size(), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFreeAsync(dev, st);
    if (e == hipSuccess)
        for (size_t b = 0; b < blocks; ++b)
            for (int q = 0; q < nc; ++q) counts_host[q] += rows[b * (size_t)nc + q];
    if (e != hipSuccess) return fail_hip(e, "cloudy_closure_stats");
    return CLOUDY_OK;
}

int cloudy_finite_2d_integrals(const cloudy_plan *plan, size_t n, size_t ld, const void *params_dev, void *F_dev,
                               void *stream) {
    int rc = check_batch(plan, n, ld, params_dev, F_dev);
    if (rc) return rc;
    LaunchReq r{OP_FINITE_2D, IN_PARAMS, 0, 0, n, ld, params_dev, F_dev, nullptr,
                (hipStream_t)stream};
    return run(plan, r);
}

int cloudy_compute_thresholds(const cloudy_plan *plan, size_t n, size_t ld, const void *params_dev,
                              void *thresholds_dev, void *stream) {
    int rc = check_batch(plan, n, ld, params_dev, thresholds_dev);
    if (rc) return rc;
    LaunchReq r{OP_FINITE_2D, IN_PARAMS, 0, 0, n, ld, params_dev, nullptr, thresholds_dev,
                (hipStream_t)stream};
    return run(plan, r);
}

int cloudy_sedimentation_flux(const cloudy_plan *plan, size_t n, size_t ld, const void *mom_dev, void *flux_dev,
                              void *stream) {
    int rc = check_batch(plan, n, ld, mom_dev, flux_dev);
    if (rc) return rc;
    if (plan->h.n_vel < 1) return fail(CLOUDY_EINVAL, "plan has no terminal-velocity coefficients (n_vel = 0)");
    LaunchReq r{OP_SEDI, IN_MOMENTS, 1, 0, n, ld, mom_dev, flux_dev, nullptr,
                (hipStream_t)stream};
    return run(plan, r);
}

int cloudy_cond_evap(const cloudy_plan *plan, size_t n, size_t ld, const void *mom_dev, const double *s_dev, double s,
                     double xi, void *dmom_dev, void *stream) {
    int rc = check_batch(plan, n, ld, mom_dev, dmom_dev);
    if (rc) return rc;
    LaunchReq r{OP_COND, IN_MOMENTS, 1, 0, n, ld, mom_dev, dmom_dev, nullptr, (hipStream_t)stream};
    // rhs_condensation!: xi_normalized = p.xi / norms[2]^(2/3) (box_model_helpers.jl:65); rho_l = 1000 (Condensation.jl:26)
    const double xi_n = xi / std::pow(plan->h.norms[1], 2.0 / 3.0);
    r.coef = 3 * xi_n * std::pow(4 * M_PI / 3, 2.0 / 3.0) / std::pow(1000.0, 1.0 / 3.0);
    r.s_scalar = s;
    r.s_dev = s_dev;
    return run(plan, r);
}

int cloudy_standard_N_q(const cloudy_plan *plan, size_t n, size_t ld, const void *mom_dev, double size_cutoff,
                        void *nq_dev, void *stream) {
    int rc = check_batch(plan, n, ld, mom_dev, nq_dev);
    if (rc) return rc;
    if (!(size_cutoff > 0)) return fail(CLOUDY_EINVAL, "size_cutoff must be positive");
    LaunchReq r{OP_NQ, IN_MOMENTS, 1, 0, n, ld, mom_dev, nq_dev, nullptr, (hipStream_t)stream};
    r.s_scalar = size_cutoff;
    return run(plan, r);
}

int cloudy_rainshaft_sources(const cloudy_plan *plan, size_t n, size_t ld, const void *mom_dev, void *coal_source_dev,
                             void *sedi_flux_dev, void *stream) {
    int rc = check_batch(plan, n, ld, mom_dev, coal_source_dev);
    if (rc) return rc;
    if (n > 0 && !sedi_flux_dev) return fail(CLOUDY_EINVAL, "device buffer is NULL");
    if (plan->h.n_vel < 1) return fail(CLOUDY_EINVAL, "plan has no terminal-velocity coefficients (n_vel = 0)");
    if (plan->h.threshold_style != CLOUDY_FIXED_THRESHOLD)
        return fail(CLOUDY_EINVAL, "make_rainshaft_rhs uses FixedThreshold (rainshaft_helpers.jl:70)");
    LaunchReq r1{OP_COAL, IN_MOMENTS, 1, 1, n, ld, mom_dev, coal_source_dev, nullptr,
                 (hipStream_t)stream};
    // The cell-body kernel compiled for the plan writes the sedimentation flux as well (one launch, one read of the
    // moments, one closure inversion); the ahead-of-time path is two launches.
    if (plan->jit_on && n > 0) {
        std::call_once(plan->rs_once, [&] { (void)jit_get_rainshaft(plan->h, plan->rs_coal, plan->rs_log); });
        if (plan->rs_coal != nullptr) {
            r1.out2 = sedi_flux_dev;
            return run(plan, r1);
        }
    }
    rc = run(plan, r1);
    if (rc) return rc;
    LaunchReq r2{OP_SEDI, IN_MOMENTS, 1, 1, n, ld, mom_dev, sedi_flux_dev, nullptr,
                 (hipStream_t)stream};
    return run(plan, r2);
}

int cloudy_rainshaft_rhs(const cloudy_plan *plan, size_t nz, size_t n_columns, size_t ld, const void *mom_dev, double dz,
                         void *flux_work_dev, void *rhs_dev, void *stream) {
    const size_t n = nz * n_columns;
    int rc = check_batch(plan, n, ld, mom_dev, rhs_dev);
    if (rc) return rc;
    if (nz < 1 || !(dz > 0)) return fail(CLOUDY_EINVAL, "nz must be >= 1 and dz positive");
    // Round 5: one launch when the plan has the column kernels compiled for it and a workgroup holds a column (nz <= 1024):
    // the RHS_ONLY instance of the column integrator's body -- sources, flux, exchange through LDS, divergence, sum -- writes
    // rhs and the cell fluxes; fp64 planes: the same bits as the two launches below (test).  CLOUDY_HIP_RS_FUSED_RHS=0: off.
    // (the argument checks of the two-launch path come first -- whether a call is valid must not depend on whether the
    // kernels compiled for the plan are on)
    if (!flux_work_dev && n > 0) return fail(CLOUDY_EINVAL, "device buffer is NULL");
    if (plan->h.n_vel < 1) return fail(CLOUDY_EINVAL, "plan has no terminal-velocity coefficients (n_vel = 0)");
    if (plan->h.threshold_style != CLOUDY_FIXED_THRESHOLD)
        return fail(CLOUDY_EINVAL, "make_rainshaft_rhs uses FixedThreshold (rainshaft_helpers.jl:70)");
    if (n > 0 && nz <= 1024 && plan->jit_on && plan->h.coal_style != CLOUDY_NUMERICAL_COAL && plan->h.mode != MODE_MOVING) {
        const char *fe = std::getenv("CLOUDY_HIP_RS_FUSED_RHS");
        if (!(fe && fe[0] == '0')) {
            DeviceGuard guard(plan->h.device);
            if (guard.err != hipSuccess) return fail_hip(guard.err, "selecting the plan's device");
            const RsPick pk = pick_rainshaft(plan, nz, n_columns, true);
            if (pk.fn != nullptr) {
                const double *nodes = plan->h.nodes_dev;
                int nzi = (int)nz;
                const unsigned bs = (unsigned)pk.bs;
                const size_t cpb = bs / nz;
                void *args[] = {&nodes, &nzi, &n_columns, &ld, &mom_dev, &rhs_dev, &flux_work_dev, &dz};
                hipError_t e = hipModuleLaunchKernel(pk.fn, (unsigned)((n_columns + cpb - 1) / cpb), 1, 1, bs, 1, 1, 0, (hipStream_t)stream,
                                                     args, nullptr);
                if (e != hipSuccess) return fail_hip(e, "column right-hand side launch");
                return CLOUDY_OK;
            }
        }
    }
    rc = cloudy_rainshaft_sources(plan, n, ld, mom_dev, rhs_dev, flux_work_dev, stream);
    if (rc || n == 0) return rc;
    DeviceGuard guard(plan->h.device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "selecting the plan's device");
    const unsigned g = (unsigned)((n + kBlock - 1) / kBlock);
    if (plan->h.dtype != CLOUDY_F64)
        hipLaunchKernelGGL(rainshaft_divergence_kernel<float>, dim3(g), dim3(kBlock), 0, (hipStream_t)stream, n, ld,
                           plan->h.nmom, nz, dz, (const float *)flux_work_dev, (float *)rhs_dev);
    else
        hipLaunchKernelGGL(rainshaft_divergence_kernel<double>, dim3(g), dim3(kBlock), 0, (hipStream_t)stream, n, ld,
                           plan->h.nmom, nz, dz, (const double *)flux_work_dev, (double *)rhs_dev);
    HIP_TRY(hipGetLastError());
    return CLOUDY_OK;
}

int cloudy_rainshaft_ssprk33_steps(const cloudy_plan *plan, size_t nz, size_t n_columns, size_t ld, const void *u_in_dev,
                                   void *u_out_dev, double dz, double dt, int n_steps, void *stream) {
    const size_t n = nz * n_columns;
    int rc = check_batch(plan, n, ld, u_in_dev, u_out_dev);
    if (rc) return rc;
    if (nz < 1 || !(dz > 0)) return fail(CLOUDY_EINVAL, "nz must be >= 1 and dz positive");
    if (n_steps < 0 || !(dt == dt)) return fail(CLOUDY_EINVAL, "n_steps must be >= 0 and dt not NaN");
    if (plan->h.n_vel < 1) return fail(CLOUDY_EINVAL, "plan has no terminal-velocity coefficients (n_vel = 0)");
    if (plan->h.threshold_style != CLOUDY_FIXED_THRESHOLD)
        return fail(CLOUDY_EINVAL, "make_rainshaft_rhs uses FixedThreshold (rainshaft_helpers.jl:70)");
    if (n == 0) return CLOUDY_OK;
    // no fused kernel for this column height: taller than a workgroup, or (round 6) a plan whose LDS rows leave no
    // workgroup size that holds a column (5+ modes with a threshold beyond 256 cells, ...) and that the ahead-of-time
    // integrator (<= 256 cells, <= CLOUDY_AOT_MAX_MODES modes) does not serve either
    bool staged = nz > 1024;
    if (!staged && plan->h.coal_style != CLOUDY_NUMERICAL_COAL) {
        const bool aot = nz <= (size_t)kRainshaftBlock && plan->h.N <= CLOUDY_AOT_MAX_MODES && plan->h.P <= CLOUDY_AOT_MAX_P;
        if (!aot) {
            DeviceGuard guard(plan->h.device);
            if (guard.err != hipSuccess) return fail_hip(guard.err, "selecting the plan's device");
            staged = plan->h.mode == MODE_MOVING || pick_rainshaft(plan, nz, n_columns, false).fn == nullptr;
        }
    }
    if (staged) {
        // Round 5 (VERDICT r4 missing #4): the reference's cell loop is unbounded in nz (rainshaft_helpers.jl:55-78), the fused
        // kernel keeps a column inside one workgroup (<= 1024 cells).  Taller columns are stepped stage by stage on the stream:
        // cloudy_rainshaft_rhs (cell sources + flux divergence: two launches) and one update launch per stage, with three
        // stream-ordered scratch arrays of the state's size (allocated and released on the caller's stream per call).
        //
        // Experiment switch CLOUDY_HIP_GRAPH=1 (round 5, measured SLOWER on ROCm 7.2 and therefore off): from the second step on
        // ONE step -- nine launches -- is captured into a hipGraph and replayed.  The first step runs eagerly (it builds whatever
        // the plan still has to compile; nothing may be compiled or loaded during a capture), the second is recorded on a stream
        // of this call (the caller's stream may be the legacy NULL stream, which cannot capture) ordered after the caller's
        // stream by an event and before it by another.  Same bits (test_tall_columns_replay_a_captured_step); per step of
        // 3 x 1500 / 64 x 1500 / 256 x 4000 cells: eager 0.143 / 0.128 / 0.52 ms, replayed 0.158 / 0.162 / 0.84 ms, plus 3-5 ms
        // of capture and instantiation per call (profiles/r05_tall_columns_graph_experiment.txt):
        // the eager launches already queue back to back on the stream, and the graph's kernel nodes do not run closer together.
        DeviceGuard guard(plan->h.device);
        if (guard.err != hipSuccess) return fail_hip(guard.err, "selecting the plan's device");
        hipStream_t user = (hipStream_t)stream, st = user;
        const size_t esz = plan->h.dtype != CLOUDY_F64 ? sizeof(float) : sizeof(double);
        const size_t bytes = (size_t)plan->h.nmom * ld * esz;
        bool graph_on = false;
        if (const char *e = std::getenv("CLOUDY_HIP_GRAPH")) graph_on = n_steps >= 4 && e[0] == '1';
        if (graph_on) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(user, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) graph_on = false;
            (void)hipGetLastError();
        }
        hipStream_t own = nullptr;
        hipEvent_t ev_in = nullptr, ev_out = nullptr;
        if (graph_on) {
            graph_on = hipStreamCreateWithFlags(&own, hipStreamNonBlocking) == hipSuccess &&
                       hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) == hipSuccess &&
                       hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) == hipSuccess &&
                       hipEventRecord(ev_in, user) == hipSuccess && hipStreamWaitEvent(own, ev_in, 0) == hipSuccess;
            if (graph_on) st = own;
            else (void)hipGetLastError();
        }
        const auto release = [&] {
            if (ev_in) (void)hipEventDestroy(ev_in);
            if (ev_out) (void)hipEventDestroy(ev_out);
            if (own) (void)hipStreamDestroy(own);   // (returns at once; the stream goes when its work is done)
        };
        hipError_t e = hipSuccess;
        if (u_out_dev != u_in_dev) e = hipMemcpyAsync(u_out_dev, u_in_dev, bytes, hipMemcpyDeviceToDevice, st);
        char *ws = nullptr;
        if (e == hipSuccess) e = hipMallocAsync((void **)&ws, 3 * bytes, st);
        if (e != hipSuccess) {
            release();
            return fail_hip(e, "scratch of the stage-by-stage column integrator");
        }
        void *up = ws, *f = ws + bytes, *flux = ws + 2 * bytes;
        const unsigned g = (unsigned)((n + kBlock - 1) / kBlock);
        const int planes = plan->h.nmom;
        const auto stage = [&](int sidx) -> hipError_t {
            if (plan->h.dtype != CLOUDY_F64)
                hipLaunchKernelGGL(rainshaft_stage_kernel<float>, dim3(g), dim3(kBlock), 0, st, sidx, n, ld, planes, (float *)up,
                                   (float *)u_out_dev, (const float *)f, dt);
            else
                hipLaunchKernelGGL(rainshaft_stage_kernel<double>, dim3(g), dim3(kBlock), 0, st, sidx, n, ld, planes, (double *)up,
                                   (double *)u_out_dev, (const double *)f, dt);
            return hipGetLastError();
        };
        rc = CLOUDY_OK;
        const auto one_step = [&] {
            for (int sidx = 0; sidx < 3 && rc == CLOUDY_OK && e == hipSuccess; ++sidx) {
                rc = cloudy_rainshaft_rhs(plan, nz, n_columns, ld, u_out_dev, dz, flux, f, st);
                if (rc == CLOUDY_OK) e = stage(sidx);
            }
        };
        int step = 0;
        if (n_steps > 0) {
            one_step();
            ++step;
        }
        if (graph_on && rc == CLOUDY_OK && e == hipSuccess) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                one_step();   // recorded, not run
                const hipError_t ec = hipStreamEndCapture(st, &graph);
                const bool recorded = rc == CLOUDY_OK && e == hipSuccess && ec == hipSuccess && graph != nullptr;
                rc = CLOUDY_OK;   // (a failure while recording leaves nothing half done: the eager loop below takes over)
                e = hipSuccess;
                if (recorded && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    for (; step < n_steps && e == hipSuccess; ++step) e = hipGraphLaunch(exec, st);
                    (void)hipGraphExecDestroy(exec);
                }
                if (graph) (void)hipGraphDestroy(graph);
            }
            (void)hipGetLastError();
        }
        for (; step < n_steps && rc == CLOUDY_OK && e == hipSuccess; ++step) one_step();
        if (rc == CLOUDY_OK && e == hipSuccess && n_steps > 0) e = stage(3);
        (void)hipFreeAsync(ws, st);
        if (own != nullptr) {   // the caller's stream goes on after this call's work
            hipError_t eo = hipEventRecord(ev_out, own);
            if (eo == hipSuccess) eo = hipStreamWaitEvent(user, ev_out, 0);
            if (eo != hipSuccess && e == hipSuccess) e = eo;
        }
        release();
        if (rc != CLOUDY_OK) return rc;
        if (e != hipSuccess) return fail_hip(e, "stage-by-stage column integrator");
        return CLOUDY_OK;
    }
    LaunchReq r{OP_RAINSHAFT_SSPRK33, IN_MOMENTS, 1, 1, n, ld, u_in_dev, u_out_dev, nullptr, (hipStream_t)stream};
    r.dt = dt;
    r.n_steps = n_steps;
    r.nz = nz;
    r.dz = dz;
    return run(plan, r);
}

size_t cloudy_moment_sums_workspace_bytes(int planes) {
    return planes < 1 ? 0 : sizeof(double) * (size_t)kSumBlocks * (size_t)planes;
}

int cloudy_moment_sums_ws(const cloudy_plan *plan, size_t n, size_t ld, int planes, const void *arr_dev, double *sums_dev,
                          void *workspace_dev, size_t workspace_bytes, void *stream) {
    if (!plan) return fail(CLOUDY_EINVAL, "plan is NULL");
    if (planes < 1 || planes > CLOUDY_MAX_MOMENTS + 64) return fail(CLOUDY_EINVAL, "planes out of range");
    if (ld < n) return fail(CLOUDY_EINVAL, "ld must be >= n_parcels");
    if (!sums_dev || (n > 0 && !arr_dev)) return fail(CLOUDY_EINVAL, "device buffer is NULL");
    if (!workspace_dev || workspace_bytes < cloudy_moment_sums_workspace_bytes(planes))
        return fail(CLOUDY_EINVAL, "workspace must hold cloudy_moment_sums_workspace_bytes(planes) = %zu bytes",
                    cloudy_moment_sums_workspace_bytes(planes));
    DeviceGuard guard(plan->h.device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "selecting the plan's device");
    size_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > (size_t)kSumBlocks) blocks = kSumBlocks;
    if (blocks < 1) blocks = 1;
    double *partial = static_cast<double *>(workspace_dev);
    if (plan->h.dtype != CLOUDY_F64)
        hipLaunchKernelGGL(plane_partial_sums_kernel<float>, dim3((unsigned)blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, n, ld, planes, (const float *)arr_dev, partial);
    else
        hipLaunchKernelGGL(plane_partial_sums_kernel<double>, dim3((unsigned)blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, n, ld, planes, (const double *)arr_dev, partial);
    hipLaunchKernelGGL(plane_final_sums_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, (int)blocks, planes,
                       (const double *)partial, sums_dev);
    HIP_TRY(hipGetLastError());
    return CLOUDY_OK;
}

int cloudy_moment_sums(const cloudy_plan *plan, size_t n, size_t ld, int planes, const void *arr_dev, double *sums_dev,
                       void *stream) {
    if (!plan) return fail(CLOUDY_EINVAL, "plan is NULL");
    // the plan's own workspace: one reduction at a time per plan (header); cloudy_moment_sums_ws is the re-entrant form
    return cloudy_moment_sums_ws(plan, n, ld, planes, arr_dev, sums_dev, plan->h.partial_dev,
                                 sizeof(double) * kSumBlocks * (CLOUDY_MAX_MOMENTS + 64), stream);
}

int cloudy_coal_rhs_host(const cloudy_plan *plan, size_t n, size_t ld, const void *mom_host, void *dmom_host) {
    int rc = check_batch(plan, n, ld, mom_host, dmom_host);
    if (rc) return rc;
    if (n == 0) return CLOUDY_OK;
    const size_t bytes = (size_t)plan->h.nmom * ld * (plan->h.dtype != CLOUDY_F64 ? sizeof(float) : sizeof(double));
    DeviceGuard guard(plan->h.device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "selecting the plan's device");
    char *buf = nullptr;
    HIP_TRY(hipMalloc((void **)&buf, 2 * bytes));
    hipError_t e = hipMemcpy(buf, mom_host, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = cloudy_coal_rhs(plan, n, ld, buf, buf + bytes, nullptr);
        if (rc == CLOUDY_OK) e = hipMemcpy(dmom_host, buf + bytes, bytes, hipMemcpyDeviceToHost);
    }
    (void)hipFree(buf);
    if (e != hipSuccess) return fail_hip(e, "host staging copy");
    return rc;
}

int cloudy_time_coal_rhs(const cloudy_plan *plan, size_t n, size_t ld, const void *mom_dev, void *dmom_dev,
                         void *stream, int iters, float *ms_per_launch) {
    if (iters < 1 || !ms_per_launch) return fail(CLOUDY_EINVAL, "iters must be >= 1 and ms_per_launch non-NULL");
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    int rc = CLOUDY_OK;
    hipError_t he = hipEventRecord(e0, (hipStream_t)stream);
    for (int i = 0; i < iters && rc == CLOUDY_OK && he == hipSuccess; ++i)
        rc = cloudy_coal_rhs(plan, n, ld, mom_dev, dmom_dev, stream);
    if (he == hipSuccess) he = hipEventRecord(e1, (hipStream_t)stream);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != CLOUDY_OK) return rc;
    if (he != hipSuccess) return fail_hip(he, "event timing");
    *ms_per_launch = ms / (float)iters;
    return CLOUDY_OK;
}

int cloudy_timer_begin(void *stream, void **timer_out) {
    if (!timer_out) return fail(CLOUDY_EINVAL, "timer_out is NULL");
    hipEvent_t *ev = new (std::nothrow) hipEvent_t[2];
    if (!ev) return fail(CLOUDY_ENOMEM, "out of host memory");
    hipError_t e = hipEventCreate(&ev[0]);
    if (e == hipSuccess) e = hipEventCreate(&ev[1]);
    if (e == hipSuccess) e = hipEventRecord(ev[0], (hipStream_t)stream);
    if (e != hipSuccess) {
        delete[] ev;
        return fail_hip(e, "cloudy_timer_begin");
    }
    *timer_out = ev;
    return CLOUDY_OK;
}

int cloudy_timer_end(void *timer, void *stream, float *ms_total) {
    if (!timer || !ms_total) return fail(CLOUDY_EINVAL, "timer / ms_total is NULL");
    hipEvent_t *ev = static_cast<hipEvent_t *>(timer);
    hipError_t e = hipEventRecord(ev[1], (hipStream_t)stream);
    if (e == hipSuccess) e = hipEventSynchronize(ev[1]);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev[0], ev[1]);
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    delete[] ev;
    if (e != hipSuccess) return fail_hip(e, "cloudy_timer_end");
    *ms_total = ms;
    return CLOUDY_OK;
}

int cloudy_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int cloudy_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return CLOUDY_OK;
}
int cloudy_device_pci_bus_id(int device, char *buf, int len) {
    if (!buf || len < 13) return fail(CLOUDY_EINVAL, "buf is NULL or shorter than 13 bytes (\"0000:00:00.0\")");
    buf[0] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(CLOUDY_ENODEVICE, "no HIP device");
    if (device < 0 || device >= n) return fail(CLOUDY_EINVAL, "device ordinal out of range");
    HIP_TRY(hipDeviceGetPCIBusId(buf, len, device));
    return CLOUDY_OK;
}
int cloudy_malloc(void **dev_ptr, size_t bytes) {
    if (!dev_ptr) return fail(CLOUDY_EINVAL, "dev_ptr is NULL");
    HIP_TRY(hipMalloc(dev_ptr, bytes));
    return CLOUDY_OK;
}
int cloudy_free(void *dev_ptr) {
    HIP_TRY(hipFree(dev_ptr));
    return CLOUDY_OK;
}
int cloudy_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return CLOUDY_OK;
}
int cloudy_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return CLOUDY_OK;
}
int cloudy_memset(void *dst, int value, size_t bytes, void *stream) {
    HIP_TRY(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
    return CLOUDY_OK;
}
int cloudy_stream_synchronize(void *stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return CLOUDY_OK;
}

}  // extern "C"
