// instantiation unit: the NumericalCoalStyle (fixed Gauss rule) kernels of the N = 2 mode family
#include "launch_quad_impl.hpp"
namespace cloudy {
hipError_t launch_quad_n2(const HostPlan &h, const LaunchReq &r) { return launch_quad<2>(h, r); }
}  // namespace cloudy
