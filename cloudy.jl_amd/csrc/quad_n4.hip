// instantiation unit: the NumericalCoalStyle (fixed Gauss rule) kernels of the N = 4 mode family
#include "launch_quad_impl.hpp"
namespace cloudy {
hipError_t launch_quad_n4(const HostPlan &h, const LaunchReq &r) { return launch_quad<4>(h, r); }
}  // namespace cloudy
