// instantiation unit: the NumericalCoalStyle (fixed Gauss rule) kernels of the N = 3 mode family
#include "launch_quad_impl.hpp"
namespace cloudy {
hipError_t launch_quad_n3(const HostPlan &h, const LaunchReq &r) { return launch_quad<3>(h, r); }
}  // namespace cloudy
