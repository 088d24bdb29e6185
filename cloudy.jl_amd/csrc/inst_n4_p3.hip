// instantiation unit: every kernel of the N = 4 modes, P = 3 (tensor order 2) family
#include "launch_impl.hpp"
namespace cloudy {
hipError_t launch_n4_p3(const HostPlan &h, const LaunchReq &r) { return launch_np<4, 3>(h, r); }
}  // namespace cloudy
