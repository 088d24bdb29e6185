// instantiation unit: every kernel of the N = 4 modes, P = 1 (tensor order 0) family
#include "launch_impl.hpp"
namespace cloudy {
hipError_t launch_n4_p1(const HostPlan &h, const LaunchReq &r) { return launch_np<4, 1>(h, r); }
}  // namespace cloudy
