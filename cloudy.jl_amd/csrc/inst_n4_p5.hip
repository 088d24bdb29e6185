// instantiation unit: every kernel of the N = 4 modes, P = 5 (tensor order 4) family
#include "launch_impl.hpp"
namespace cloudy {
hipError_t launch_n4_p5(const HostPlan &h, const LaunchReq &r) { return launch_np<4, 5>(h, r); }
}  // namespace cloudy
