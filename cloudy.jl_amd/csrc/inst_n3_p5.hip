// instantiation unit: every kernel of the N = 3 modes, P = 5 (tensor order 4) family
#include "launch_impl.hpp"
namespace cloudy {
hipError_t launch_n3_p5(const HostPlan &h, const LaunchReq &r) { return launch_np<3, 5>(h, r); }
}  // namespace cloudy
