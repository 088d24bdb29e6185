// instantiation unit: every kernel of the N = 2 modes, P = 2 (tensor order 1) family
#include "launch_impl.hpp"
namespace cloudy {
hipError_t launch_n2_p2(const HostPlan &h, const LaunchReq &r) { return launch_np<2, 2>(h, r); }
}  // namespace cloudy
