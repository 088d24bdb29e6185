// instantiation unit: every kernel of the N = 1 modes, P = 2 (tensor order 1) family
#include "launch_impl.hpp"
namespace cloudy {
hipError_t launch_n1_p2(const HostPlan &h, const LaunchReq &r) { return launch_np<1, 2>(h, r); }
}  // namespace cloudy
