// instantiation unit: every kernel of the N = 3 modes, P = 1 (tensor order 0) family
#include "launch_impl.hpp"
namespace cloudy {
hipError_t launch_n3_p1(const HostPlan &h, const LaunchReq &r) { return launch_np<3, 1>(h, r); }
}  // namespace cloudy
