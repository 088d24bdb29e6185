// instantiation unit: every kernel of the N = 1 modes, P = 1 (tensor order 0) family
#include "launch_impl.hpp"
namespace cloudy {
hipError_t launch_n1_p1(const HostPlan &h, const LaunchReq &r) { return launch_np<1, 1>(h, r); }
}  // namespace cloudy
