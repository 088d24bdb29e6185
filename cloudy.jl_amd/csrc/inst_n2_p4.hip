// instantiation unit: every kernel of the N = 2 modes, P = 4 (tensor order 3) family
#include "launch_impl.hpp"
namespace cloudy {
hipError_t launch_n2_p4(const HostPlan &h, const LaunchReq &r) { return launch_np<2, 4>(h, r); }
}  // namespace cloudy
