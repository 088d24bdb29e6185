// instantiation unit: every kernel of the N = 4 mode family (P = 1..5, all threshold modes)
#include "launch_impl.hpp"
namespace cloudy {
hipError_t launch_n4(const HostPlan &h, const LaunchReq &r) { return launch_n<4>(h, r); }
}  // namespace cloudy
