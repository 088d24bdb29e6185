// instantiation unit: every kernel of the N = 2 mode family (P = 1..5, all threshold modes)
#include "launch_impl.hpp"
namespace cloudy {
hipError_t launch_n2(const HostPlan &h, const LaunchReq &r) { return launch_n<2>(h, r); }
}  // namespace cloudy
