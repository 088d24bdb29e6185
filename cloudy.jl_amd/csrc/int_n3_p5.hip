// instantiation unit: the fused integrators of the N = 3 modes, P = 5 (tensor order 4) family
#include "launch_int_impl.hpp"
namespace cloudy {
template hipError_t launch_int<3, 5>(const HostPlan &h, const LaunchReq &r);
}  // namespace cloudy
