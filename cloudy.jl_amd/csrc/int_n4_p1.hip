// instantiation unit: the fused integrators of the N = 4 modes, P = 1 (tensor order 0) family
#include "launch_int_impl.hpp"
namespace cloudy {
template hipError_t launch_int<4, 1>(const HostPlan &h, const LaunchReq &r);
}  // namespace cloudy
