// instantiation unit: the fused integrators of the N = 1 modes, P = 3 (tensor order 2) family
#include "launch_int_impl.hpp"
namespace cloudy {
template hipError_t launch_int<1, 3>(const HostPlan &h, const LaunchReq &r);
}  // namespace cloudy
