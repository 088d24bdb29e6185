// instantiation unit: the fused integrators of the N = 1 modes, P = 4 (tensor order 3) family
#include "launch_int_impl.hpp"
namespace cloudy {
template hipError_t launch_int<1, 4>(const HostPlan &h, const LaunchReq &r);
}  // namespace cloudy
