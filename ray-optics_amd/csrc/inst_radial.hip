// inst_radial.hip -- the trace kernels of feature instance F_RADIAL (rox_device.hpp):
// one translation unit per instance so that the instances compile in parallel.
#include "rox_device.hpp"

namespace rox {
void launch_radial(const LaunchCfg &k, const TraceArgs &a) { launch_instance<F_RADIAL>(k, a); }
void launch_radial_batch(const LaunchCfg &k, const TraceArgs *items) { launch_instance_batch<F_RADIAL>(k, items); }
}  // namespace rox
