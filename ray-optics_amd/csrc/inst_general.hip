// inst_general.hip -- the trace kernels of feature instance F_ALL (rox_device.hpp):
// one translation unit per instance so that the instances compile in parallel.
#include "rox_device.hpp"

namespace rox {
void launch_general(const LaunchCfg &k, const TraceArgs &a) { launch_instance<F_ALL>(k, a); }
void launch_general_batch(const LaunchCfg &k, const TraceArgs *items) { launch_instance_batch<F_ALL>(k, items); }
}  // namespace rox
