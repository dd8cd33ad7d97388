// inst_aplist.hip -- the trace kernels of feature instance F_APLIST (rox_device.hpp):
// one translation unit per instance so that the instances compile in parallel.
#include "rox_device.hpp"

namespace rox {
void launch_aplist(const LaunchCfg &k, const TraceArgs &a) { launch_instance<F_APLIST>(k, a); }
void launch_aplist_batch(const LaunchCfg &k, const TraceArgs *items) { launch_instance_batch<F_APLIST>(k, items); }
}  // namespace rox
