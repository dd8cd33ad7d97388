// inst_lean.hip -- the trace kernels of feature instance 0 (rox_device.hpp):
// one translation unit per instance so that the instances compile in parallel.
#include "rox_device.hpp"

namespace rox {
void launch_lean(const LaunchCfg &k, const TraceArgs &a) { launch_instance<0>(k, a); }
void launch_lean_batch(const LaunchCfg &k, const TraceArgs *items) { launch_instance_batch<0>(k, items); }
}  // namespace rox
