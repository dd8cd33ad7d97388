// inst_even.hip -- the trace kernels of feature instance F_EVEN (rox_device.hpp):
// one translation unit per instance so that the instances compile in parallel.
#include "rox_device.hpp"

namespace rox {
void launch_even(const LaunchCfg &k, const TraceArgs &a) { launch_instance<F_EVEN>(k, a); }
void launch_even_batch(const LaunchCfg &k, const TraceArgs *items) { launch_instance_batch<F_EVEN>(k, items); }
}  // namespace rox
