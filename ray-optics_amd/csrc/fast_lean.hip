// fast_lean.hip -- the tolerance-mode (ROX_FAST_FP64) trace kernels of feature instance
// 0 (rox_device.hpp, "tolerance mode"): every output mode (FULL packets: taken by the host where they pay).  One translation unit per
// instance so that the instances compile in parallel.
#include "rox_device.hpp"

namespace rox {
void launch_lean_fast(const LaunchCfg &k, const TraceArgs &a) { launch_instance<(0) | F_FAST>(k, a); }
void launch_lean_fast_batch(const LaunchCfg &k, const TraceArgs *items)
{
    launch_instance_batch<(0) | F_FAST>(k, items);
}
}  // namespace rox
