// fast_evenap.hip -- the tolerance-mode (ROX_FAST_FP64) trace kernels of feature instance
// F_EVEN | F_APLIST (rox_device.hpp, "tolerance mode"): every output mode (FULL packets: taken by the host where they pay).  One translation unit per
// instance so that the instances compile in parallel.
#include "rox_device.hpp"

namespace rox {
void launch_evenap_fast(const LaunchCfg &k, const TraceArgs &a) { launch_instance<(F_EVEN | F_APLIST) | F_FAST>(k, a); }
void launch_evenap_fast_batch(const LaunchCfg &k, const TraceArgs *items)
{
    launch_instance_batch<(F_EVEN | F_APLIST) | F_FAST>(k, items);
}
}  // namespace rox
