// fast_poly.hip -- the tolerance-mode (ROX_FAST_FP64) trace kernels of feature instance
// F_POLY (rox_device.hpp, "tolerance mode"): every output mode (FULL packets: taken by the host where they pay).  One translation unit per
// instance so that the instances compile in parallel.
#include "rox_device.hpp"

namespace rox {
void launch_poly_fast(const LaunchCfg &k, const TraceArgs &a) { launch_instance<(F_POLY) | F_FAST>(k, a); }
void launch_poly_fast_batch(const LaunchCfg &k, const TraceArgs *items)
{
    launch_instance_batch<(F_POLY) | F_FAST>(k, items);
}
}  // namespace rox
