// inst_evenap.hip -- the trace kernels of feature instance F_EVEN | F_APLIST (rox_device.hpp):
// even aspheres + clear-aperture lists, i.e. a Zemax import with EVENASPH surfaces.
// One translation unit per instance so that the instances compile in parallel.
#include "rox_device.hpp"

namespace rox {
void launch_evenap(const LaunchCfg &k, const TraceArgs &a) { launch_instance<F_EVEN | F_APLIST>(k, a); }
void launch_evenap_batch(const LaunchCfg &k, const TraceArgs *items)
{
    launch_instance_batch<F_EVEN | F_APLIST>(k, items);
}
}  // namespace rox
