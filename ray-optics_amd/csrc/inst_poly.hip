// inst_poly.hip -- the trace kernels of feature instance F_POLY (rox_device.hpp):
// one translation unit per instance so that the instances compile in parallel.
#include "rox_device.hpp"

namespace rox {
void launch_poly(const LaunchCfg &k, const TraceArgs &a) { launch_instance<F_POLY>(k, a); }
void launch_poly_batch(const LaunchCfg &k, const TraceArgs *items) { launch_instance_batch<F_POLY>(k, items); }
}  // namespace rox
