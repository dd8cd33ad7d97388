// gtab_general.hip -- the trace kernels of the general instance with the surface table left in
// GLOBAL memory and read through scalar loads in EVERY output mode (F_ALL | F_GTAB,
// rox_device.hpp ctblp): the instance of tables that do not fit the 160 KiB of LDS a workgroup
// may have -- a SequentialModel has no size limit (rayoptics/seq/sequential.py:167-202).
// Bit-identical to the LDS instances (same functions, another pointer type).  Its reduced-output
// kernels are the general instance's own where gtab_of() sends those to the global table anyway
// (the same template instantiations: emitted once per translation unit, one definition linked).
#include "rox_device.hpp"

namespace rox {
void launch_general_gtab(const LaunchCfg &k, const TraceArgs &a) { launch_instance<F_ALL | F_GTAB>(k, a); }
void launch_general_gtab_batch(const LaunchCfg &k, const TraceArgs *items)
{
    launch_instance_batch<F_ALL | F_GTAB>(k, items);
}
}  // namespace rox
