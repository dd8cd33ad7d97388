// gtab_general.hip -- the trace kernels of the general instance with the surface table left in
// GLOBAL memory and read through scalar loads (F_ALL | F_GTAB, rox_device.hpp ctblp): the
// instance of tables that do not fit the 160 KiB of LDS a workgroup may have -- a
// SequentialModel has no size limit (rayoptics/seq/sequential.py:167-202).  Bit-identical to
// the LDS instances (same functions, another pointer type).
#include "rox_device.hpp"

namespace rox {
#if (ROX_GTAB_EXACT & 1)
// (the general instance itself reads its table that way in this build: inst_general.hip)
void launch_general_gtab(const LaunchCfg &k, const TraceArgs &a) { launch_general(k, a); }
void launch_general_gtab_batch(const LaunchCfg &k, const TraceArgs *items) { launch_general_batch(k, items); }
#else
void launch_general_gtab(const LaunchCfg &k, const TraceArgs &a) { launch_instance<F_ALL | F_GTAB>(k, a); }
void launch_general_gtab_batch(const LaunchCfg &k, const TraceArgs *items)
{
    launch_instance_batch<F_ALL | F_GTAB>(k, items);
}
#endif
}  // namespace rox
