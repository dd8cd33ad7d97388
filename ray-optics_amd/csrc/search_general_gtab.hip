// search_general_gtab.hip -- the search kernels (rox_search.hpp) of a system whose table does
// not fit the LDS of a workgroup: the trial-ray trace of feature instance F_ALL over the
// table left in global memory (F_GTAB, scalar loads).
#include "rox_search.hpp"

namespace rox {
ROX_SEARCH_INSTANCE(general_gtab, F_ALL | F_GTAB)
}  // namespace rox
