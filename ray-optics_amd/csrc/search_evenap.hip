// search_evenap.hip -- the search kernels (chief-ray aiming, vignetting search, wide-angle pupil
// search; rox_search.hpp) with the trial-ray trace of feature instance F_EVEN | F_APLIST (a Zemax
// import with EVENASPH surfaces): one translation unit per instance so that the instances
// compile in parallel.
#include "rox_search.hpp"

namespace rox {
ROX_SEARCH_INSTANCE(evenap, F_EVEN | F_APLIST)
}  // namespace rox
