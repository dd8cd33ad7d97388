// search_general.hip -- the search kernels (chief-ray aiming, vignetting search, wide-angle pupil
// search; rox_search.hpp) with the trial-ray trace of feature instance F_ALL: one translation
// unit per instance so that the instances compile in parallel.
#include "rox_search.hpp"

namespace rox {
ROX_SEARCH_INSTANCE(general, F_ALL)
}  // namespace rox
