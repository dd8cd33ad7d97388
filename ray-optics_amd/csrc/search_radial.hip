// search_radial.hip -- the search kernels (chief-ray aiming, vignetting search, wide-angle pupil
// search; rox_search.hpp) with the trial-ray trace of feature instance F_RADIAL: one translation
// unit per instance so that the instances compile in parallel.
#include "rox_search.hpp"

namespace rox {
ROX_SEARCH_INSTANCE(radial, F_RADIAL)
}  // namespace rox
