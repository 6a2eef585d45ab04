// Instantiations of the fast fused lnpost kernel and of the start-point kernel for (ISO_KIND_TRACK, 1 star(s)); see iso_fast_kernel.h
// (the sampler kernels of the shape: iso_fast_stretch_track1.hip - a translation unit of their own since round 6: the two halves
// compile side by side, and the sampler half was the longest single compilation of the build).
#include "iso_fast_kernel.h"

namespace iso {
ISO_DEFINE_FAST_LAUNCHER(launch_fast_track1, ISO_KIND_TRACK, 1)
ISO_DEFINE_START_LAUNCHER(launch_start_track1, ISO_KIND_TRACK, 1)
}  // namespace iso
