// Instantiations of the fast fused lnpost kernel for (ISO_KIND_TRACK, 1 star(s)); see iso_fast_kernel.h.
#include "iso_fast_kernel.h"

namespace iso {
ISO_DEFINE_FAST_LAUNCHER(launch_fast_track1, ISO_KIND_TRACK, 1)
ISO_DEFINE_STRETCH_LAUNCHER(launch_stretch_track1, ISO_KIND_TRACK, 1)
}  // namespace iso
