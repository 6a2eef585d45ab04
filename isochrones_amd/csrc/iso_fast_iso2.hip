// Instantiations of the fast fused lnpost kernel for (ISO_KIND_ISO, 2 star(s)); see iso_fast_kernel.h.
#include "iso_fast_kernel.h"

namespace iso {
ISO_DEFINE_FAST_LAUNCHER(launch_fast_iso2, ISO_KIND_ISO, 2)
ISO_DEFINE_STRETCH_LAUNCHER(launch_stretch_iso2, ISO_KIND_ISO, 2)
ISO_DEFINE_START_LAUNCHER(launch_start_iso2, ISO_KIND_ISO, 2)
}  // namespace iso
