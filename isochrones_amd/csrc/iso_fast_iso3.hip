// Instantiations of the fast fused lnpost kernel for (ISO_KIND_ISO, 3 star(s)); see iso_fast_kernel.h.
#include "iso_fast_kernel.h"

namespace iso {
ISO_DEFINE_FAST_LAUNCHER(launch_fast_iso3, ISO_KIND_ISO, 3)
ISO_DEFINE_STRETCH_LAUNCHER(launch_stretch_iso3, ISO_KIND_ISO, 3)
ISO_DEFINE_START_LAUNCHER(launch_start_iso3, ISO_KIND_ISO, 3)
}  // namespace iso

#ifdef ISO_PHASE_CLOCK
// instrumentation build (tools/phase_clock_shape.py): the shader-clock stamps of the last evaluation workgroup 0 ran
extern "C" int iso_debug_phase_stamps_iso3(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(iso::fastk::g_phase_stamps), 16 * sizeof(unsigned long long));
}
#endif
