// Instantiations of the stretch-move sampler kernels (step-wise and persistent forms, fast/sampler.h) for (ISO_KIND_ISO, 2 star(s));
// the batch and start-point kernels of the shape: iso_fast_iso2.hip.
#include "iso_fast_kernel.h"

namespace iso {
ISO_DEFINE_STRETCH_LAUNCHER(launch_stretch_iso2, ISO_KIND_ISO, 2)
}  // namespace iso

#ifdef ISO_PHASE_CLOCK
// instrumentation build (tools/phase_clock_shape.py): the shader-clock stamps of the last evaluation workgroup 0 ran
extern "C" int iso_debug_phase_stamps_iso2(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(iso::fastk::g_phase_stamps), 16 * sizeof(unsigned long long));
}
#endif
