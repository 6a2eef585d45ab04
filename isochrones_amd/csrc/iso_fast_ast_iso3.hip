// Sampler kernels (step-wise + persistent) of models with asteroseismic terms for (ISO_KIND_ISO, 3 star(s));
// see iso_fast_kernel.h and fast/sampler.h.
#include "iso_fast_kernel.h"

namespace iso {
ISO_DEFINE_STRETCH_AST_LAUNCHER(launch_stretch_ast_iso3, ISO_KIND_ISO, 3)
}  // namespace iso
