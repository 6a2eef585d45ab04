// Instantiations of the band-tiled fused lnpost kernel (13-32 bands); see fast/lnpost_wave.h (k_lnpost_wide).
#include "iso_fast_kernel.h"

namespace iso {

bool launch_fast_wide(int kind, int n_stars, const FastArgs& A, hipStream_t s)
{
    const dim3 g((unsigned)((A.n + BLOCK - 1) / BLOCK)), b(BLOCK);
    const size_t sh = (size_t)(((A.axes_len + 1) & ~1) + fastk::coop_lds_doubles(fastk::WIDE_TILE)) * sizeof(double);
    if (kind == ISO_KIND_TRACK) {
        if (n_stars != 1) return false;
        note_kernel("k_lnpost_wide<%d, 1>", ISO_KIND_TRACK);
        hipLaunchKernelGGL((fastk::k_lnpost_wide<ISO_KIND_TRACK, 1>), g, b, sh, s, A);
        return true;
    }
    switch (n_stars) {
    case 1: note_kernel("k_lnpost_wide<%d, 1>", ISO_KIND_ISO); hipLaunchKernelGGL((fastk::k_lnpost_wide<ISO_KIND_ISO, 1>), g, b, sh, s, A); return true;
    case 2: note_kernel("k_lnpost_wide<%d, 2>", ISO_KIND_ISO); hipLaunchKernelGGL((fastk::k_lnpost_wide<ISO_KIND_ISO, 2>), g, b, sh, s, A); return true;
    case 3: note_kernel("k_lnpost_wide<%d, 3>", ISO_KIND_ISO); hipLaunchKernelGGL((fastk::k_lnpost_wide<ISO_KIND_ISO, 3>), g, b, sh, s, A); return true;
    }
    return false;
}

}  // namespace iso
