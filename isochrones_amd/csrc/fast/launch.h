// kernel launch switches over the band count
// (part of iso_fast_kernel.h: included inside namespace iso::fastk)
#pragma once

// dynamic LDS of the persistent form; the host uses it to decide whether an ensemble fits
inline size_t stretch_persist_lds_bytes(int axes_len, int nb, int W, int np, bool dense = false)
{
    return (size_t)(((axes_len + 1) & ~1) + BLOCK * persist_slot_stride(dense, nb, np - 4) +
                    persist_extra_doubles(W, np, persist_slim(dense, nb, np - 4))) * sizeof(double);
}

template <int KIND, int NS, bool ASTERO = false>
inline bool launch_stretch_nb(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s)
{
    const dim3 b(BLOCK);
    if (S.nsteps > 0) {                                       // persistent: workgroups own whole ensembles
        const int64_t n_ens = S.n_active / (S.W >> 1);
        const int G = persist_group(S.W);
        const dim3 gp((unsigned)((n_ens + G - 1) / G));
        auto shp = [&](int n) { return stretch_persist_lds_bytes(A.axes_len, n, S.W, NS + 4, S.dense != 0); };
        switch (nb) {
        // with S.occupancy_query set: report resident workgroups per CU of this instantiation, launch nothing
#define ISO_PERSIST_CASE(N)                                                                               \
        case N:                                                                                           \
            if (S.occupancy_query) {                                                                      \
                const hipError_t qe = S.dense                                                             \
                    ? hipOccupancyMaxActiveBlocksPerMultiprocessor(S.occupancy_query,                     \
                                                                   k_stretch_persist<KIND, NS, N, true, ASTERO>,  \
                                                                   BLOCK, shp(N))                         \
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(S.occupancy_query,                     \
                                                                   k_stretch_persist<KIND, NS, N, false, ASTERO>, \
                                                                   BLOCK, shp(N));                        \
                return qe == hipSuccess;                                                                  \
            }                                                                                             \
            if (S.dense) hipLaunchKernelGGL((k_stretch_persist<KIND, NS, N, true, ASTERO>), gp, b, shp(N), s, A, S);   \
            else if (S.multi) hipLaunchKernelGGL((k_stretch_persist<KIND, NS, N, false, ASTERO>), gp, b, shp(N), s, A, S); \
            else if (S.std_priors) hipLaunchKernelGGL((k_stretch_persist<KIND, NS, N, false, ASTERO, true, true>), gp, b, shp(N), s, A, S); \
            else hipLaunchKernelGGL((k_stretch_persist<KIND, NS, N, false, ASTERO, true>), gp, b, shp(N), s, A, S);   \
            return true;
            ISO_PERSIST_CASE(0) ISO_PERSIST_CASE(1) ISO_PERSIST_CASE(2) ISO_PERSIST_CASE(3) ISO_PERSIST_CASE(4) ISO_PERSIST_CASE(5)
            ISO_PERSIST_CASE(6) ISO_PERSIST_CASE(7) ISO_PERSIST_CASE(8) ISO_PERSIST_CASE(9) ISO_PERSIST_CASE(10)
            ISO_PERSIST_CASE(11) ISO_PERSIST_CASE(12)
#undef ISO_PERSIST_CASE
        default: return false;
        }
    }
    const dim3 g((unsigned)((S.n_active + BLOCK - 1) / BLOCK));
    auto sh = [&](int n) { return (size_t)(((A.axes_len + 1) & ~1) + coop_lds_doubles(n)) * sizeof(double); };
    switch (nb) {
    case 0: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 0, ASTERO>), g, b, sh(0), s, A, S); return true;
    case 1: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 1, ASTERO>), g, b, sh(1), s, A, S); return true;
    case 2: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 2, ASTERO>), g, b, sh(2), s, A, S); return true;
    case 3: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 3, ASTERO>), g, b, sh(3), s, A, S); return true;
    case 4: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 4, ASTERO>), g, b, sh(4), s, A, S); return true;
    case 5: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 5, ASTERO>), g, b, sh(5), s, A, S); return true;
    case 6: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 6, ASTERO>), g, b, sh(6), s, A, S); return true;
    case 7: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 7, ASTERO>), g, b, sh(7), s, A, S); return true;
    case 8: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 8, ASTERO>), g, b, sh(8), s, A, S); return true;
    case 9: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 9, ASTERO>), g, b, sh(9), s, A, S); return true;
    case 10: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 10, ASTERO>), g, b, sh(10), s, A, S); return true;
    case 11: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 11, ASTERO>), g, b, sh(11), s, A, S); return true;
    case 12: hipLaunchKernelGGL((k_stretch_half<KIND, NS, 12, ASTERO>), g, b, sh(12), s, A, S); return true;
    default: return false;
    }
}

template <int KIND, int NS, bool PACKED, bool MULTI, bool ASTERO = false>
inline bool launch_nb(int nb, const FastArgs& A, hipStream_t s)
{
    const dim3 g((unsigned)((A.n + BLOCK - 1) / BLOCK)), b(BLOCK);
    auto sh = [&](int n) { return (size_t)(((A.axes_len + 1) & ~1) + (PACKED ? coop_lds_doubles(n) : 0)) * sizeof(double); };
    switch (nb) {
    case 0: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 0, PACKED, MULTI, ASTERO>), g, b, sh(0), s, A); return true;
    case 1: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 1, PACKED, MULTI, ASTERO>), g, b, sh(1), s, A); return true;
    case 2: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 2, PACKED, MULTI, ASTERO>), g, b, sh(2), s, A); return true;
    case 3: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 3, PACKED, MULTI, ASTERO>), g, b, sh(3), s, A); return true;
    case 4: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 4, PACKED, MULTI, ASTERO>), g, b, sh(4), s, A); return true;
    case 5: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 5, PACKED, MULTI, ASTERO>), g, b, sh(5), s, A); return true;
    case 6: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 6, PACKED, MULTI, ASTERO>), g, b, sh(6), s, A); return true;
    case 7: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 7, PACKED, MULTI, ASTERO>), g, b, sh(7), s, A); return true;
    case 8: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 8, PACKED, MULTI, ASTERO>), g, b, sh(8), s, A); return true;
    case 9: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 9, PACKED, MULTI, ASTERO>), g, b, sh(9), s, A); return true;
    case 10: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 10, PACKED, MULTI, ASTERO>), g, b, sh(10), s, A); return true;
    case 11: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 11, PACKED, MULTI, ASTERO>), g, b, sh(11), s, A); return true;
    case 12: hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, 12, PACKED, MULTI, ASTERO>), g, b, sh(12), s, A); return true;
    default: return false;
    }
}
