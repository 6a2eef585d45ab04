// interp_mag on the corner-packed tables (rows a7-a8 of the scope table: mags.py:8-124): the same
// wave-cooperative gathers as the fused lnpost kernel, minus priors and likelihood.  One lane owns one
// sample: bracket (LDS axes), 384-B model-cell gather -> (Teff, logg, feh, Mbol), bracket on the BC
// axes, 128*NB-B BC-cell gather, mag_b = Mbol + 5 log10(d/10) - BC_b.
#include "iso_fast_kernel.h"

namespace iso {
namespace fastk {

template <int KIND, int NB>
__global__ __launch_bounds__(BLOCK, 2) void k_interp_mag_fast(const FastArgs A, const MagOut O)
{
    extern __shared__ double lds[];
    for (int j = threadIdx.x; j < A.axes_len; j += BLOCK) lds[j] = A.axes_blob[j];
    __syncthreads();
    const CoopLds L = coop_lds<NB>(lds, A.axes_len);
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < A.n;
    const int64_t ii = active ? i : (A.n - 1);
    const double* __restrict__ src = A.pars + ii * A.stride_n;
    const double p0 = src[0], p1 = src[A.stride_p], p2 = src[2 * A.stride_p];
    const double dist = src[3 * A.stride_p], AV = src[4 * A.stride_p];
    // track: (mass, eep, feh) -> table axes (feh, mass, eep);  iso: (eep, age, feh) -> (age, feh, eep)
    const double x0 = (KIND == ISO_KIND_TRACK) ? p2 : p1;
    const double x1 = (KIND == ISO_KIND_TRACK) ? p0 : p2;
    const double eep = (KIND == ISO_KIND_TRACK) ? p1 : p0;
    const bool ok3 = bool(active & !(x0 != x0) & !(x1 != x1) & !(eep != eep) & !lds_oob(lds, A.m0, x0) &
                     !lds_oob(lds, A.m1, x1) & !eep_oob(A, eep));
    int i0 = 0, i1 = 0, i2 = 0;
    W3 w;
    w.t0 = w.t1 = w.t2 = 0.0;
    if (ok3) {
        lds_bracket2(lds, A.m0, A.m1, x0, x1, i0, i1, w.t0, w.t1);
        eep_bracket(A, lds, eep, i2, w.t2);
    }
    double star[6];
    coop_star(A, L, ok3, cell3(A, i0, i1, i2), w, star);
    const double T = star[0], g = star[1], f = star[2];
    if (active) {
        if (O.Teff) O.Teff[i] = T;
        if (O.logg) O.logg[i] = g;
        if (O.feh) O.feh[i] = f;
    }
    if (!O.mags) return;                                    // wave-uniform
    const bool ok4 = bool(ok3 & !(AV != AV) & !(T != T) & !(g != g) & !(f != f) & !lds_oob(lds, A.b0, T) &
                     !lds_oob(lds, A.b1, g) & !lds_oob(lds, A.b2, f) & !lds_oob(lds, A.b3, AV));
    int j0 = 0, j1 = 0, j2 = 0, j3 = 0;
    W4 w4v;
    w4v.t0 = w4v.t1 = w4v.t2 = w4v.t3 = 0.0;
    if (ok4) lds_bracket4(lds, A.b0, A.b1, A.b2, A.b3, T, g, f, AV, j0, j1, j2, j3, w4v.t0, w4v.t1, w4v.t2, w4v.t3);
    double bc[NB];
    coop_bc<NB>(A, L, ok4, cell4(A, j0, j1, j2, j3), w4v, bc);
    if (active) {
        const double dm = 5.0 * log10(dist / 10.0);
        double* __restrict__ o = O.mags + i * NB;
#pragma unroll
        for (int b = 0; b < NB; ++b) o[b] = star[3] + dm - bc[b];
    }
}

template <int KIND>
static bool launch_mag_nb(int nb, const FastArgs& A, const MagOut& O, hipStream_t s)
{
    const dim3 g((unsigned)((A.n + BLOCK - 1) / BLOCK)), b(BLOCK);
    auto sh = [&](int n) { return (size_t)(((A.axes_len + 1) & ~1) + coop_lds_doubles(n)) * sizeof(double); };
    switch (nb) {
#define ISO_MAG_CASE(N) \
    case N: note_kernel("k_interp_mag_fast<%d, %d>", KIND, N); hipLaunchKernelGGL((k_interp_mag_fast<KIND, N>), g, b, sh(N), s, A, O); return true;
        ISO_MAG_CASE(1) ISO_MAG_CASE(2) ISO_MAG_CASE(3) ISO_MAG_CASE(4) ISO_MAG_CASE(5) ISO_MAG_CASE(6)
        ISO_MAG_CASE(7) ISO_MAG_CASE(8) ISO_MAG_CASE(9) ISO_MAG_CASE(10) ISO_MAG_CASE(11) ISO_MAG_CASE(12)
#undef ISO_MAG_CASE
    default: return false;
    }
}

}  // namespace fastk

bool launch_interp_mag_fast(int kind, int nb, const FastArgs& A, const MagOut& O, hipStream_t s)
{
    return kind == ISO_KIND_TRACK ? fastk::launch_mag_nb<ISO_KIND_TRACK>(nb, A, O, s)
                                  : fastk::launch_mag_nb<ISO_KIND_ISO>(nb, A, O, s);
}

}  // namespace iso
