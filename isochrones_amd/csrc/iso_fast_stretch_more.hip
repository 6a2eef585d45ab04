// Device-resident stretch-move samplers (fast/sampler_any.h) for the two remaining model shapes that used to fit through
// framework launches:
//   k_stretch_isotrack<NB>   IsoTrackModel - one star asked to agree with BOTH grids (reference isochrones/starmodel.py:2010-2104):
//                            parameters (eep, mass, age, feh, distance, AV); lnlike = isochrone-grid likelihood at
//                            (eep, age, feh, d, AV) + track-grid likelihood at (mass, eep, feh, d, AV), the parallax term
//                            counted once (the isochrone-side model carries none); lnprior = the track model's prior + the
//                            age prior in closed form.  Two fused evaluations (lnpost_wave) per proposal, both tables'
//                            axes staged in LDS side by side.
//   k_stretch_wide<KIND,NS>  BasicStarModel with 13-32 bands: the band-tiled evaluation of k_lnpost_wide.
#include "iso_fast_kernel.h"

namespace iso {
namespace fastk {

#include "fast/sampler_any.h"

typedef const __attribute__((address_space(4))) DevModel* const_model_ptr;

template <int NB>
struct IsoTrackEval {
    const double* lds;
    int lane;
    template <class Par>
    __device__ __forceinline__ double operator()(kernarg_ptr kp, bool active, Par par) const
    {
        constexpr size_t OFF_I = ANY_EVAL_ARGS, OFF_T = OFF_I + kernarg_align8(sizeof(FastArgs)),
                         OFF_P = OFF_T + kernarg_align8(sizeof(FastArgs));
        const FastArgs& Ai = kernarg_at<FastArgs>(kp, OFF_I);
        const FastArgs& At = kernarg_at<FastArgs>(kp, OFF_T);
        const IsoTrackAge& P = kernarg_at<IsoTrackAge>(kp, OFF_P);
        const int base_i = (Ai.axes_len + 1) & ~1, base_t = (At.axes_len + 1) & ~1;
        const double* lds_t = lds + base_i;
        CoopLds L;
        L.req = const_cast<double*>(lds) + base_i + base_t + (threadIdx.x >> 6) * 64 * slot_stride(NB);
        L.rsp = L.req;
        L.stride = slot_stride(NB);
        L.lane = lane;
        const double eep = par(0), mass = par(1), age = par(2), feh = par(3), dist = par(4), AV = par(5);
        const double pt[5] = {mass, eep, feh, dist, AV}, pi[5] = {eep, age, feh, dist, AV};
        const DevModel& Mt = *(const DevModel*)((const_model_ptr)(uintptr_t)At.m);
        const DevModel& Mi = *(const DevModel*)((const_model_ptr)(uintptr_t)Ai.m);
        double t_prior, t_like, i_prior, i_like;
        lnpost_wave<ISO_KIND_TRACK, 1, NB, false, false, false, false, 8>(At, lds_t, L, active, Mt, pt, true, t_prior, t_like);     // (LANE 8: a lone workgroup's model gather)
        lnpost_wave<ISO_KIND_ISO, 1, NB, false, false, false, false, 8>(Ai, lds, L, active, Mi, pi, true, i_prior, i_like);
        // the age prior of the reference's AgePrior (flat in linear age): lnorm + age ln 10 inside its bounds; two
        // roundings, as the host-side composition of the batch path takes them (no contraction into an FMA)
        double ln_age = __dadd_rn(P.lnorm, __dmul_rn(age, kLn10));
        ln_age = (age < P.lo || age > P.hi) ? -f_inf() : ln_age;
        const double lnprior = t_prior + ln_age;
        const double lnlike = i_like + t_like;
        return isfinite(lnprior) ? lnprior + lnlike : -f_inf();
    }
};

template <int NB>
__global__ __launch_bounds__(BLOCK, 2) void k_stretch_isotrack(const AnyStretchArgs S, const FastArgs Ai, const FastArgs At,
                                                               const IsoTrackAge P)
{
    extern __shared__ double lds[];
    const int base_i = (Ai.axes_len + 1) & ~1;
    for (int j = threadIdx.x; j < Ai.axes_len; j += BLOCK) lds[j] = Ai.axes_blob[j];
    for (int j = threadIdx.x; j < At.axes_len; j += BLOCK) lds[base_i + j] = At.axes_blob[j];
    IsoTrackEval<NB> ev{lds, (int)(threadIdx.x & 63)};
    persist_any(ev, S, lds);
}

template <int KIND, int NS>
struct WideEval {
    const double* lds;
    const CoopLds& L;
    template <class Par>
    __device__ __forceinline__ double operator()(kernarg_ptr kp, bool active, Par par) const
    {
        const FastArgs& A = kernarg_at<FastArgs>(kp, ANY_EVAL_ARGS);
        const DevModel& M = *(const DevModel*)((const_model_ptr)(uintptr_t)A.m);
        constexpr int NP = NS + 4;
        double p[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) p[q] = par(q);
        double lnp, lnl;
        return lnpost_wave<KIND, NS, wide_tile(NS), false, false, true, false, 8>(A, lds, L, active, M, p, false, lnp, lnl);
    }
};

template <int KIND, int NS>
__global__ __launch_bounds__(BLOCK, 2) void k_stretch_wide(const AnyStretchArgs S, const FastArgs A)
{
    extern __shared__ double lds[];
    for (int j = threadIdx.x; j < A.axes_len; j += BLOCK) lds[j] = A.axes_blob[j];
    const CoopLds L = coop_lds<WIDE_TILE>(lds, A.axes_len);
    WideEval<KIND, NS> ev{lds, L};
    persist_any(ev, S, lds);
}

constexpr size_t LDS_PER_CU = 160 * 1024;

static bool launch_any(const void* fn, AnyStretchArgs& S, size_t eval_doubles, void** args, const char* name, int* query, hipStream_t s)
{
    const size_t bytes = (eval_doubles + any_own_doubles(S.W, S.NP)) * sizeof(double);
    if (bytes > LDS_PER_CU) return false;
    S.lanes = BLOCK;
    S.own_off = (int)eval_doubles;
    if (query) {
        *query = BLOCK;
        return true;
    }
    if (bytes > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    note_kernel("%s", name);
    return hipLaunchKernel(fn, dim3((unsigned)S.n_ens), dim3(BLOCK), args, bytes, s) == hipSuccess;
}

}  // namespace fastk

bool launch_stretch_isotrack(int nb, const FastArgs& Ai, const FastArgs& At, const IsoTrackAge& P, AnyStretchArgs S,
                             int* query, hipStream_t s)
{
    using namespace fastk;
    const void* fn = nullptr;
    switch (nb) {
#define ISO_IT_CASE(N) case N: fn = (const void*)k_stretch_isotrack<N>; break;
        ISO_IT_CASE(0) ISO_IT_CASE(1) ISO_IT_CASE(2) ISO_IT_CASE(3) ISO_IT_CASE(4) ISO_IT_CASE(5) ISO_IT_CASE(6)
        ISO_IT_CASE(7) ISO_IT_CASE(8) ISO_IT_CASE(9) ISO_IT_CASE(10) ISO_IT_CASE(11) ISO_IT_CASE(12)
#undef ISO_IT_CASE
    default: return false;
    }
    char name[64];
    snprintf(name, sizeof name, "k_stretch_isotrack<%d>", nb);
    void* args[] = {&S, const_cast<FastArgs*>(&Ai), const_cast<FastArgs*>(&At), const_cast<IsoTrackAge*>(&P)};
    const size_t ev = (size_t)((Ai.axes_len + 1) & ~1) + (size_t)((At.axes_len + 1) & ~1) + (size_t)BLOCK * slot_stride(nb);
    return launch_any(fn, S, ev, args, name, query, s);
}

bool launch_stretch_wide(int kind, int n_stars, const FastArgs& A, AnyStretchArgs S, int* query, hipStream_t s)
{
    using namespace fastk;
    const void* fn = nullptr;
    if (kind == ISO_KIND_TRACK) {
        if (n_stars == 1) fn = (const void*)k_stretch_wide<ISO_KIND_TRACK, 1>;
    } else {
        switch (n_stars) {
        case 1: fn = (const void*)k_stretch_wide<ISO_KIND_ISO, 1>; break;
        case 2: fn = (const void*)k_stretch_wide<ISO_KIND_ISO, 2>; break;
        case 3: fn = (const void*)k_stretch_wide<ISO_KIND_ISO, 3>; break;
        }
    }
    if (!fn) return false;
    char name[64];
    snprintf(name, sizeof name, "k_stretch_wide<%d, %d>", kind, n_stars);
    void* args[] = {&S, const_cast<FastArgs*>(&A)};
    const size_t ev = (size_t)((A.axes_len + 1) & ~1) + (size_t)coop_lds_doubles(WIDE_TILE);
    return launch_any(fn, S, ev, args, name, query, s);
}

}  // namespace iso
