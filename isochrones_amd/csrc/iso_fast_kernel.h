// Fast fused lnpost kernel (K1+K2), specialised on (parametrisation, #stars, #bands, table layout).
//
// Compared with the generic kernel in iso_hip.hip it
//   * takes the common table shape for granted: model axis 2 (EEP) exactly uniform -> O(1) index;
//     the other six axes staged in LDS together with their reciprocal spacings, so a bracket is a
//     branch-free LDS bisection + one multiply (no fp64 division);
//   * evaluates priors / likelihood in log space with host-precomputed constants (7 transcendental
//     calls per single-star sample instead of ~12, one fp64 division instead of ~56);
//   * can read "corner-packed" tables: every cell stores its own 2^D corners contiguously
//     (model: 8 corners x 6 columns = 384 B = 3 cache lines; BC: 16 corners x nb), so a sample's
//     gather is a few whole lines instead of 4-8 scattered segments.  HBM capacity (288 GB) is
//     traded for line-exact traffic: track table 1.9 GB instead of 0.32 GB.
//   * handles one sample per lane with no grid-stride loop (keeps scalar state short-lived).
//
// Semantics are those of the generic kernel / the reference (NaN and -inf conventions included);
// values agree to a few ulp (reciprocal-multiply instead of divide, log-space products).
#pragma once
#include "iso_internal.h"

namespace iso {
namespace fastk {

__device__ __forceinline__ double f_nan() { return __longlong_as_double(0x7ff8000000000000LL); }
__device__ __forceinline__ double f_inf() { return __longlong_as_double(0x7ff0000000000000LL); }

constexpr double kLogInvRoot2Pi = -0.91893853320467267;
constexpr double kInvRoot2Pi = 0.3989422804014327;
constexpr double kLn10 = 2.302585092994046;
constexpr double kInvLn10 = 0.43429448190325176;

#include "fast/brackets.h"
#include "fast/gather_lane.h"
#include "fast/priors_log.h"
#include "fast/coop_gather.h"
#include "fast/lnpost_wave.h"
#include "fast/sampler.h"
#include "fast/launch.h"

}  // namespace fastk

// one definition per translation unit iso_fast_<tag>.hip
// (the catalog form is only built on the corner-packed layout)
#define ISO_DEFINE_FAST_LAUNCHER(NAME, KIND, NS)                                              \
    bool NAME(int nb, bool packed, bool multi, const FastArgs& A, hipStream_t s)              \
    {                                                                                         \
        if (A.astq) return packed && !multi && fastk::launch_nb<KIND, NS, true, false, true>(nb, A, s); \
        if (multi) return packed && fastk::launch_nb<KIND, NS, true, true>(nb, A, s);         \
        return packed ? fastk::launch_nb<KIND, NS, true, false>(nb, A, s)                     \
                      : fastk::launch_nb<KIND, NS, false, false>(nb, A, s);                   \
    }

#define ISO_DEFINE_STRETCH_LAUNCHER(NAME, KIND, NS)                                           \
    bool NAME(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s)                 \
    {                                                                                         \
        return fastk::launch_stretch_nb<KIND, NS>(nb, A, S, s);                               \
    }

// the sampler kernels of models with asteroseismic terms live in their own translation units
// (iso_fast_ast_<tag>.hip): they double the number of instantiations and would otherwise double the build time
#define ISO_DEFINE_STRETCH_AST_LAUNCHER(NAME, KIND, NS)                                       \
    bool NAME(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s)                 \
    {                                                                                         \
        return fastk::launch_stretch_nb<KIND, NS, true>(nb, A, S, s);                         \
    }

bool launch_stretch_ast_track1(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s);
bool launch_stretch_ast_iso1(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s);
bool launch_stretch_ast_iso2(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s);
bool launch_stretch_ast_iso3(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s);
bool launch_stretch_track1(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s);
bool launch_stretch_iso1(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s);
bool launch_stretch_iso2(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s);
bool launch_stretch_iso3(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s);
bool launch_fast_track1(int nb, bool packed, bool multi, const FastArgs& A, hipStream_t s);
bool launch_fast_iso1(int nb, bool packed, bool multi, const FastArgs& A, hipStream_t s);
bool launch_fast_iso2(int nb, bool packed, bool multi, const FastArgs& A, hipStream_t s);
bool launch_fast_iso3(int nb, bool packed, bool multi, const FastArgs& A, hipStream_t s);

}  // namespace iso
