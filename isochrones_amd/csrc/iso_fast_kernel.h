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

// Natural logarithm for the fused kernels.  The device library's log() is correctly rounded at the price of ~100 VALU
// instructions (76 of them fp64); a sample of the single-star model takes four logarithms, a binary with six bands twelve
// - a fifth to a quarter of the arithmetic of a kernel that is VALU-issue bound whenever its tables are cache-resident
// (profiles/r02: MCMC-like batches, the sampler).  This is the classic reduction x = 2^k m, m in [sqrt(1/2), sqrt(2)),
// s = (m - 1) / (m + 1), log m = 2 s + s^3 (2/3 + ...) with the degree-14 minimax polynomial of the fdlibm family and a
// split ln 2: ~35 instructions, error < 0.8 ulp over 2 x 10^7 arguments (tools/fast_log_check.c runs the same
// arithmetic on the host).  Zero, negative, infinite, NaN and subnormal arguments behave as log() does.
__device__ __forceinline__ double fast_log(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1), exact (subnormals included)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m * 2.0 : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0, d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);                   // reciprocal seed, two Newton steps, one residual correction
    double t = fma(-d, r, 1.0);
    r = fma(r, t, r);
    t = fma(-d, r, 1.0);
    r = fma(r, t, r);
    double sq = f * r;
    sq = fma(r, fma(-d, sq, f), sq);
    const double z = sq * sq, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)e;
    const double v = dk * 6.93147180369123816490e-01 - ((hfsq - fma(sq, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
    const double special = (x == 0.0) ? -f_inf() : ((x > 0.0) ? x : f_nan());       // 0 -> -inf, +inf -> +inf, else NaN
    return (x > 0.0 && x < f_inf()) ? v : special;
}

__device__ __forceinline__ double fast_log10(double x) { return fast_log(x) * kInvLn10; }

// ---- the two halves of a wavefront (gfx950: v_permlane32_swap_b32 swaps lanes 32..63 of one register with lanes 0..31
// of another - one vector instruction per dword where __shfl_xor(x, 32) is two ds_bpermute through the LDS pipeline plus
// the selects around it) ----
// lower / upper = the value lane (l & 31) / lane (l & 31) + 32 holds, in every lane
__device__ __forceinline__ void halves_f64(double x, double& lower, double& upper)
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    lower = __hiloint2double((int)r1[0], (int)r0[0]);
    upper = __hiloint2double((int)r1[1], (int)r0[1]);
}
// ---- the four rows of a wavefront (gfx950: v_permlane16_swap_b32 swaps rows 1 / 3 - lanes 16..31 / 48..63 - of one register
// with rows 0 / 2 of another) ----
// r0 / r1 / r2 = the value lane (l & 15) / (l & 15) + 16 / (l & 15) + 32 holds, in every lane: three swaps per dword
__device__ __forceinline__ void rows3_u32(unsigned x, unsigned& r0, unsigned& r1, unsigned& r2)
{
    const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);       // a[0] = rows {0, 0, 2, 2}, a[1] = rows {1, 1, 3, 3}
    const auto e = __builtin_amdgcn_permlane32_swap(a[0], a[0], false, false); // e[0] = row 0 everywhere, e[1] = row 2 everywhere
    const auto o = __builtin_amdgcn_permlane32_swap(a[1], a[1], false, false); // o[0] = row 1 everywhere
    r0 = e[0];
    r1 = o[0];
    r2 = e[1];
}
__device__ __forceinline__ void rows3_f64(double x, double& r0, double& r1, double& r2)
{
    unsigned l0, l1, l2, h0, h1, h2;
    rows3_u32((unsigned)__double2loint(x), l0, l1, l2);
    rows3_u32((unsigned)__double2hiint(x), h0, h1, h2);
    r0 = __hiloint2double((int)h0, (int)l0);
    r1 = __hiloint2double((int)h1, (int)l1);
    r2 = __hiloint2double((int)h2, (int)l2);
}
// the value every star's lane of a one-star-per-lane move holds (lnpost_wave's LANE bits 4 / 5): out[s] = star s's
template <int NS>
__device__ __forceinline__ void stars_f64(double x, double* out)
{
    static_assert(NS == 2 || NS == 3, "one star per lane: binaries (halves of the wave) and triples (rows)");
    if constexpr (NS == 2) halves_f64(x, out[0], out[1]);
    else rows3_f64(x, out[0], out[1], out[2]);
}
// (Tried on top of it, round 5: logarithms / exponentials of a single star's fit two at a time, the second argument in the idle
// upper half of the wave.  121 vector instructions fewer per move and 0.4 % - a lone wave is bound by the latency of its
// dependent instructions, and two independent logarithms already overlap in the pipeline - and the restructured code was no
// longer bit-identical to the step-wise kernel's under -ffp-contract=fast: taken back.)

// Instrumentation builds only (tools/phase_clock.py: -DISO_PHASE_CLOCK): lane 0 of workgroup 0 stores the shader clock at
// the phase boundaries of an evaluation; `val` is pinned so that the phase's result exists when the clock is read.
#ifdef ISO_PHASE_CLOCK
static __device__ unsigned long long g_phase_stamps[16];
#define ISO_STAMP(k, val)                                                                                   \
    do {                                                                                                    \
        asm volatile("" : "+v"(val));                                                                       \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_stamps[k] = __builtin_readcyclecounter();          \
    } while (0)
#define ISO_STAMP_HERE(k)                                                                                   \
    do {                                                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_stamps[k] = __builtin_readcyclecounter();          \
    } while (0)
#else
#define ISO_STAMP(k, val) do { } while (0)
#define ISO_STAMP_HERE(k) do { } while (0)
#endif

// gathers of the single-model sampler (lnpost_wave's LANE): measured on cfg 4 (phase clocks, profiles/r03) the BC cell
// (8 x 16 B per band) comes in sooner lane-per-sample (1 750 -> 700-1 000 cycles), the model cell (24 x 16 B, three
// lines touched eight times each by every lane) does not (3 700 -> 5 500-6 500): cooperative model gather, lane BC gather
// persistent sampler kernels: argument blocks re-read from the kernel-argument segment in every half-step (sampler.h);
// -DISO_KERNARG_REREAD=0 builds the A/B counterpart
#ifndef ISO_KERNARG_REREAD
#define ISO_KERNARG_REREAD 1
#endif
// (bit 2: the table-free priors are evaluated while the primary's model cell is on its way)
// (bit 3: the model gather by three lanes per sample without cross-lane sums - 9.11 -> 8.73 us per step on cfg 4)
#ifndef ISO_UNI_LANE
#define ISO_UNI_LANE 14
#endif
#ifndef ISO_DENSE_LANE
#define ISO_DENSE_LANE 0
#endif
// the register-capped catalog kernel reads the shared priors through scalar loads (the host never picks it for a launch whose
// stars have priors of their own); 0 builds the A/B counterpart
#ifndef ISO_DENSE_SHARED
#define ISO_DENSE_SHARED 1
#endif
#ifndef ISO_MULTI_LANE
#define ISO_MULTI_LANE 0
#endif
// register-capped catalog kernel, workgroups that their ensembles do not fill: moves packed into full waves, rotated over the
// SIMDs (sampler.h); 0 = spread evenly as in the latency forms (A/B switch)
#ifndef ISO_DENSE_PACKED
#define ISO_DENSE_PACKED 1
#endif
// model-table gather of the cooperative form: 1 = three lanes per sample EVERYWHERE, each summing all eight corners of one
// column pair (no cross-lane sums); 0 = only where lnpost_wave's LANE bit 3 asks for it, four lanes per sample with DPP quad
// sums otherwise (coop_gather.h).  Same bits either way.  (A/B switch.)
#ifndef ISO_COOP_STAR3
#define ISO_COOP_STAR3 0
#endif
#ifndef ISO_COOP_STAR3_PER
#define ISO_COOP_STAR3_PER 2
#endif
// the resident catalog kernel of stars that share the reference's default priors (STDP without UNI): a small catalog is
// latency-bound like a single star's fit - lane BC gather (one band: lnpost_wave.h), the table-free priors during the model
// gather and (round 6) the model gather by three lanes per sample: 313 stars at 32 walkers 3.03 -> 2.81 ms, 1 250 stars
// 3.50 -> 3.43, 10^4 stars 8.39 -> 8.23-8.41 (profiles/r06/catalog_msl.jsonl; the same rows)
#ifndef ISO_MULTI_STD_LANE
#define ISO_MULTI_STD_LANE 14
#endif

#include "fast/brackets.h"
#include "fast/gather_lane.h"
#include "fast/priors_log.h"
#include "fast/coop_gather.h"
#include "fast/lnpost_wave.h"
#include "fast/sampler.h"
#include "fast/start_points.h"
#include "fast/launch.h"

}  // namespace fastk

// one definition per translation unit iso_fast_<tag>.hip
// (the fused kernels read the corner-packed tables only; a model whose interpolator has none runs the generic kernel)
#define ISO_DEFINE_FAST_LAUNCHER(NAME, KIND, NS)                                              \
    bool NAME(int nb, bool multi, const FastArgs& A, hipStream_t s)                           \
    {                                                                                         \
        if (A.astq) return !multi && fastk::launch_nb<KIND, NS, false, true>(nb, A, s);       \
        return multi ? fastk::launch_nb<KIND, NS, true>(nb, A, s) : fastk::launch_nb<KIND, NS, false>(nb, A, s); \
    }

#define ISO_DEFINE_START_LAUNCHER(NAME, KIND, NS)                                             \
    bool NAME(int nb, const FastArgs& A, const fastk::StartArgs& T, hipStream_t s)            \
    {                                                                                         \
        return fastk::launch_start_nb<KIND, NS>(nb, A, T, s);                                 \
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
bool launch_fast_wide(int kind, int n_stars, const FastArgs& A, hipStream_t s);
bool launch_start_track1(int nb, const FastArgs& A, const fastk::StartArgs& T, hipStream_t s);
bool launch_start_iso1(int nb, const FastArgs& A, const fastk::StartArgs& T, hipStream_t s);
bool launch_start_iso2(int nb, const FastArgs& A, const fastk::StartArgs& T, hipStream_t s);
bool launch_start_iso3(int nb, const FastArgs& A, const fastk::StartArgs& T, hipStream_t s);
bool launch_fast_track1(int nb, bool multi, const FastArgs& A, hipStream_t s);
bool launch_fast_iso1(int nb, bool multi, const FastArgs& A, hipStream_t s);
bool launch_fast_iso2(int nb, bool multi, const FastArgs& A, hipStream_t s);
bool launch_fast_iso3(int nb, bool multi, const FastArgs& A, hipStream_t s);

}  // namespace iso
