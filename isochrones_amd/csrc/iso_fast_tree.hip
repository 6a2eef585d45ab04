// Generic StarModel over a flattened ObservationTree ("next" row f4; reference starmodel.py:538-613,
// observation.py:464-491, 1181-1234) on the corner-packed tables: the control flow of k_lnpost_tree
// (iso_hip.hip) with every model-star gather done by the wave-cooperative coop_star / coop_bc of the
// fused kernel.  One lane owns one sample; the leaf loop has a wave-uniform trip count, so all 64 lanes
// reach every cooperative gather together.
#include <cstdlib>

#include "iso_fast_kernel.h"
#include "fast/tree_mailbox.h"

namespace iso {
namespace fastk {

#include "fast/tree_eval.h"

// threads per workgroup: four waves for the register form, one wave for the LDS-resident runtime form
template <int NL>
constexpr int tree_block() { return NL > 0 ? BLOCK : 64; }

template <int NB, int NL>
__global__ __launch_bounds__(tree_block<NL>(), 2) void k_lnpost_tree_fast(const FastArgs A, const DevTree* __restrict__ Tp)
{
    constexpr int TB = tree_block<NL>();
    extern __shared__ double lds[];
    for (int j = threadIdx.x; j < A.axes_len; j += TB) lds[j] = A.axes_blob[j];
    __syncthreads();
    const CoopLds L = coop_lds<NB>(lds, A.axes_len);
    // (the tree's record through the constant address space: scalar loads whatever the compiler can prove about the pointer -
    // as the sampler and the mailbox wave read it; 110.5-112.4 -> 109.0-110.1 us)
    typedef const __attribute__((address_space(4))) DevTree* const_tree_ptr;
    const DevTree& T = *(const DevTree*)((const_tree_ptr)(uintptr_t)Tp);
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    const bool active = i < A.n;
    const int64_t ii = active ? i : (A.n - 1);
    const double* __restrict__ src = A.pars + ii * A.stride_n;
    auto par = [&](int j) { return src[j * A.stride_p]; };        // parameters stay in memory (L1/L2 hits)
    TreeLeaves<NB, NL> S;
    // (unused by the register form.)  The gather slots in front of it are one per lane of THIS workgroup: TB, not BLOCK - the
    // one-wave runtime form had been given the 256 slots of a four-wave workgroup, 14 of its 33 KB, which halved the
    // workgroups a CU holds
    S.lds_ = lds + ((A.axes_len + 1) & ~1) + TB * slot_stride(NB) + threadIdx.x;
    S.stride_ = TB;
    double lnp, lnl;
    const double post = tree_lnpost<NB, NL>(A, T, lds, L, active, par, S, A.lnlike != nullptr, lnp, lnl);
    if (!active) return;
    if (A.lnpost) A.lnpost[i] = post;
    if (A.lnprior) A.lnprior[i] = lnp;
    if (A.lnlike) A.lnlike[i] = lnl;
}

// The model's resident mailbox wave (fast/tree_mailbox.h): one wave polls the request lines and evaluates the one sample with
// all 64 lanes helping in the gathers - tree_lnpost, the batch kernel's device function, with the parameters taken from the
// request words (scalars: every lane computes the sample's brackets, lane 0 owns the result).  Register form only (trees of
// 1-4 stars with up to eight bands - the shapes of the reference's examples); other trees keep the launch per call.
// (A mailbox mode INSIDE the batch kernel was tried first, to save these instantiations: the second inlined copy of the
// evaluation cost the batch kernel 30-40 registers and 250 spilled scalar registers.)
template <int NB, int NL>
__global__ __launch_bounds__(64, 2) void k_mailbox_tree(const FastArgs A, const DevTree* __restrict__ Tp, IsoTreeBox* mb,
                                                        unsigned long long idle_ticks, unsigned long long life_ticks)
{
    static_assert(NL > 0, "register form");
    extern __shared__ double lds[];
    for (int j = threadIdx.x; j < A.axes_len; j += 64) lds[j] = A.axes_blob[j];
    __syncthreads();
    // (a lone wave: the sampler's form of the evaluation - three-lane model gather, stars two at a time; the same bits as the
    // batch kernel's form)
    constexpr int NREQ = tree_requests(NL, NB);
    const CoopLds L = coop_lds_multi<NREQ>(lds, A.axes_len, slot_stride(NB));
    TreeLeaves<NB, NL> S;
    S.lds_ = nullptr;
    S.stride_ = 64;
    double* lpar = lds + ((A.axes_len + 1) & ~1) + NREQ * 64 * slot_stride(NB);      // the request's 32 words behind the gather slots
    const int lane = (int)threadIdx.x;
    // The tree's record (term tables, prior constants: 7 KB) through the constant address space - scalar loads, as the
    // sampler reads it: through the plain pointer an evaluation read it field by field with dependent VECTOR loads, a few
    // hundred cycles each with nothing else to run on a lone wave (lnpost(p) 19.8-20.9 -> 18.0-18.9 us; a copy in LDS: the
    // same, for 7 KB - profiles/r06/tree_mailbox_record_ab.txt)
    typedef const __attribute__((address_space(4))) DevTree* const_tree_ptr;
    const DevTree& T = *(const DevTree*)((const_tree_ptr)(uintptr_t)Tp);
    auto sys_load = [](const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    auto sys_store = [](unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    unsigned long long last = sys_load(&mb->done[0]);
    const unsigned long long t_start = wall_clock64();
    unsigned long long t_idle = t_start;
    const int np = T.n_params;
    for (;;) {
        const unsigned long long w = sys_load(&mb->req[lane & 31]);
        const int wlo = (int)(unsigned)(w & 0xFFFFFFFFull), whi = (int)(unsigned)(w >> 32);
        const unsigned long long seq = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(whi, 0) << 32) |
                                       (unsigned long long)(unsigned)__builtin_amdgcn_readlane(wlo, 0);
        if (seq == last) {
            const unsigned long long now = wall_clock64();
            const bool leave = (now - t_idle > idle_ticks) | (now - t_start > life_ticks) | (sys_load(&mb->ctl[1]) != 0);
            if (leave) break;                              // (wave-uniform)
            continue;
        }
#ifdef ISO_MAILBOX_CLOCK
        const unsigned long long t_seen = wall_clock64();
#ifdef ISO_PHASE_CLOCK
        const unsigned long long c_seen = __builtin_readcyclecounter();
#endif
#endif
        // parameter j of the request: the word lane 1 + j read
        auto word = [&](int j) {
            return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(whi, 1 + j) << 32) |
                   (unsigned long long)(unsigned)__builtin_amdgcn_readlane(wlo, 1 + j);
        };
        uint32_t c = 0x9E3779B9u;                          // mailbox_checksum (iso_internal.h) over the np parameter words
        for (int q = 0; q < np; ++q) {
            const unsigned long long x = word(q);
            c = (c ^ (uint32_t)x) * 0x85EBCA6Bu;
            c = (c ^ (uint32_t)(x >> 32)) * 0xC2B2AE35u + (uint32_t)q;
        }
        if (c != (uint32_t)(seq >> 32)) continue;          // the lines did not arrive together: poll again
        const bool parts = ((seq >> 8) & 1) != 0;
        // the parameters go through LDS and are read from there as the batch kernel reads them from memory: plain loads, so
        // that the evaluation is the same expression tree in both kernels (with the words taken straight from the request
        // registers the compiler shared subexpressions differently and contracted other multiply-adds: lnlike of the
        // one-band one-star tree differed from the batch kernel's in the last bits)
        __builtin_amdgcn_wave_barrier();
        if (lane < 32) lpar[lane] = __longlong_as_double((long long)w);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double* __restrict__ src = lpar + 1;
        auto par = [&](int j) { return src[j]; };
        double lnp, lnl;
        const double post = tree_lnpost<NB, NL, true, true>(A, T, lds, L, lane == 0, par, S, parts, lnp, lnl);
        if (lane == 0) {
            sys_store(&mb->done[1], (unsigned long long)__double_as_longlong(post));
            sys_store(&mb->done[2], (unsigned long long)__double_as_longlong(lnp));
            sys_store(&mb->done[3], (unsigned long long)__double_as_longlong(lnl));
#ifdef ISO_MAILBOX_CLOCK
            sys_store(&mb->done[4], wall_clock64() - t_seen);
#ifdef ISO_PHASE_CLOCK          // shader-clock stamps of the evaluation's phases, from the request's arrival (tree_eval.h: ISO_STAMP)
            sys_store(&mb->done[5], g_phase_stamps[3] - c_seen);          // first model cell in
            sys_store(&mb->done[6], g_phase_stamps[6] - c_seen);          // every leaf gathered, fluxes formed
            sys_store(&mb->done[7], g_phase_stamps[7] - c_seen);          // priors
            sys_store(&mb->ctl[4], g_phase_stamps[8] - c_seen);           // likelihood
            sys_store(&mb->ctl[5], __builtin_readcyclecounter() - c_seen);
#endif
#endif
        }
        __threadfence_system();                            // results before the sequence word
        if (lane == 0) sys_store(&mb->done[0], seq);
        last = seq;
        t_idle = wall_clock64();
    }
    __threadfence_system();
    if (lane == 0) sys_store(&mb->ctl[0], 2ull);          // state: exited
}

}  // namespace fastk

template <int NL>
static bool launch_tree_nl(int nb, int n_leaves, const FastArgs& A, const DevTree* T, hipStream_t s)
{
    using namespace fastk;
    constexpr int TB = tree_block<NL>();
    const dim3 g((unsigned)((A.n + TB - 1) / TB)), b(TB);
    // the runtime-leaf form keeps n_leaves * (6 + bands) values per lane in LDS behind the staged axes and gather slots
    auto sh = [&](int n) {
        return (size_t)(((A.axes_len + 1) & ~1) + TB * slot_stride(n) + (NL > 0 ? 0 : n_leaves * (6 + n) * TB)) * sizeof(double);
    };
    // (beyond the 64 KB a launch gets without asking - seven or eight stars with 10-12 bands, five and more with 13-16 - the
    // kernel is given what it needs of the CU's 160 KB; round 6)
    auto lds_ok = [&](const void* fn, size_t bytes) {
        return bytes <= 64 * 1024 ||
               (bytes <= 160 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess);
    };
    switch (nb) {
#define ISO_TREE_CASE(N) \
    case N: if (!lds_ok((const void*)k_lnpost_tree_fast<N, NL>, sh(N))) return false; \
        note_kernel("k_lnpost_tree_fast<%d, %d>", N, NL); hipLaunchKernelGGL((k_lnpost_tree_fast<N, NL>), g, b, sh(N), s, A, T); return true;
        ISO_TREE_CASE(1) ISO_TREE_CASE(2) ISO_TREE_CASE(3) ISO_TREE_CASE(4) ISO_TREE_CASE(5) ISO_TREE_CASE(6)
        ISO_TREE_CASE(7) ISO_TREE_CASE(8)
    default: break;
    }
    if constexpr (NL != 0) {
        return false;                          // 9-12 bands exist in the runtime-leaf form only
    } else {
        switch (nb) {
            ISO_TREE_CASE(9) ISO_TREE_CASE(10) ISO_TREE_CASE(11) ISO_TREE_CASE(12)
        default: break;
        }
        // 13 ... 16 bands (ISO_TREE_MAX_BANDS): the band-tiled form, per-leaf values laid out for 16 bands
        if (nb > 12 && nb <= ISO_TREE_MAX_BANDS) {
            if (!lds_ok((const void*)k_lnpost_tree_fast<ISO_TREE_MAX_BANDS, NL>, sh(ISO_TREE_MAX_BANDS))) return false;
            note_kernel("k_lnpost_tree_fast<%d, %d>", ISO_TREE_MAX_BANDS, NL);
            hipLaunchKernelGGL((k_lnpost_tree_fast<ISO_TREE_MAX_BANDS, NL>), g, b, sh(ISO_TREE_MAX_BANDS), s, A, T);
            return true;
        }
        return false;
    }
#undef ISO_TREE_CASE
}

template <int NL>
static bool launch_tree_mailbox_nl(int nb, const FastArgs& A, const DevTree* T, IsoTreeBox* d_box, unsigned long long idle,
                                   unsigned long long life, hipStream_t s)
{
    using namespace fastk;
    auto sh = [&](int n) { return (size_t)(((A.axes_len + 1) & ~1) + tree_requests(NL, n) * 64 * slot_stride(n) + 32) * sizeof(double); };
    switch (nb) {
#define ISO_TREE_MB_CASE(N) \
    case N: note_kernel("k_mailbox_tree<%d, %d>", N, NL); hipLaunchKernelGGL((k_mailbox_tree<N, NL>), dim3(1), dim3(64), sh(N), s, A, T, d_box, idle, life); return true;
        ISO_TREE_MB_CASE(1) ISO_TREE_MB_CASE(2) ISO_TREE_MB_CASE(3) ISO_TREE_MB_CASE(4) ISO_TREE_MB_CASE(5) ISO_TREE_MB_CASE(6)
        ISO_TREE_MB_CASE(7) ISO_TREE_MB_CASE(8)
#undef ISO_TREE_MB_CASE
    default: return false;
    }
}

// the tree model's resident mailbox wave (fast/tree_mailbox.h); false: no instantiation for the shape (more than four stars,
// more than eight bands, the runtime-leaf form forced)
bool launch_tree_mailbox(int nb, int n_leaves, const FastArgs& A, const DevTree* T, IsoTreeBox* d_box, unsigned long long idle_ticks,
                         unsigned long long life_ticks, hipStream_t s)
{
    const char* rt = getenv("ISOCHRONES_AMD_TREE_RUNTIME_LEAVES");
    if (nb > 8 || (rt && rt[0] == '1')) return false;
    switch (n_leaves) {
    case 1: return launch_tree_mailbox_nl<1>(nb, A, T, d_box, idle_ticks, life_ticks, s);
    case 2: return launch_tree_mailbox_nl<2>(nb, A, T, d_box, idle_ticks, life_ticks, s);
    case 3: return launch_tree_mailbox_nl<3>(nb, A, T, d_box, idle_ticks, life_ticks, s);
    case 4: return launch_tree_mailbox_nl<4>(nb, A, T, d_box, idle_ticks, life_ticks, s);
    }
    return false;
}

// n_leaves 1..4 with up to 8 bands run the register-resident instantiation, everything else the runtime one
bool launch_tree_fast(int nb, int n_leaves, const FastArgs& A, const DevTree* T, hipStream_t s)
{
    // ISOCHRONES_AMD_TREE_RUNTIME_LEAVES=1 forces the runtime-leaf-count instantiation (tests: it otherwise
    // only serves trees with more than 4 stars or more than 8 bands)
    const char* rt = getenv("ISOCHRONES_AMD_TREE_RUNTIME_LEAVES");
    if (nb <= 8 && !(rt && rt[0] == '1')) {
        switch (n_leaves) {
        case 1: return launch_tree_nl<1>(nb, n_leaves, A, T, s);
        case 2: return launch_tree_nl<2>(nb, n_leaves, A, T, s);
        case 3: return launch_tree_nl<3>(nb, n_leaves, A, T, s);
        case 4: return launch_tree_nl<4>(nb, n_leaves, A, T, s);
        }
    }
    return launch_tree_nl<0>(nb, n_leaves, A, T, s);
}

}  // namespace iso
