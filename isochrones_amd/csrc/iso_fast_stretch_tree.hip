// Device-resident stretch-move sampler of observation-tree models: fast/sampler_any.h around fast/tree_eval.h.
// Reference: StarModel.fit_mcmc (isochrones/starmodel.py:886-972) on a model whose likelihood is ObservationTree.lnlike
// (observation.py:1181-1234) - e.g. the resolved binaries and triples of docs/multiple.ipynb.
//
// One workgroup = one ensemble for all iterations of the launch.  LDS: [axes][gather slots][leaf values (runtime-leaf form)]
// [positions W x NP][lnpost W][acceptance counters W].  The tree record is read through the constant address space
// (scalar loads: none of the kernel's stores can touch it).  Round 6, measured and not kept: the record copied into LDS when
// the kernel starts and read from there (what helps the lone mailbox wave, iso_fast_tree.hip) - resolved binary 25.2 -> 26.1
// us per step; values read from LDS live in vector registers (four stars x 6-8 bands: 48-148 B of scratch per lane) and
// the scalar loads of four waves hit the scalar cache (profiles/r06/tree_ab_lds_record.txt).
#include "iso_fast_kernel.h"

namespace iso {
namespace fastk {

#include "fast/tree_eval.h"
#include "fast/sampler_any.h"

typedef const __attribute__((address_space(4))) DevTree* const_tree_ptr;

template <int NB, int NL>
struct TreeEval {
    const double* lds;
    const CoopLds& L;
    TreeLeaves<NB, NL>& S;
    template <class Par>
    __device__ __forceinline__ double operator()(kernarg_ptr kp, bool active, Par par) const
    {
        const FastArgs& A = kernarg_at<FastArgs>(kp, ANY_EVAL_ARGS);
        const DevTree* Tp = kernarg_at<const DevTree*>(kp, ANY_EVAL_ARGS + kernarg_align8(sizeof(FastArgs)));
        const DevTree& T = *(const DevTree*)((const_tree_ptr)(uintptr_t)Tp);
        double lnp, lnl;
        return tree_lnpost<NB, NL, true>(A, T, lds, L, active, par, S, false, lnp, lnl);
    }
};

template <int NB, int NL>
__global__ __launch_bounds__(BLOCK, 2) void k_stretch_tree(const AnyStretchArgs S, const FastArgs A, const DevTree* __restrict__ Tp)
{
    extern __shared__ double lds[];
    for (int j = threadIdx.x; j < A.axes_len; j += BLOCK) lds[j] = A.axes_blob[j];
    constexpr int NREQ = 1;
    const CoopLds L = coop_lds<NB>(lds, A.axes_len);
    TreeLeaves<NB, NL> leaves;
    // runtime-leaf form: this lane's column of the [slot][lane] block; lanes beyond S.lanes never take a move (their waves
    // skip the evaluation), so the block is S.lanes wide
    leaves.lds_ = lds + ((A.axes_len + 1) & ~1) + NREQ * BLOCK * slot_stride(NB) + (threadIdx.x < (unsigned)S.lanes ? threadIdx.x : 0);
    leaves.stride_ = S.lanes;
    TreeEval<NB, NL> ev{lds, L, leaves};
    persist_any(ev, S, lds);       // (its first barrier also covers the staged axes)
}

// doubles of LDS in front of the sampler's arrays
static inline size_t tree_eval_doubles(int axes_len, int nb, int n_leaves, bool runtime, int lanes)
{
    return (size_t)((axes_len + 1) & ~1) + (size_t)BLOCK * slot_stride(nb) + (runtime ? (size_t)n_leaves * (6 + nb) * lanes : 0);
}

constexpr size_t LDS_PER_CU = 160 * 1024;

template <int NL>
static bool launch_stretch_tree_nl(int nb, int n_leaves, const FastArgs& A, const DevTree* T, AnyStretchArgs S, int* query, hipStream_t s)
{
    constexpr bool RT = NL == 0;
    // 13 ... 16 bands: the band-tiled runtime-leaf form, laid out for ISO_TREE_MAX_BANDS bands (fast/tree_eval.h)
    const int nb_layout = nb > 12 ? ISO_TREE_MAX_BANDS : nb;
    if (nb > 12 && (!RT || nb > ISO_TREE_MAX_BANDS)) return false;
    // the widest lane count whose LDS fits a CU
    size_t bytes = 0;
    int lanes = BLOCK;
    for (; lanes >= 64; lanes -= 64) {
        bytes = (tree_eval_doubles(A.axes_len, nb_layout, n_leaves, RT, lanes) + any_own_doubles(S.W, S.NP)) * sizeof(double);
        if (bytes <= LDS_PER_CU) break;
        if (!RT) return false;                 // nothing shrinks with the lane count in the register form
    }
    if (lanes < 64) return false;
    S.lanes = lanes;
    S.own_off = (int)tree_eval_doubles(A.axes_len, nb_layout, n_leaves, RT, lanes);
    const void* fn = nullptr;
    switch (nb) {
#define ISO_TREE_STRETCH_CASE(N) case N: fn = (const void*)k_stretch_tree<N, NL>; break;
        ISO_TREE_STRETCH_CASE(1) ISO_TREE_STRETCH_CASE(2) ISO_TREE_STRETCH_CASE(3) ISO_TREE_STRETCH_CASE(4)
        ISO_TREE_STRETCH_CASE(5) ISO_TREE_STRETCH_CASE(6) ISO_TREE_STRETCH_CASE(7) ISO_TREE_STRETCH_CASE(8)
    default: break;
    }
    if constexpr (RT) {
        switch (nb) {
            ISO_TREE_STRETCH_CASE(9) ISO_TREE_STRETCH_CASE(10) ISO_TREE_STRETCH_CASE(11) ISO_TREE_STRETCH_CASE(12)
        default: break;
        }
        if (nb > 12) fn = (const void*)k_stretch_tree<ISO_TREE_MAX_BANDS, NL>;
    }
#undef ISO_TREE_STRETCH_CASE
    if (!fn) return false;
    if (query) {                     // "is there a kernel for this shape, and with how many lanes" - nothing is launched
        *query = lanes;
        return true;
    }
    if (bytes > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    note_kernel("k_stretch_tree<%d, %d>", nb_layout, NL);
    void* args[] = {&S, const_cast<FastArgs*>(&A), &T};
    return hipLaunchKernel(fn, dim3((unsigned)S.n_ens), dim3(BLOCK), args, bytes, s) == hipSuccess;
}

}  // namespace fastk

// 1-4 model stars with up to 8 bands: leaf values in registers; everything else (5-8 stars, 9-12 bands) in LDS.
// `query` non-null: report whether the shape has a kernel whose LDS fits (and the lanes it would use), launch nothing.
bool launch_stretch_tree(int nb, int n_leaves, const FastArgs& A, const DevTree* T, const AnyStretchArgs& S, int* query,
                         hipStream_t s)
{
    using namespace fastk;
    const char* rt = getenv("ISOCHRONES_AMD_TREE_RUNTIME_LEAVES");     // tests: force the runtime-leaf instantiation
    if (nb <= 8 && !(rt && rt[0] == '1')) {
        switch (n_leaves) {
        case 1: return launch_stretch_tree_nl<1>(nb, n_leaves, A, T, S, query, s);
        case 2: return launch_stretch_tree_nl<2>(nb, n_leaves, A, T, S, query, s);
        case 3: return launch_stretch_tree_nl<3>(nb, n_leaves, A, T, S, query, s);
        case 4: return launch_stretch_tree_nl<4>(nb, n_leaves, A, T, S, query, s);
        }
    }
    return launch_stretch_tree_nl<0>(nb, n_leaves, A, T, S, query, s);
}

}  // namespace iso

#ifdef ISO_PHASE_CLOCK
// instrumentation build (tools/phase_clock_any.py): the shader-clock stamps of the last half-step workgroup 0 ran
extern "C" int iso_debug_phase_stamps_tree(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(iso::fastk::g_phase_stamps), 16 * sizeof(unsigned long long));
}
#endif
