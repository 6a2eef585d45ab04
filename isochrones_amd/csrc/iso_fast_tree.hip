// Generic StarModel over a flattened ObservationTree ("next" row f4; reference starmodel.py:538-613,
// observation.py:464-491, 1181-1234) on the corner-packed tables: the control flow of k_lnpost_tree
// (iso_hip.hip) with every model-star gather done by the wave-cooperative coop_star / coop_bc of the
// fused kernel.  One lane owns one sample; the leaf loop has a wave-uniform trip count, so all 64 lanes
// reach every cooperative gather together.
#include <cstdlib>

#include "iso_fast_kernel.h"

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
    const DevTree& T = *Tp;
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
    if (sh(nb) > 64 * 1024) return false;      // (7-8 stars x 10-12 bands: the generic tree kernel takes those)
    switch (nb) {
#define ISO_TREE_CASE(N) \
    case N: note_kernel("k_lnpost_tree_fast<%d, %d>", N, NL); hipLaunchKernelGGL((k_lnpost_tree_fast<N, NL>), g, b, sh(N), s, A, T); return true;
        ISO_TREE_CASE(1) ISO_TREE_CASE(2) ISO_TREE_CASE(3) ISO_TREE_CASE(4) ISO_TREE_CASE(5) ISO_TREE_CASE(6)
        ISO_TREE_CASE(7) ISO_TREE_CASE(8)
    default: break;
    }
    if constexpr (NL != 0) {
        return false;                          // 9-12 bands exist in the runtime-leaf form only
    } else {
        switch (nb) {
            ISO_TREE_CASE(9) ISO_TREE_CASE(10) ISO_TREE_CASE(11) ISO_TREE_CASE(12)
        default: return false;
        }
    }
#undef ISO_TREE_CASE
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
