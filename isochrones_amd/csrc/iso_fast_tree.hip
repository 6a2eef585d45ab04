// Generic StarModel over a flattened ObservationTree ("next" row f4; reference starmodel.py:538-613,
// observation.py:464-491, 1181-1234) on the corner-packed tables: the control flow of k_lnpost_tree
// (iso_hip.hip) with every model-star gather done by the wave-cooperative coop_star / coop_bc of the
// fused kernel.  One lane owns one sample; the leaf loop has a wave-uniform trip count, so all 64 lanes
// reach every cooperative gather together.
#include <cstdlib>

#include "iso_fast_kernel.h"

namespace iso {
namespace fastk {

// NL > 0: the tree has exactly NL model stars, every per-leaf array is indexed at compile time and lives
// in registers (the common 1-4 star trees).  NL = 0: runtime leaf count (5-8 stars, or more than 8 bands): the
// per-leaf values live in LDS, [slot][lane] so that the lanes of a wave touch consecutive addresses - one wave per
// workgroup, n_leaves * (6 + NB) * 64 doubles (5 stars x 3 bands: 23 KB).  (Per-lane scratch arrays, the first form,
// cost 456-1160 B of scratch per lane: 200 MB of write traffic per 10^6 samples.)
template <int NB, int NL>
struct TreeLeaves {
    static constexpr bool STATIC = NL > 0;
    static constexpr int ML = STATIC ? NL : 1;
    static constexpr int PER = 6 + NB;
    double star_[ML][6];
    double flux_[ML][NB];
    double* lds_;          // NL = 0: this lane's column of the [slot][lane] block
    int stride_;           // lanes per workgroup

    __device__ __forceinline__ void set_star(int l, int q, double v)
    {
        if constexpr (STATIC) star_[l][q] = v;
        else lds_[(l * PER + q) * stride_] = v;
    }
    __device__ __forceinline__ void set_flux(int l, int b, double v)
    {
        if constexpr (STATIC) flux_[l][b] = v;
        else lds_[(l * PER + 6 + b) * stride_] = v;
    }
    __device__ __forceinline__ double star(int l, int q) const
    {
        if constexpr (STATIC) return star_[l][q];
        else return lds_[(l * PER + q) * stride_];
    }

    __device__ __forceinline__ double addmags(uint32_t mask, int band, int n_leaves) const
    {
        double tot = 0.0;
        if constexpr (STATIC) {
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int b = 0; b < NB; ++b) tot += (((mask >> l) & 1u) && b == band) ? flux_[l][b] : 0.0;
        } else {
            for (int l = 0; l < n_leaves; ++l)
                if (mask & (1u << l)) tot += lds_[(l * PER + 6 + band) * stride_];
        }
        return -2.5 * fast_log10(tot);
    }

    __device__ __forceinline__ double prop(int leaf, int q) const
    {
        if constexpr (STATIC) {
            double v = 0.0;
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int k = 0; k < 6; ++k) v = (l == leaf && k == q) ? star_[l][k] : v;
            return v;
        } else {
            return lds_[(leaf * PER + q) * stride_];
        }
    }
};

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
    const int n_leaves = (NL > 0) ? NL : T.n_leaves;
    TreeLeaves<NB, NL> S;
    // (unused by the register form.)  The gather slots in front of it are one per lane of THIS workgroup: TB, not BLOCK - the
    // one-wave runtime form had been given the 256 slots of a four-wave workgroup, 14 of its 33 KB, which halved the
    // workgroups a CU holds
    S.lds_ = lds + ((A.axes_len + 1) & ~1) + TB * slot_stride(NB) + threadIdx.x;
    S.stride_ = TB;
    // ---- every model star: model-table gather, then magnitudes as fluxes ----
    auto leaf = [&](int l) {
        const int s = T.leaf_system[l];
        const int base = T.sys_base[s], N = T.n_stars[s];
        const double eep = par(base + T.leaf_slot[l]), age = par(base + N), feh = par(base + N + 1);
        const double dist = par(base + N + 2), AV = par(base + N + 3);
        const bool ok3 = bool(active & !(age != age) & !(feh != feh) & !(eep != eep) & !lds_oob(lds, A.m0, age) &
                         !lds_oob(lds, A.m1, feh) & !eep_oob(A, eep));
        int i0 = 0, i1 = 0, i2 = 0;
        W3 w;
        w.t0 = w.t1 = w.t2 = 0.0;
        if (ok3) {
            lds_bracket2(lds, A.m0, A.m1, age, feh, i0, i1, w.t0, w.t1);
            eep_bracket(A, lds, eep, i2, w.t2);
        }
        double v[6];
        coop_star(A, L, ok3, cell3(A, i0, i1, i2), w, v);
#pragma unroll
        for (int q = 0; q < 6; ++q) S.set_star(l, q, v[q]);
        const double Tf = v[0], g = v[1], f = v[2];
        const bool ok4 = bool(ok3 & !(AV != AV) & !(Tf != Tf) & !(g != g) & !(f != f) & !lds_oob(lds, A.b0, Tf) &
                         !lds_oob(lds, A.b1, g) & !lds_oob(lds, A.b2, f) & !lds_oob(lds, A.b3, AV));
        int j0 = 0, j1 = 0, j2 = 0, j3 = 0;
        W4 w4v;
        w4v.t0 = w4v.t1 = w4v.t2 = w4v.t3 = 0.0;
        if (ok4) lds_bracket4(lds, A.b0, A.b1, A.b2, A.b3, Tf, g, f, AV, j0, j1, j2, j3, w4v.t0, w4v.t1, w4v.t2, w4v.t3);
        double bc[NB];
        coop_bc<NB>(A, L, ok4, cell4(A, j0, j1, j2, j3), w4v, bc);
        const double dm = 5 * log10(dist / 10.0);
#pragma unroll
        for (int b = 0; b < NB; ++b) S.set_flux(l, b, exp10(-0.4 * (v[3] + dm - bc[b])));
    };
    if constexpr (NL > 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l) leaf(l);
    } else {
        for (int l = 0; l < n_leaves; ++l) leaf(l);
    }
    if (!active) return;                      // no cooperative work below
    // ---- lnprior (starmodel.py:557-613) ----
    double lnp = 0.0;
    bool dead = false;
    for (int s = 0; s < T.n_systems && !dead; ++s) {
        const int base = T.sys_base[s], N = T.n_stars[s];
        const DevPrior* pri[4] = {&T.prior_age, &T.prior_feh, &T.prior_distance, &T.prior_AV};
        for (int j = 0; j < 4 && !dead; ++j) {
            const double val = par(base + N + j);
            if (val < T.bound_lo[j] || val > T.bound_hi[j]) { dead = true; break; }
            lnp += ln_pdf<false>(*pri[j], val, 0.0);
            if (!isfinite(lnp)) dead = true;
        }
        for (int j = 1; j < N && !dead; ++j)
            if (!(par(base + j) <= par(base + j - 1))) dead = true;
        if (dead) break;
        auto eep_prior = [&](int l) {
            if (T.leaf_system[l] != s) return;
            const double eep = par(base + T.leaf_slot[l]);
            double term;
            if (eep < T.eep_lo || eep > T.eep_hi) {
                term = -f_inf();
            } else {
                const double lc = ln_call(T.prior_mass, S.star(l, 4)), deriv = S.star(l, 5);
                term = (lc == -f_inf()) ? ((deriv != deriv) ? f_nan() : -f_inf()) : lc + fast_log(deriv);
            }
            lnp += term;
        };
        if constexpr (NL > 0) {
#pragma unroll
            for (int l = 0; l < NL; ++l) eep_prior(l);
        } else {
            for (int l = 0; l < n_leaves; ++l) eep_prior(l);
        }
    }
    if (dead) lnp = -f_inf();
    const bool prior_ok = isfinite(lnp);
    // ---- lnlike (observation.py:1181-1234): -inf as soon as the running sum is not finite ----
    double lnl = f_nan();
    if (A.lnlike || prior_ok) {
        lnl = 0.0;
        bool bad = false;
        for (int t = 0; t < T.n_terms && !bad; ++t) {
            const iso_tree_term& tt = T.terms[t];
            double mag = tt.mag;
            double mod = S.addmags(tt.mask, tt.band, n_leaves);
            if (tt.relative) {
                mod -= S.addmags(tt.ref_mask, tt.band, n_leaves);
                mag -= tt.ref_mag;
            }
            const double r = mag - mod;
            lnl += -0.5 * (r * r) / (tt.unc * tt.unc) + T.term_g0[t];
            if (!isfinite(lnl)) bad = true;
        }
        for (int k = 0; k < T.n_spec && !bad; ++k) {
            const iso_tree_prop& sp = T.spec[k];
            const double r = sp.a - S.prop(sp.leaf, sp.prop);
            lnl += -0.5 * (r * r) / (sp.b * sp.b) + T.spec_g0[k];
            if (!isfinite(lnl)) bad = true;
        }
        for (int k = 0; k < T.n_limits && !bad; ++k) {
            const iso_tree_prop& lm = T.limits[k];
            const double mod = S.prop(lm.leaf, lm.prop);
            if (mod < lm.a || mod > lm.b || !isfinite(mod)) bad = true;
        }
        if (!bad) {
            for (int s = 0; s < T.n_systems; ++s)
                if (T.has_plx[s]) {
                    const double r = T.plx_val[s] - 1.0 / par(T.sys_base[s] + T.n_stars[s] + 2) * 1000.0;
                    lnl += -0.5 * (r * r) / (T.plx_unc[s] * T.plx_unc[s]) + T.plx_g0[s];
                }
            for (int s = 0; s < T.n_systems; ++s)
                if (T.has_av[s]) {
                    const double r = T.av_val[s] - par(T.sys_base[s] + T.n_stars[s] + 3);
                    lnl += -0.5 * (r * r) / (T.av_unc[s] * T.av_unc[s]) + T.av_g0[s];
                }
            if (!isfinite(lnl)) bad = true;
        }
        if (bad) lnl = -f_inf();
    }
    if (A.lnpost) A.lnpost[i] = prior_ok ? lnp + lnl : -f_inf();
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
