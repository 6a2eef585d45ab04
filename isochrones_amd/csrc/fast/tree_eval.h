// Observation-tree lnpost of one sample per lane on the corner-packed tables: the evaluation shared by the batch kernel
// (k_lnpost_tree_fast, iso_fast_tree.hip) and the device-resident sampler of tree models (k_stretch_tree,
// iso_fast_stretch_tree.hip).  Reference semantics: isochrones/starmodel.py:538-613 (lnpost / lnprior of a generic
// StarModel), observation.py:464-491 (model magnitudes of a node), observation.py:1181-1234 (ObservationTree.lnlike).
// (included inside namespace iso::fastk, after iso_fast_kernel.h)
#pragma once

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

// lnpost of the lane's sample.  ALL 64 lanes of a wave must call this together (the gathers are wave-cooperative);
// `active` = the lane really has a sample.  par(j) = parameter j of the lane's sample: global memory in the batch kernel
// (L1 / L2 hits), the proposal rebuilt from two LDS rows in the sampler.  `want_like`: evaluate the likelihood even
// where the prior is not finite (the batch entry point's lnlike output).  An inactive lane's results mean nothing.
// LONE: the caller is a lone workgroup (the sampler): model gather by three lanes per sample (coop_star's THREE).
template <int NB, int NL, bool LONE = false, class Par>
__device__ __forceinline__ double tree_lnpost(const FastArgs& A, const DevTree& T, const double* lds, const CoopLds& L,
                                              bool active, Par par, TreeLeaves<NB, NL>& S, bool want_like,
                                              double& lnp_out, double& lnl_out)
{
    const int n_leaves = (NL > 0) ? NL : T.n_leaves;
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
        if (l == 0) { ISO_STAMP(2, w.t2); }
        coop_star<false, NoWorkBetween, false, LONE || ISO_COOP_STAR3 != 0>(A, L, ok3, cell3(A, i0, i1, i2), w, v);
        if (l == 0) { ISO_STAMP(3, v[0]); }
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
        if (l == 0) { ISO_STAMP(4, w4v.t3); }
        coop_bc<NB>(A, L, ok4, cell4(A, j0, j1, j2, j3), w4v, bc);
        if (l == 0) { ISO_STAMP(5, bc[0]); }
        const double dm = fma(fast_log(dist), 5.0 * kInvLn10, -5.0);      // 5 log10(d / 10), as lnpost_wave takes it
#pragma unroll
        for (int b = 0; b < NB; ++b) S.set_flux(l, b, exp10(-0.4 * (v[3] + dm - bc[b])));
    };
    if constexpr (NL > 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l) leaf(l);
    } else {
        for (int l = 0; l < n_leaves; ++l) leaf(l);
    }
    ISO_STAMP_HERE(6);                        // every leaf gathered, fluxes formed
    lnp_out = lnl_out = f_nan();
    if (!active) return f_nan();              // no cooperative work below
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
    ISO_STAMP(7, lnp);
    const bool prior_ok = isfinite(lnp);
    // ---- lnlike (observation.py:1181-1234): -inf as soon as the running sum is not finite ----
    double lnl = f_nan();
    if (want_like || prior_ok) {
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
            lnl += T.term_g0[t] - (r * r) * T.term_hinv[t];
            if (!isfinite(lnl)) bad = true;
        }
        for (int k = 0; k < T.n_spec && !bad; ++k) {
            const iso_tree_prop& sp = T.spec[k];
            const double r = sp.a - S.prop(sp.leaf, sp.prop);
            lnl += T.spec_g0[k] - (r * r) * T.spec_hinv[k];
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
                    lnl += T.plx_g0[s] - (r * r) * T.plx_hinv[s];
                }
            for (int s = 0; s < T.n_systems; ++s)
                if (T.has_av[s]) {
                    const double r = T.av_val[s] - par(T.sys_base[s] + T.n_stars[s] + 3);
                    lnl += T.av_g0[s] - (r * r) * T.av_hinv[s];
                }
            if (!isfinite(lnl)) bad = true;
        }
        if (bad) lnl = -f_inf();
    }
    ISO_STAMP(8, lnl);
    lnp_out = lnp;
    lnl_out = lnl;
    return prior_ok ? lnp + lnl : -f_inf();
}
