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
    double mag_[ML][NB];   // register form only: the magnitude each flux was formed from (single-leaf nodes, addmags)
    double* lds_;          // NL = 0: this lane's column of the [slot][lane] block
    int stride_;           // lanes per workgroup

    __device__ __forceinline__ void set_star(int l, int q, double v)
    {
        if constexpr (STATIC) star_[l][q] = v;
        else lds_[(l * PER + q) * stride_] = v;
    }
    // flux of leaf l in band b = 10^(-0.4 mag)
    __device__ __forceinline__ void set_flux(int l, int b, double mag)
    {
        const double v = exp10(-0.4 * mag);
        if constexpr (STATIC) {
            flux_[l][b] = v;
            mag_[l][b] = mag;
        } else {
            lds_[(l * PER + 6 + b) * stride_] = v;
        }
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
            // A node above ONE model star (the resolved components of docs/multiple.ipynb; mask is wave-uniform): the reference
            // still goes through addmags, -2.5 log10(10^(-0.4 m)) = m to an ulp while 10^(-0.4 m) stays inside the doubles
            // (|m| < 700, as in lnpost_wave's flux sum) - the logarithm is left out there and only there.
            if ((mask & (mask - 1u)) == 0u && mask != 0u) {
                double m = 0.0;
#pragma unroll
                for (int l = 0; l < NL; ++l)
#pragma unroll
                    for (int b = 0; b < NB; ++b) m = (((mask >> l) & 1u) && b == band) ? mag_[l][b] : m;
                if (!__ballot(fabs(m) > 700.0)) return m;
            }
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

    // the same with the band a compile-time constant (the band-major likelihood of the register form): selects over leaves only
    template <int B>
    __device__ __forceinline__ double addmags_band(uint32_t mask) const
    {
        static_assert(STATIC, "register form");
        if ((mask & (mask - 1u)) == 0u && mask != 0u) {          // (wave-uniform) a node above one model star: see addmags
            double m = 0.0;
#pragma unroll
            for (int l = 0; l < NL; ++l) m = ((mask >> l) & 1u) ? mag_[l][B] : m;
            if (!__ballot(fabs(m) > 700.0)) return m;
        }
        double tot = 0.0;
#pragma unroll
        for (int l = 0; l < NL; ++l) tot += ((mask >> l) & 1u) ? flux_[l][B] : 0.0;
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

// request slots per lane a LONE workgroup's evaluation needs: register form with two or three stars (or four with up to three
// bands) - two stars at a time; four stars with more bands keep one at a time (the second flight's registers would spill
// 50-120 B per lane at the 256 the kernel has)
__host__ __device__ constexpr int tree_requests(int nl, int nb) { return (nl == 2 || nl == 3 || (nl == 4 && nb <= 3)) ? 2 : 1; }

#ifndef ISO_TREE_BANDMAJOR
#define ISO_TREE_BANDMAJOR 1
#endif
// terms evaluated side by side in the band-major likelihood (a slot beyond a band's last term repeats that term and adds
// nothing).  Measured (profiles/r06/tree_ab.txt; resolved binary, 256 walkers x 5 000 / 10^6-row batch): term-major with
// exits 26.0-26.3 us per step / 112.6-114.8 us, band-major one term at a time 25.2-25.3 / 111.1-112.9, two at a time
// 26.6 / 113.7-116.5 (87 -> 122 registers in the batch kernel): one.
#ifndef ISO_TREE_TERM_UNROLL
#define ISO_TREE_TERM_UNROLL 1
#endif

// sum of the photometric terms of bands B .. NB - 1 (compile-time recursion over the band)
template <int NB, int NL, int B>
__device__ __forceinline__ void bterm_sum(const DevTree& T, const TreeLeaves<NB, NL>& S, double& lnl)
{
    if constexpr (B < NB) {
        const int t0 = T.bterm_first[B], t1 = T.bterm_first[B + 1];
        for (int t = t0; t < t1; t += ISO_TREE_TERM_UNROLL) {
            double term[ISO_TREE_TERM_UNROLL];
#pragma unroll
            for (int u = 0; u < ISO_TREE_TERM_UNROLL; ++u) {
                const bool have = t + u < t1;                  // (wave-uniform)
                const DevTreeTerm& tt = T.bterms[have ? t + u : t];
                double mod = S.template addmags_band<B>(tt.mask);
                // (a node that is not relative has an empty reference mask: the select keeps the loads and the logarithm of
                // the two sums independent of the flag)
                const double ref = tt.relative ? S.template addmags_band<B>(tt.ref_mask) : 0.0;
                const double r = tt.dmag - (mod - ref);
                // (explicit fused multiply-adds in the likelihood's terms: the evaluation is instantiated in several kernels -
                // batch, sampler, mailbox wave - and left to -ffp-contract=fast the compiler fused a term's g0 - r^2 h one
                // way in one of them and another way in the next, a 1-ulp difference between the batch kernel and the mailbox)
                term[u] = have ? fma(-(r * r), tt.hinv, tt.g0) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < ISO_TREE_TERM_UNROLL; ++u) lnl += term[u];
        }
        bterm_sum<NB, NL, B + 1>(T, S, lnl);
    }
}

// lnpost of the lane's sample.  ALL 64 lanes of a wave must call this together (the gathers are wave-cooperative);
// `active` = the lane really has a sample.  par(j) = parameter j of the lane's sample: global memory in the batch kernel
// (L1 / L2 hits), the proposal rebuilt from two LDS rows in the sampler.  `want_like`: evaluate the likelihood even
// where the prior is not finite (the batch entry point's lnlike output).  An inactive lane's results mean nothing.
// LONE: the caller is a lone workgroup (the sampler): model gather by three lanes per sample (coop_star's THREE).
// PAIRS: the stars two at a time through the multi-request gathers (the caller laid the gather slots out for tree_requests
// per lane).  The mailbox wave's form - ONE sample, nothing else in flight: 22.2 -> 19.9 us per lnpost(p) call.  The device
// sampler keeps one star at a time: with 32-48 moves per wave and four waves per workgroup the round trips of the waves
// already overlap, and the larger flights cost registers - resolved binary 25.2 -> 26.7 us per step (profiles/r06/tree_ab.txt).
template <int NB, int NL, bool LONE = false, bool PAIRS = false, class Par>
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
        const double dm = fma(fast_log(dist), 5.0 * kInvLn10, -5.0);      // 5 log10(d / 10), as lnpost_wave takes it
        if constexpr (NB > FAST_MAX_NB) {
            // 13 ... ISO_TREE_MAX_BANDS bands (round 6; runtime-leaf form only): the BC cell of A.nb_total bands is taken in tiles
            // of eight (coop_bc_tile, the band-tiled BasicStarModel kernels' gather); NB is the width the per-leaf values
            // are laid out for, the band count a run-time number
            static_assert(NL == 0, "band-tiled trees keep their per-leaf values in LDS");
            constexpr int TILE = 8;
            const int nbt = A.nb_total;
            const uint32_t c4 = cell4(A, j0, j1, j2, j3);
            for (int b0 = 0; b0 < nbt; b0 += TILE) {                  // wave-uniform
                double bc[TILE];
                coop_bc_tile<TILE>(A, L, ok4, c4, w4v, nbt, b0, bc);
#pragma unroll
                for (int b = 0; b < TILE; ++b)
                    if (b0 + b < nbt) S.set_flux(l, b0 + b, v[3] + dm - bc[b]);
            }
        } else {
            double bc[NB];
            if (l == 0) { ISO_STAMP(4, w4v.t3); }
            coop_bc<NB>(A, L, ok4, cell4(A, j0, j1, j2, j3), w4v, bc);
            if (l == 0) { ISO_STAMP(5, bc[0]); }
#pragma unroll
            for (int b = 0; b < NB; ++b) S.set_flux(l, b, v[3] + dm - bc[b]);
        }
    };
    if constexpr (PAIRS && tree_requests(NL, NB) == 2) {
        // A lone WAVE (the mailbox wave of the per-point callback): the stars two at a time through the multi-request gathers
        // (coop_gather.h) - brackets of both, ONE flight for both model cells, BC brackets of both, ONE flight for both BC cells -
        // instead of four dependent memory round trips one behind the other.  Same brackets, same per-round arithmetic: the
        // values are those of the one-star-at-a-time form, bit for bit.
        constexpr int NPAIR = NL / 2;
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) {
            bool ok3[2], ok4[2];
            uint32_t c3[2], c4[2];
            W3 w3[2];
            W4 w4[2];
            double dist2[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int l = 2 * pr + h2;
                const int s = T.leaf_system[l];
                const int base = T.sys_base[s], N = T.n_stars[s];
                const double eep = par(base + T.leaf_slot[l]), age = par(base + N), feh = par(base + N + 1);
                dist2[h2] = par(base + N + 2);
                ok3[h2] = bool(active & !(age != age) & !(feh != feh) & !(eep != eep) & !lds_oob(lds, A.m0, age) &
                               !lds_oob(lds, A.m1, feh) & !eep_oob(A, eep));
                int i0 = 0, i1 = 0, i2 = 0;
                w3[h2].t0 = w3[h2].t1 = w3[h2].t2 = 0.0;
                if (ok3[h2]) {
                    lds_bracket2(lds, A.m0, A.m1, age, feh, i0, i1, w3[h2].t0, w3[h2].t1);
                    eep_bracket(A, lds, eep, i2, w3[h2].t2);
                }
                c3[h2] = cell3(A, i0, i1, i2);
            }
            double v2[2][6];
            coop_star_multi<2>(A, L, ok3, c3, w3, v2);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int l = 2 * pr + h2;
                const int s = T.leaf_system[l];
                const double AV = par(T.sys_base[s] + T.n_stars[s] + 3);
#pragma unroll
                for (int q = 0; q < 6; ++q) S.set_star(l, q, v2[h2][q]);
                const double Tf = v2[h2][0], g = v2[h2][1], f = v2[h2][2];
                ok4[h2] = bool(ok3[h2] & !(AV != AV) & !(Tf != Tf) & !(g != g) & !(f != f) & !lds_oob(lds, A.b0, Tf) &
                               !lds_oob(lds, A.b1, g) & !lds_oob(lds, A.b2, f) & !lds_oob(lds, A.b3, AV));
                int j0 = 0, j1 = 0, j2 = 0, j3 = 0;
                w4[h2].t0 = w4[h2].t1 = w4[h2].t2 = w4[h2].t3 = 0.0;
                if (ok4[h2]) lds_bracket4(lds, A.b0, A.b1, A.b2, A.b3, Tf, g, f, AV, j0, j1, j2, j3, w4[h2].t0, w4[h2].t1, w4[h2].t2, w4[h2].t3);
                c4[h2] = cell4(A, j0, j1, j2, j3);
            }
            double bc2[2][NB];
            coop_bc_multi<NB, 2>(A, L, ok4, c4, w4, bc2);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int l = 2 * pr + h2;
                const double dm = fma(fast_log(dist2[h2]), 5.0 * kInvLn10, -5.0);
#pragma unroll
                for (int b = 0; b < NB; ++b) S.set_flux(l, b, v2[h2][3] + dm - bc2[h2][b]);
            }
        }
        if constexpr (NL & 1) leaf(NL - 1);
    } else
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
    if (T.std_priors) {
        // The reference's default families (the host checks the records: DevTree.std_priors) as compile-time constants and
        // WITHOUT the reference's early exits.  A sum that has left the finite numbers never comes back (adding anything to
        // +-inf or NaN gives +-inf or NaN), so "dead at some point" = "a bound or order test failed, or the sum is not finite
        // at one of the reference's check points": the same -inf for the same samples, the same additions in the same order
        // for the others.  Straight-line selects instead of a switch per prior on a family that first has to arrive from
        // memory: resolved-binary fit 31.0 -> 25.9 us per step (profiles/r05/tree_tail_ab.txt).
        for (int s = 0; s < T.n_systems; ++s) {
            const int base = T.sys_base[s], N = T.n_stars[s];
            const double v_age = par(base + N), v_feh = par(base + N + 1), v_dist = par(base + N + 2), v_av = par(base + N + 3);
            dead |= bool((v_age < T.bound_lo[0]) | (v_age > T.bound_hi[0]) | (v_feh < T.bound_lo[1]) | (v_feh > T.bound_hi[1]) |
                         (v_dist < T.bound_lo[2]) | (v_dist > T.bound_hi[2]) | (v_av < T.bound_lo[3]) | (v_av > T.bound_hi[3]));
            const double t_age = ln_pdf<false, ISO_PRIOR_FLATLOG>(T.prior_age, v_age, 0.0);
            const double t_feh = ln_pdf<false, ISO_PRIOR_FEH>(T.prior_feh, v_feh, 0.0);
            const double t_dist = ln_pdf<false, ISO_PRIOR_POWERLAW>(T.prior_distance, v_dist, 0.0);
            const double t_av = ln_pdf<false, ISO_PRIOR_FLAT>(T.prior_AV, v_av, 0.0);
            lnp += t_age;
            dead |= !isfinite(lnp);
            lnp += t_feh;
            dead |= !isfinite(lnp);
            lnp += t_dist;
            dead |= !isfinite(lnp);
            lnp += t_av;
            dead |= !isfinite(lnp);
            for (int j = 1; j < N; ++j) dead |= !(par(base + j) <= par(base + j - 1));
            auto eep_prior_std = [&](int l) {
                if (T.leaf_system[l] != s) return;              // (wave-uniform)
                const double eep = par(base + T.leaf_slot[l]);
                const double lc = ln_call<ISO_PRIOR_CHABRIER>(T.prior_mass, S.star(l, 4)), deriv = S.star(l, 5);
                const double inside = (lc == -f_inf()) ? ((deriv != deriv) ? f_nan() : -f_inf()) : lc + fast_log(deriv);
                lnp += (eep < T.eep_lo || eep > T.eep_hi) ? -f_inf() : inside;
            };
            if constexpr (NL > 0) {
#pragma unroll
                for (int l = 0; l < NL; ++l) eep_prior_std(l);
            } else {
                for (int l = 0; l < n_leaves; ++l) eep_prior_std(l);
            }
        }
    } else
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
#if ISO_TREE_BANDMAJOR
        if constexpr (NL > 0) {
            // REGISTER FORM: the terms band by band (DevTree.bterms, sorted on the host), the band a compile-time index, so that
            // a node's flux sum selects over its leaves only (the term-major loop resolved (leaf, band) through NL x NB selects,
            // twice for one-leaf nodes), and WITHOUT the reference's exit after a non-finite partial sum: a sum that has left
            // the finite numbers never comes back, so "-inf as soon as the running sum is not finite" (observation.py:1209-
            // 1230) is "-inf if the sum of the photometric terms is not finite" - one test behind the loop instead of a
            // loop-carried exit that put every term's logarithm behind the previous one's.  The additions run band-major
            // instead of in tree order: the same terms, rounding apart.
            bterm_sum<NB, NL, 0>(T, S, lnl);
            bad = !isfinite(lnl);
        } else
#endif
        for (int t = 0; t < T.n_terms && !bad; ++t) {
            const iso_tree_term& tt = T.terms[t];
            double mag = tt.mag;
            double mod = S.addmags(tt.mask, tt.band, n_leaves);
            if (tt.relative) {
                mod -= S.addmags(tt.ref_mask, tt.band, n_leaves);
                mag -= tt.ref_mag;
            }
            const double r = mag - mod;
            lnl += fma(-(r * r), T.term_hinv[t], T.term_g0[t]);
            if (!isfinite(lnl)) bad = true;
        }
        // (register form: no exits - a partial sum that is not finite stays so, one test at the end says the same as the
        // reference's test after every addition; the exits only put each term behind the one before)
        constexpr bool EXITS = !(ISO_TREE_BANDMAJOR && NL > 0);
        for (int k = 0; k < T.n_spec && !(EXITS && bad); ++k) {
            const iso_tree_prop& sp = T.spec[k];
            const double r = sp.a - S.prop(sp.leaf, sp.prop);
            lnl += fma(-(r * r), T.spec_hinv[k], T.spec_g0[k]);
            if (EXITS && !isfinite(lnl)) bad = true;
        }
        for (int k = 0; k < T.n_limits && !(EXITS && bad); ++k) {
            const iso_tree_prop& lm = T.limits[k];
            const double mod = S.prop(lm.leaf, lm.prop);
            if (mod < lm.a || mod > lm.b || !isfinite(mod)) bad = true;
        }
        if (!EXITS || !bad) {
            for (int s = 0; s < T.n_systems; ++s)
                if (T.has_plx[s]) {
                    const double r = fma(-(1.0 / par(T.sys_base[s] + T.n_stars[s] + 2)), 1000.0, T.plx_val[s]);
                    lnl += fma(-(r * r), T.plx_hinv[s], T.plx_g0[s]);
                }
            for (int s = 0; s < T.n_systems; ++s)
                if (T.has_av[s]) {
                    const double r = T.av_val[s] - par(T.sys_base[s] + T.n_stars[s] + 3);
                    lnl += fma(-(r * r), T.av_hinv[s], T.av_g0[s]);
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
