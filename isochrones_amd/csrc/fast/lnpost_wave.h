// lnpost of one sample per lane (shared by the batch, sampler and catalog kernels) and the batch kernel
// (part of iso_fast_kernel.h: included inside namespace iso::fastk)
#pragma once

// lnpost of the lane's sample (p = its NS+4 parameters).  Shared by the batch kernel and the
// sampler kernel.  The gathers are wave-cooperative (corner-packed tables), so ALL 64 lanes of the wave must
// call this function together; `active` = the lane really has a sample (inactive lanes only help).
// MASKED (catalog rows and the sampler kernels): a band whose observed magnitude is NaN is a band this star
// was not observed in - its term is skipped, as the reference drops NaN measurements when it builds a model
// (starmodel.py:1427-1433).
// TILED (batch kernels for 13-32 bands, k_lnpost_wide): NB is the width of a band tile and the photometric terms are
// taken tile by tile over the A.nb_total bands of the corner-packed BC cell (each star's BC bracket is found once and kept).
// STDP: the model's priors are the reference's default families (Chabrier mass, flat-in-age, local-disk [Fe/H],
// power-law distance, flat AV - the host checks the records): the families are compile-time constants here.
// LANE (bit 0: model table and asteroseismic pair, bit 1: BC table): every lane gathers its own sample
// from the corner-packed table (gather_lane.h) - the latency form of a lone workgroup.
// bit 2 of LANE: the priors that do not depend on the model table are evaluated between the issue of the primary's
// model gather and the use of its data (coop_star's `between`).
// bit 3 of LANE: the cooperative model gather takes three lanes per sample and no cross-lane sums (coop_star's THREE): fewer
// instructions, more loads - a lone workgroup's trade.
// bit 4 of LANE (binaries, NS = 2): ONE STAR PER LANE.  The caller has given lanes l and l + 32 of a wave the same sample;
// lane l walks the primary's chain (EEP bracket, model gather, BC brackets, BC gather), lane l + 32 the companion's, side
// by side instead of one after the other - a lone wave has nothing else to overlap them with, and a second star then costs
// one exchange across the wave's halves (6 + NB values) instead of most of an evaluation.  After the exchanges both lanes
// hold both stars' numbers and do the same arithmetic in the same order: the result is the one of the plain form, bit for bit.
// bit 5 of LANE (triples, NS = 3): the same with one star per ROW of the wave - lanes l, l + 16, l + 32 (l < 16) hold the same
// sample and walk star 0 / 1 / 2; row 3 only helps with the gathers.  The three stars' chains were the whole of a triple's
// half-step (round 5: 55 us per step of a 9-band triple against 20 for the binary); side by side they cost one chain and
// two exchanges across the rows (v_permlane16_swap + v_permlane32_swap, three vector instructions per dword).
// Bands up to which the BC gather is taken lane-per-sample where LANE asks for it.  Measured per step of one star's fit
// (tools/single_fit_shapes.py, profiles/r04/lane_bc_cap_ab.jsonl; 256 walkers): 1 band 8.97 us against 9.67 cooperative;
// 2 / 3 / 4 bands 10.08 / 10.58 / 11.63 against 9.35 / 9.67 / 9.93; lifted to 8 bands: 5 / 6 / 8 bands 12.6 / 13.8 / 22.6
// against 10.3 / 10.7 / 11.9 (16 loads of 16 B per band and lane).  The resident catalog kernel at 3 bands: 10^4 stars
// 9.49 -> 8.20 ms cooperative.  Only 32-walker single fits liked 2-3 bands lane-wise (3-6 %).  So: one band.
#ifndef ISO_LANE_BC_MAX_BANDS
#define ISO_LANE_BC_MAX_BANDS 1
#endif
constexpr int LANE_BC_MAX_BANDS = ISO_LANE_BC_MAX_BANDS;
template <int KIND, int NS, int NB, bool ASTERO = false, bool MASKED = false, bool TILED = false, bool STDP = false,
          int LANE = 0>
__device__ __forceinline__ double lnpost_wave(const FastArgs& A, const double* lds, const CoopLds& L, bool active,
                                              const DevModel& M, const DevModel& MP, const double* __restrict__ p, bool want_parts,
                                              double& lnp_out, double& lnl_out)
{
    // MP: the block the mass / age / [Fe/H] / A_V priors and the EEP bounds are read from - the sample's own block M, or
    // (catalogs whose stars share them: FastArgs.shared_priors) the first star's; observations and the distance prior
    // are always M's
    const double q1 = p[NS], feh_par = p[NS + 1], dist = p[NS + 2], AV = p[NS + 3];

    // ---- model table: axes 0/1 are shared by all components of an isochrone system ----
    const double x0 = (KIND == ISO_KIND_TRACK) ? p[2] : q1;        // feh | age
    const double x1 = (KIND == ISO_KIND_TRACK) ? p[0] : feh_par;   // mass | feh
    const bool ok01 = bool(active & !(x0 != x0) & !(x1 != x1) & !lds_oob(lds, A.m0, x0) & !lds_oob(lds, A.m1, x1));
    // brackets are computed for every lane, usable or not (lut_start clamps; an unusable lane's numbers are never used): a
    // branch here would make the table reads wait for the bounds test
    int i0, i1;
    W3 w;
    w.t2 = 0.0;
    lds_bracket2(lds, A.m0, A.m1, x0, x1, i0, i1, w.t0, w.t1);
    // the prior terms that need nothing from the tables: evaluated either right after the gathers (as written in the
    // reference, starmodel.py:1616-1635) or, where a lone workgroup would only wait, while the primary's cell is on its way;
    // added up further down in the reference's order either way, so the sum is the same number
    constexpr int K_MASS = STDP ? ISO_PRIOR_CHABRIER : -1, K_AGE = STDP ? ISO_PRIOR_FLATLOG : -1, K_FEH = STDP ? ISO_PRIOR_FEH : -1,
                  K_DIST = STDP ? ISO_PRIOR_POWERLAW : -1, K_AV = STDP ? ISO_PRIOR_FLAT : -1;
    constexpr bool OVERLAP = (LANE & 4) != 0 && (LANE & 1) == 0;
    constexpr bool THREE = (LANE & 8) != 0 || ISO_COOP_STAR3 != 0;       // model gather by three lanes per sample (coop_gather.h)
    double ld = 0.0, t_first = 0.0, t_feh = 0.0, t_dist = 0.0, t_av = 0.0;
    auto table_free_priors = [&]() {
        ld = fast_log(dist);
        t_first = (KIND == ISO_KIND_TRACK) ? ln_pdf<false, K_MASS>(MP.prior_mass, p[0], 0.0) : ln_pdf<false, K_AGE>(MP.prior_age, q1, 0.0);
        t_feh = ln_pdf<false, K_FEH>(MP.prior_feh, feh_par, 0.0);
        t_dist = ln_pdf<true, K_DIST>(M.prior_distance, dist, ld);
        t_av = ln_pdf<false, K_AV>(MP.prior_AV, AV, 0.0);
    };
    double star[NS][6];
    double astero[2] = {0.0, 0.0};
    constexpr bool SPREAD = (LANE & (16 | 32)) != 0;
    static_assert(!SPREAD || (((LANE & 16) ? NS == 2 : NS == 3) && (LANE & 48) != 48 && KIND == ISO_KIND_ISO && !TILED && !ASTERO && (LANE & 1) == 0),
                  "one star per lane: binaries (bit 4) / triples (bit 5) on the isochrone grid, cooperative model gather");
    // the star this lane walks in the one-star-per-lane forms (a triple's idle fourth row shadows star 2 and asks for nothing)
    const int my_star = !SPREAD ? 0 : (NS == 2 ? (L.lane >> 5) : ((L.lane >> 4) < 2 ? (L.lane >> 4) : 2));
    const bool my_row = !SPREAD || NS == 2 || (L.lane >> 4) < 3;
    if constexpr (SPREAD) {
        double eep = p[0];
#pragma unroll
        for (int s = 1; s < NS; ++s) eep = my_star == s ? p[s] : eep;
        const bool ok = bool(ok01 & my_row & !(eep != eep) & !eep_oob(A, eep));
        int i2;
        eep_bracket(A, lds, eep, i2, w.t2);
        const uint32_t cell = cell3(A, i0, i1, i2);
        double mine[6];
        // (DEEP: the primaries' rounds and the companions' in one flight)
        if constexpr (OVERLAP) coop_star<true, decltype(table_free_priors)&, true, THREE>(A, L, ok, cell, w, mine, table_free_priors);
        else coop_star<false, NoWorkBetween, true, THREE>(A, L, ok, cell, w, mine);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            double each[NS];
            stars_f64<NS>(mine[q], each);       // (binary: primary's lane = lower half, companion's = upper; triple: rows 0 / 1 / 2)
#pragma unroll
            for (int s = 0; s < NS; ++s) star[s][q] = each[s];
        }
    }
#pragma unroll
    for (int s = 0; s < (SPREAD ? 0 : NS); ++s) {
        const double eep = (KIND == ISO_KIND_TRACK) ? p[1] : p[s];
        const bool ok = bool(ok01 & !(eep != eep) & !eep_oob(A, eep));
        int i2;
        eep_bracket(A, lds, eep, i2, w.t2);
        {
            uint32_t cell = cell3(A, i0, i1, i2);
            ISO_STAMP(2, cell);
            if constexpr (LANE & 1) lane_star(A, ok, cell, w, star[s]);
            else if constexpr (OVERLAP) {
                if (s == 0) coop_star<true, decltype(table_free_priors)&, false, THREE>(A, L, ok, cell, w, star[s], table_free_priors);
                else coop_star<false, NoWorkBetween, false, THREE>(A, L, ok, cell, w, star[s]);
            } else coop_star<false, NoWorkBetween, false, THREE>(A, L, ok, cell, w, star[s]);
            ISO_STAMP(3, star[s][0]);
            // asteroseismic pair of the primary (reference starmodel.py:1603-1612); a separate instantiation,
            // because even a never-taken branch here costs the common kernel registers (measured: +29 %)
            if (ASTERO && s == 0) {
                if constexpr (LANE & 1) lane_pair(A.astq, ok && M.has_numax, cell, w, astero);
                else coop_pair(A.astq, L, ok && M.has_numax, cell, w, astero);
            }
        }
    }

    // ---- lnprior ----
    if constexpr (!OVERLAP) table_free_priors();
    double lnp = 0.0;
    bool rejected = false;
    if (NS == 2) rejected = p[1] > p[0];
    if (NS == 3) rejected = !(p[0] > p[1]) && (p[1] > p[2]);
    if (KIND == ISO_KIND_TRACK) lnp += t_first;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const double eep = (KIND == ISO_KIND_TRACK) ? p[1] : p[s];
        lnp += (KIND == ISO_KIND_TRACK) ? eep_term<K_AGE>(MP, MP.prior_age, eep, star[s][4], star[s][5])
                                        : eep_term<K_MASS>(MP, MP.prior_mass, eep, star[s][4], star[s][5]);
    }
    if (KIND == ISO_KIND_ISO) lnp += t_first;
    lnp += t_feh;
    lnp += t_dist;
    lnp += t_av;
    if (rejected) lnp = -f_inf();
    ISO_STAMP(4, lnp);
    const bool prior_ok = active && isfinite(lnp);
    const bool go = active && (prior_ok || want_parts);     // evaluate the likelihood for this lane
    lnp_out = lnp;
    lnl_out = f_nan();

    // ---- lnlike ----
    double lnl = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if constexpr (STDP) {          // a select, not a branch: the nine constants arrive in one batch of scalar loads
            const double val = M.spec_val[q], g0 = M.spec_g0[q], hinv = M.spec_hinv[q];
            const double r = val - star[0][q];
            const double term = g0 - r * r * hinv;
            lnl += (val == val) ? term : 0.0;
        } else {                       // (the batch / catalog kernels sit at their register caps: fetched when needed)
            const double val = M.spec_val[q];
            if (val == val) {
                const double r = val - star[0][q];
                lnl += M.spec_g0[q] - r * r * M.spec_hinv[q];
            }
        }
    }
    const double dm = fma(ld, 5.0 * kInvLn10, -5.0);   // 5*log10(d/10)
    // the parallax term needs nothing from the BC table: a lone workgroup's form (LANE bit 2) takes its division before the BC
    // gather instead of behind it (added to the sum in the same place either way; cfg 4: 8.50 -> 8.44 us per step)
    constexpr bool PLX_EARLY = (LANE & 4) != 0;
    double plx_term = 0.0;
    if constexpr (PLX_EARLY) {
        if (M.has_parallax) {
            const double r = M.plx_val - 1000.0 / dist;
            plx_term = M.plx_g0 - r * r * M.plx_hinv;
        }
    }
    if constexpr (TILED) {
        static_assert(NB > 0, "a band tile has at least one band");
        const int nbt = A.nb_total;
        const bool okA = bool(go & !(AV != AV) & !lds_oob(lds, A.b3, AV));
        bool okb[NS];
        uint32_t cellb[NS];
        W4 wb[NS];
    #pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double T = star[s][0], g = star[s][1], f = star[s][2];
            okb[s] = bool(okA & !(T != T) & !(g != g) & !(f != f) & !lds_oob(lds, A.b0, T) & !lds_oob(lds, A.b1, g) &
                          !lds_oob(lds, A.b2, f));
            int j0, j1, j2, j3;
            lds_bracket4(lds, A.b0, A.b1, A.b2, A.b3, T, g, f, AV, j0, j1, j2, j3, wb[s].t0, wb[s].t1, wb[s].t2, wb[s].t3);
            cellb[s] = cell4(A, j0, j1, j2, j3);
        }
        for (int b0 = 0; b0 < nbt; b0 += NB) {                   // wave-uniform
            double tot[NB], rel[NS > 1 ? NB : 1];
    #pragma unroll
            for (int s = 0; s < NS; ++s) {
                double bc[NB];
                coop_bc_tile<NB>(A, L, okb[s], cellb[s], wb[s], nbt, b0, bc);
    #pragma unroll
                for (int b = 0; b < NB; ++b) {                   // the flux sum of the untiled form below, band by band
                    const double mag = star[s][3] + dm - bc[b];
                    if (NS == 1) {
                        tot[b] = mag;
                    } else if (s == 0) {
                        const bool far = fabs(mag) > 700.0;
                        tot[b] = far ? 0.0 : mag;
                        rel[b] = 1.0;
                        if (__ballot(far)) rel[b] = far ? exp10(-0.4 * mag) : 1.0;
                    } else {
                        rel[b] += exp10(-0.4 * (mag - tot[b]));
                    }
                }
            }
    #pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int band = min(b0 + b, nbt - 1);
                const double mag = (NS == 1) ? tot[b] : fma(-2.5, fast_log10(rel[b]), tot[b]);
                const double mv = M.mag_val[band];
                const double r = mv - mag;
                if (b0 + b < nbt && (!MASKED || mv == mv)) lnl += M.mag_g0[band] - r * r * M.mag_hinv[band];
            }
        }
    } else if constexpr (SPREAD && NB > 0) {
        // each lane gathers the BC of its own star; the halves of the wave swap them; the flux sum then runs as below
        double tot[NB], rel[NB];
        const bool okA = bool(go & !(AV != AV) & !lds_oob(lds, A.b3, AV));
        double bcm[NB];
        {
            double T = star[0][0], g = star[0][1], f = star[0][2];
#pragma unroll
            for (int s = 1; s < NS; ++s) {
                T = my_star == s ? star[s][0] : T;
                g = my_star == s ? star[s][1] : g;
                f = my_star == s ? star[s][2] : f;
            }
            const bool ok = bool(okA & my_row & !(T != T) & !(g != g) & !(f != f) & !lds_oob(lds, A.b0, T) &
                                 !lds_oob(lds, A.b1, g) & !lds_oob(lds, A.b2, f));
            int j0, j1, j2, j3;
            W4 w4v;
            lds_bracket4(lds, A.b0, A.b1, A.b2, A.b3, T, g, f, AV, j0, j1, j2, j3, w4v.t0, w4v.t1, w4v.t2, w4v.t3);
            const uint32_t cell = cell4(A, j0, j1, j2, j3);
            if constexpr ((LANE & 2) != 0 && NB <= LANE_BC_MAX_BANDS) lane_bc<NB>(A, ok, cell, w4v, bcm);
            else coop_bc<NB, true>(A, L, ok, cell, w4v, bcm);
        }
    #pragma unroll
        for (int b = 0; b < NB; ++b) {
            double bcs[NS];
            stars_f64<NS>(bcm[b], bcs);
            {   // primary (the s == 0 step of the loop below)
                const double mag = star[0][3] + dm - bcs[0];
                const bool far = fabs(mag) > 700.0;
                tot[b] = far ? 0.0 : mag;
                rel[b] = 1.0;
                if (__ballot(far)) rel[b] = far ? exp10(-0.4 * mag) : 1.0;
            }
    #pragma unroll
            for (int s = 1; s < NS; ++s) {   // companions
                const double mag = star[s][3] + dm - bcs[s];
                rel[b] += exp10(-0.4 * (mag - tot[b]));
            }
        }
    #pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double mag = fma(-2.5, fast_log10(rel[b]), tot[b]);
            if constexpr (STDP) {
                const double mv = M.mag_val[b], g0 = M.mag_g0[b], hinv = M.mag_hinv[b];
                const double r = mv - mag;
                const double term = g0 - r * r * hinv;
                lnl += (!MASKED || mv == mv) ? term : 0.0;
            } else {
                const double r = M.mag_val[b] - mag;
                if (!MASKED || M.mag_val[b] == M.mag_val[b]) lnl += M.mag_g0[b] - r * r * M.mag_hinv[b];
            }
        }
    } else if constexpr (NB > 0) {   // NB = 0: spectroscopy / parallax only, the BC table is never touched
        double tot[NB], rel[NS > 1 ? NB : 1];
        const bool okA = bool(go & !(AV != AV) & !lds_oob(lds, A.b3, AV));
    #pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double T = star[s][0], g = star[s][1], f = star[s][2];
            const bool ok = bool(okA & !(T != T) & !(g != g) & !(f != f) & !lds_oob(lds, A.b0, T) &
                                 !lds_oob(lds, A.b1, g) & !lds_oob(lds, A.b2, f));
            double bc[NB];
            int j0, j1, j2, j3;
            W4 w4v;
            lds_bracket4(lds, A.b0, A.b1, A.b2, A.b3, T, g, f, AV, j0, j1, j2, j3, w4v.t0, w4v.t1, w4v.t2, w4v.t3);
            {
                uint32_t cell = cell4(A, j0, j1, j2, j3);
                ISO_STAMP(5, cell);
                // (the lane form keeps 16 NB pieces of 16 B in flight per lane: it pays for one band only - LANE_BC_MAX_BANDS)
                if constexpr ((LANE & 2) != 0 && NB <= LANE_BC_MAX_BANDS) lane_bc<NB>(A, ok, cell, w4v, bc);
                else coop_bc<NB>(A, L, ok, cell, w4v, bc);
                ISO_STAMP(6, bc[0]);
            }
            // total magnitude of an unresolved system (reference utils.py:67-75: -2.5 log10 sum_c 10^(-0.4 m_c)) written
            // relative to the primary, m_0 - 2.5 log10(1 + sum_{c>0} 10^(-0.4 (m_c - m_0))): the same number to a few ulp
            // with one exponential less per band.  Beyond |m_0| = 700 mag (distances below 1e-140 pc or above 1e140 pc)
            // the reference's fluxes leave the double range - they overflow to inf or fade through the subnormals to 0,
            // and its result with them; there the sum is taken about 0 instead of m_0, i.e. exactly as the reference
            // writes it, so that even those values and their +-inf pattern are reproduced.
    #pragma unroll
            for (int b = 0; b < NB; ++b) {
                const double mag = star[s][3] + dm - bc[b];
                if (NS == 1) {
                    tot[b] = mag;
                } else if (s == 0) {
                    const bool far = fabs(mag) > 700.0;              // false for NaN
                    tot[b] = far ? 0.0 : mag;
                    rel[b] = 1.0;
                    if (__ballot(far)) rel[b] = far ? exp10(-0.4 * mag) : 1.0;     // wave-uniform branch, never taken in practice
                } else {
                    rel[b] += exp10(-0.4 * (mag - tot[b]));
                }
            }
        }
    #pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double mag = (NS == 1) ? tot[b] : fma(-2.5, fast_log10(rel[b]), tot[b]);
            if constexpr (STDP) {
                const double mv = M.mag_val[b], g0 = M.mag_g0[b], hinv = M.mag_hinv[b];
                const double r = mv - mag;
                const double term = g0 - r * r * hinv;
                lnl += (!MASKED || mv == mv) ? term : 0.0;
            } else {
                const double r = M.mag_val[b] - mag;
                if (!MASKED || M.mag_val[b] == M.mag_val[b]) lnl += M.mag_g0[b] - r * r * M.mag_hinv[b];
            }
        }
    }
    if (M.has_parallax) {
        if constexpr (PLX_EARLY) {
            lnl += plx_term;
        } else {
            const double r = M.plx_val - 1000.0 / dist;
            lnl += M.plx_g0 - r * r * M.plx_hinv;
        }
    }
    if (ASTERO && M.has_numax) {
        const double r = M.numax_val - astero[0];
        lnl += M.numax_g0 - r * r * M.numax_hinv;
        if (M.has_dnu) {
            const double r2 = M.dnu_val - astero[1];
            lnl += M.dnu_g0 - r2 * r2 * M.dnu_hinv;
        }
    }
    ISO_STAMP(7, lnl);
    lnl_out = go ? lnl : f_nan();
    return prior_ok ? lnp + lnl : -f_inf();
}

template <int KIND, int NS, int NB, bool ASTERO = false, bool MASKED = false, bool TILED = false, bool STDP = false,
          int LANE = 0>
__device__ __forceinline__ double lnpost_wave(const FastArgs& A, const double* lds, const CoopLds& L, bool active,
                                              const DevModel& M, const double* __restrict__ p, bool want_parts,
                                              double& lnp_out, double& lnl_out)
{
    return lnpost_wave<KIND, NS, NB, ASTERO, MASKED, TILED, STDP, LANE>(A, lds, L, active, M, M, p, want_parts, lnp_out, lnl_out);
}

// LDS layout of the fast kernels: [axes blob, rounded to an even count][request slots][response slots]
template <int NB>
__device__ __forceinline__ CoopLds coop_lds(double* lds, int axes_len)
{
    constexpr int REQ_STRIDE = slot_stride(NB);
    const int base = (axes_len + 1) & ~1;
    const int wave = threadIdx.x >> 6;
    CoopLds L;
    L.req = lds + base + wave * 64 * REQ_STRIDE;
    L.rsp = L.req;
    L.stride = REQ_STRIDE;
    L.lane = threadIdx.x & 63;
    return L;
}

// waves per SIMD the register allocator must leave room for: 6 for the small single-star kernels
// (88 -> 80 VGPR, a few dwords of scratch; measured +2 %), otherwise whatever the kernel needs
// Multiple systems with many bands: the blanket 4-wave cap of rounds 1-3 held them to 128 registers and 288-364 B of
// scratch per lane.  tools/sweep_fast_waves.py built the two- and three-star kernels for 2, 3 and 4 waves and timed every
// (stars, bands) shape on 10^6-row batches, memory-bound and posterior-like (profiles/r04/fast_waves_*.jsonl): no
// difference up to 8 (binaries) / 7 (triples) bands; beyond, 3 waves win by 9-20 % on both workloads (binary, 12 bands:
// 414 -> 336 us; triple, 11 bands: 543 -> 447 us), and the triple with 12 bands takes 2 (622 -> 482 us).
#ifdef ISO_FAST_WAVES_MULTI
constexpr int fast_min_waves(int ns, int nb) { return (ns == 1 && nb <= 2) ? 6 : (ns == 1 ? 4 : ISO_FAST_WAVES_MULTI); }
#else
constexpr int fast_min_waves(int ns, int nb)
{
    if (ns == 1) return nb <= 2 ? 6 : 4;
    if (ns == 2) return nb >= 9 ? 3 : 4;
    return nb >= 12 ? 2 : (nb >= 8 ? 3 : 4);
}
#endif

// 13-32 bands: the same evaluation with the photometric terms taken in tiles of WIDE_TILE bands (the BC cell of a
// 32-band model is 4 KB: registers hold one tile of it at a time).  Batch form only.
constexpr int WIDE_TILE = 8;
// tile width per multiplicity: a system of 2-3 stars keeps every component's BC bracket and a flux sum per band of the
// tile alive, which at 8 bands spills (84 B of scratch per lane for a binary at 128 registers; with tiles of 4 none -
// 100-118 registers - once MachineLICM no longer hoists constants out of the tile loop)
constexpr int wide_tile(int ns) { return ns == 1 ? WIDE_TILE : 4; }

template <int KIND, int NS>
__global__ __launch_bounds__(BLOCK, 4) void k_lnpost_wide(const FastArgs A)
{
    extern __shared__ double lds[];
    for (int j = threadIdx.x; j < A.axes_len; j += BLOCK) lds[j] = A.axes_blob[j];
    __syncthreads();
    const CoopLds L = coop_lds<WIDE_TILE>(lds, A.axes_len);
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < A.n;
    const int64_t ii = active ? i : (A.n - 1);
    const DevModel& M = A.m[0];
    constexpr int NP = NS + 4;
    double p[NP];
    {
        const double* __restrict__ src = A.pars + ii * A.stride_n;
#pragma unroll
        for (int j = 0; j < NP; ++j) p[j] = src[j * A.stride_p];
    }
    double lnp, lnl;
    const double r = lnpost_wave<KIND, NS, wide_tile(NS), false, false, true>(A, lds, L, active, M, p, A.lnlike != nullptr, lnp, lnl);
    if (active) {
        if (A.lnpost) A.lnpost[i] = r;
        if (A.lnprior) A.lnprior[i] = lnp;
        if (A.lnlike) A.lnlike[i] = lnl;
    }
    if (A.done_flag) {               // single-workgroup host-callback launch: results first, then the flag
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            *reinterpret_cast<volatile unsigned long long*>(A.done_flag) = A.done_seq;
            __threadfence_system();
        }
    }
}

// (the asteroseismic instantiation carries two more gathered values and one more cooperative gather: at the 80
// registers of 6 waves it spills 56-68 B per lane and every added byte of scratch shows - 117 -> 134 us when the
// non-uniform-axis branch raised it from 56 to 68 B; at 5 waves it does not spill)
template <int KIND, int NS, int NB, bool MULTI, bool ASTERO = false>
__global__ __launch_bounds__(BLOCK, ASTERO ? (fast_min_waves(NS, NB) > 5 ? 5 : fast_min_waves(NS, NB)) : fast_min_waves(NS, NB))
void k_lnpost_fast(const FastArgs A)
{
    extern __shared__ double lds[];
    for (int j = threadIdx.x; j < A.axes_len; j += BLOCK) lds[j] = A.axes_blob[j];
    __syncthreads();
    const CoopLds L = coop_lds<NB>(lds, A.axes_len);
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < A.n;
    const int64_t ii = active ? i : (A.n - 1);         // inactive lanes shadow the last sample
    const DevModel& M = A.m[MULTI ? A.star_id[ii] : 0];
    constexpr int NP = NS + 4;
    double p[NP];
    {
        const double* __restrict__ src = A.pars + ii * A.stride_n;
#pragma unroll
        for (int j = 0; j < NP; ++j) p[j] = src[j * A.stride_p];
    }
    double lnp, lnl;
    const DevModel& MP = (MULTI && A.shared_priors) ? A.m[0] : M;
    const double r = lnpost_wave<KIND, NS, NB, ASTERO, MULTI>(A, lds, L, active, M, MP, p, A.lnlike != nullptr, lnp, lnl);
    if (active) {
        if (A.lnpost) A.lnpost[i] = r;
        if (A.lnprior) A.lnprior[i] = lnp;
        if (A.lnlike) A.lnlike[i] = lnl;
    }
    if (A.done_flag) {               // single-workgroup host-callback launch: results first, then the flag
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            *reinterpret_cast<volatile unsigned long long*>(A.done_flag) = A.done_seq;
            __threadfence_system();
        }
    }
}
