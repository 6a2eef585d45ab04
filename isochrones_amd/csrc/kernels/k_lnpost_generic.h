// K1+K2 generic fused lnpost kernel (any axis kind, any band count).  One lane = one sample; when the interpolator /
// model carry corner-packed tables (Grid3V::hotq, Grid4V::tabq) every gather reads its cell's contiguous block -
// 384 B + 128 B per band, whole 128-B lines - instead of 8 + 16 scattered rows of the compact tables.
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

// -------------------------------------------------------------------------------------------
// K1+K2 fused: lnpost
// -------------------------------------------------------------------------------------------
struct PostArgs {
    Grid3V g3;
    Grid4V g4;           // packed to the model's bands (ncol == n_bands)
    const DevModel* m;
    const double* pars;
    int64_t stride_n, stride_p, n;
    double *lnpost, *lnprior, *lnlike;
};

// NB > 0: compile-time band count (register-resident accumulators); NB == 0: runtime loop.
// (Whether lnprior / lnlike are wanted as well is a run-time flag: as a template axis it doubled the family - 72 kernels -
// for a fallback kernel whose time goes into its lane-per-sample gathers.)
template <int KIND, int NS, int NB>
__global__ __launch_bounds__(BLOCK, 2) void k_lnpost(const PostArgs A)
{
    const bool PARTS = A.lnprior != nullptr || A.lnlike != nullptr;
    extern __shared__ double lds[];
    stage_axes<3>(A.g3.ax, lds);
    stage_axes<4>(A.g4.ax, lds);
    __syncthreads();
    const DevModel& M = *A.m;
    constexpr int NP = NS + 4;
    // one sample per lane, no grid-stride loop: a loop keeps all ~110 dwords of the two grid descriptors alive around its
    // body - more scalar registers than the machine has, 167-208 of them spilled to vector lanes and fetched back with
    // v_readlane (3 801 vector instructions per wave against 1 273 in the fused kernel); straight-line code loads a
    // descriptor field where it is used
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < A.n) {
        double p[NP];
        {
            const double* __restrict__ src = A.pars + i * A.stride_n;
#pragma unroll
            for (int j = 0; j < NP; ++j) p[j] = src[j * A.stride_p];
        }
        const double q1 = p[NS], feh_par = p[NS + 1], dist = p[NS + 2], AV = p[NS + 3];
        // q1: track -> eep (p[1]); iso -> age.   For the track case NS == 1: p = (mass, eep, feh, d, AV)

        // ---- locate + gather every component on the hot model table ----
        Cell3 c3[NS];
        bool ok3[NS];
        double star[NS][6];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            double x0, x1, x2;
            if (KIND == ISO_KIND_TRACK) to_axes<KIND>(p[0], p[1], p[2], x0, x1, x2);
            else to_axes<KIND>(p[s], q1, feh_par, x0, x1, x2);
            ok3[s] = locate3(A.g3, lds, x0, x1, x2, c3[s]);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (ok3[s]) {
                if (A.g3.hotq) gather3q(A.g3, c3[s], star[s]);      // corner-packed: whole lines
                else gather3<6>(A.g3, c3[s], star[s]);
            } else {
#pragma unroll
                for (int q = 0; q < 6; ++q) star[s][q] = d_nan();
            }
        }

        // ---- lnprior (reference: starmodel.py:1616-1635) ----
        double lnp = 0.0;
        bool rejected = false;
        if (NS == 2) rejected = p[1] > p[0];
        if (NS == 3) rejected = !(p[0] > p[1]) && (p[1] > p[2]);
        if (KIND == ISO_KIND_TRACK) lnp += prior_lnpdf(M.prior_mass, p[0]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double eep = (KIND == ISO_KIND_TRACK) ? p[1] : p[s];
            double term;
            if (eep < M.eep_lo || eep > M.eep_hi) {
                term = -d_inf();
            } else {
                const DevPrior& orig = (KIND == ISO_KIND_TRACK) ? M.prior_age : M.prior_mass;
                const double pdf = prior_call(orig, star[s][4]) * star[s][5];
                term = (pdf != 0) ? log(pdf) : -d_inf();
            }
            lnp += term;
        }
        if (KIND == ISO_KIND_ISO) lnp += prior_lnpdf(M.prior_age, q1);
        lnp += prior_lnpdf(M.prior_feh, feh_par);
        lnp += prior_lnpdf(M.prior_distance, dist);
        lnp += prior_lnpdf(M.prior_AV, AV);
        if (rejected) lnp = -d_inf();
        const bool prior_ok = isfinite(lnp);

        // ---- lnlike (reference: likelihood.py:16-147, starmodel.py:1599-1612) ----
        double lnl = d_nan();
        if (PARTS || prior_ok) {
            lnl = 0.0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const double val = M.spec_val[q];
                if (val == val) lnl += gauss_term(val, M.spec_g0[q], M.spec_unc2[q], star[0][q]);
            }
            const double dm = 5 * log10(dist / 10.0);
            Cell4 c4[NS];
            bool ok4[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) ok4[s] = locate4(A.g4, lds, star[s][0], star[s][1], star[s][2], AV, c4[s]);
            if (NB > 0) {
                double tot[NB > 0 ? NB : 1];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    double bc[NB > 0 ? NB : 1];
                    if (ok4[s]) {
                        if (A.g4.tabq) gather4q_packed<(NB > 0 ? NB : 1)>(A.g4, c4[s], bc);
                        else gather4_packed<(NB > 0 ? NB : 1)>(A.g4, c4[s], bc);
                    } else {
#pragma unroll
                        for (int b = 0; b < NB; ++b) bc[b] = d_nan();
                    }
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const double mag = star[s][3] + dm - bc[b];
                        if (NS == 1) tot[b] = mag;
                        else tot[b] = (s == 0 ? 0.0 : tot[b]) + exp10(-0.4 * mag);
                    }
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const double mag = (NS == 1) ? tot[b] : -2.5 * log10(tot[b]);
                    lnl += gauss_term(M.mag_val[b], M.mag_g0[b], M.mag_unc2[b], mag);
                }
            } else {
                for (int b = 0; b < M.n_bands; ++b) {
                    double tot = 0.0;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const double bc = !ok4[s] ? d_nan()
                                          : (A.g4.tabq ? gather4q_col(A.g4, c4[s], b) : gather4_col(A.g4, c4[s], b));
                        const double mag = star[s][3] + dm - bc;
                        if (NS == 1) tot = mag;
                        else tot += exp10(-0.4 * mag);
                    }
                    const double mag = (NS == 1) ? tot : -2.5 * log10(tot);
                    lnl += gauss_term(M.mag_val[b], M.mag_g0[b], M.mag_unc2[b], mag);
                }
            }
            if (M.has_parallax) lnl += gauss_term(M.plx_val, M.plx_g0, M.plx_unc2, 1000.0 / dist);
            if (M.has_numax) {
                double a2[2];
                if (ok3[0]) {
                    gather3_astero(A.g3, c3[0], a2);
                } else {
                    a2[0] = a2[1] = d_nan();
                }
                lnl += gauss_term(M.numax_val, M.numax_g0, M.numax_unc2, a2[0]);
                if (M.has_dnu) lnl += gauss_term(M.dnu_val, M.dnu_g0, M.dnu_unc2, a2[1]);
            }
        }
        if (A.lnpost) A.lnpost[i] = prior_ok ? lnp + lnl : -d_inf();
        if (PARTS) {
            if (A.lnprior) A.lnprior[i] = lnp;
            if (A.lnlike) A.lnlike[i] = lnl;
        }
    }
}
