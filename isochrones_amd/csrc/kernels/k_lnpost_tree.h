// generic observation-tree lnpost kernel
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

// -------------------------------------------------------------------------------------------
// "next" row f4: generic StarModel over a flattened ObservationTree
// (reference semantics: isochrones/starmodel.py:538-613, observation.py:464-491, 1181-1234)
// -------------------------------------------------------------------------------------------
struct TreeArgs {
    Grid3V g3;
    Grid4V g4;            // BC packed to the tree's bands (ncol == n_bands)
    const DevTree* T;
    const double* pars;
    int64_t stride_n, stride_p, n;
    double *lnpost, *lnprior, *lnlike;
};

__device__ __forceinline__ double tree_addmags(const double (*flux)[ISO_TREE_MAX_BANDS], uint32_t mask, int band,
                                               int n_leaves)
{
    double tot = 0.0;
    for (int l = 0; l < n_leaves; ++l)
        if (mask & (1u << l)) tot += flux[l][band];
    return -2.5 * log10(tot);
}

__global__ __launch_bounds__(BLOCK, 2) void k_lnpost_tree(const TreeArgs A)
{
    extern __shared__ double lds[];
    stage_axes<3>(A.g3.ax, lds);
    stage_axes<4>(A.g4.ax, lds);
    __syncthreads();
    const DevTree& T = *A.T;
    // one sample per lane (no grid-stride loop: see k_lnpost)
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < A.n) {
        double p[ISO_TREE_MAX_PARAMS];
        {
            const double* __restrict__ src = A.pars + i * A.stride_n;
            for (int j = 0; j < T.n_params; ++j) p[j] = src[j * A.stride_p];
        }
        // ---- every model star: model-table gather, then magnitudes as fluxes ----
        double star[ISO_TREE_MAX_LEAVES][6];
        double flux[ISO_TREE_MAX_LEAVES][ISO_TREE_MAX_BANDS];
        for (int l = 0; l < T.n_leaves; ++l) {
            const int s = T.leaf_system[l];
            const int base = T.sys_base[s], N = T.n_stars[s];
            const double eep = p[base + T.leaf_slot[l]], age = p[base + N], feh = p[base + N + 1];
            const double dist = p[base + N + 2], AV = p[base + N + 3];
            Cell3 c3;
            if (locate3(A.g3, lds, age, feh, eep, c3)) {
                gather3<6>(A.g3, c3, star[l]);
            } else {
                for (int q = 0; q < 6; ++q) star[l][q] = d_nan();
            }
            Cell4 c4;
            const bool ok = locate4(A.g4, lds, star[l][0], star[l][1], star[l][2], AV, c4);
            const double dm = 5 * log10(dist / 10.0);
            for (int b = 0; b < T.n_bands; ++b) {
                const double bc = ok ? gather4_col(A.g4, c4, b) : d_nan();
                flux[l][b] = exp10(-0.4 * (star[l][3] + dm - bc));
            }
        }
        // ---- lnprior (starmodel.py:557-613) ----
        double lnp = 0.0;
        bool dead = false;
        for (int s = 0; s < T.n_systems && !dead; ++s) {
            const int base = T.sys_base[s], N = T.n_stars[s];
            const DevPrior* pri[4] = {&T.prior_age, &T.prior_feh, &T.prior_distance, &T.prior_AV};
            for (int j = 0; j < 4 && !dead; ++j) {
                const double val = p[base + N + j];
                if (val < T.bound_lo[j] || val > T.bound_hi[j]) { dead = true; break; }
                lnp += prior_lnpdf(*pri[j], val);
                if (!isfinite(lnp)) dead = true;
            }
            for (int j = 1; j < N && !dead; ++j)
                if (!(p[base + j] <= p[base + j - 1])) dead = true;
            if (dead) break;
            for (int l = 0; l < T.n_leaves; ++l) {
                if (T.leaf_system[l] != s) continue;
                const double eep = p[base + T.leaf_slot[l]];
                double term;
                if (eep < T.eep_lo || eep > T.eep_hi) {
                    term = -d_inf();
                } else {
                    const double pdf = prior_call(T.prior_mass, star[l][4]) * star[l][5];
                    term = (pdf != 0) ? log(pdf) : -d_inf();
                }
                lnp += term;
            }
        }
        if (dead) lnp = -d_inf();
        const bool prior_ok = isfinite(lnp);
        // ---- lnlike (observation.py:1181-1234): -inf as soon as the running sum is not finite ----
        double lnl = d_nan();
        if (A.lnlike || prior_ok) {
            lnl = 0.0;
            bool bad = false;
            for (int t = 0; t < T.n_terms && !bad; ++t) {
                const iso_tree_term& tt = T.terms[t];
                double mag = tt.mag;
                double mod = tree_addmags(flux, tt.mask, tt.band, T.n_leaves);
                if (tt.relative) {
                    mod -= tree_addmags(flux, tt.ref_mask, tt.band, T.n_leaves);
                    mag -= tt.ref_mag;
                }
                const double r = mag - mod;
                lnl += -0.5 * (r * r) / (tt.unc * tt.unc) + T.term_g0[t];
                if (!isfinite(lnl)) bad = true;
            }
            for (int k = 0; k < T.n_spec && !bad; ++k) {
                const iso_tree_prop& sp = T.spec[k];
                const double r = sp.a - star[sp.leaf][sp.prop];
                lnl += -0.5 * (r * r) / (sp.b * sp.b) + T.spec_g0[k];
                if (!isfinite(lnl)) bad = true;
            }
            for (int k = 0; k < T.n_limits && !bad; ++k) {
                const iso_tree_prop& lm = T.limits[k];
                const double mod = star[lm.leaf][lm.prop];
                if (mod < lm.a || mod > lm.b || !isfinite(mod)) bad = true;
            }
            if (!bad) {
                for (int s = 0; s < T.n_systems; ++s)
                    if (T.has_plx[s]) {
                        const double r = T.plx_val[s] - 1.0 / p[T.sys_base[s] + T.n_stars[s] + 2] * 1000.0;
                        lnl += -0.5 * (r * r) / (T.plx_unc[s] * T.plx_unc[s]) + T.plx_g0[s];
                    }
                for (int s = 0; s < T.n_systems; ++s)
                    if (T.has_av[s]) {
                        const double r = T.av_val[s] - p[T.sys_base[s] + T.n_stars[s] + 3];
                        lnl += -0.5 * (r * r) / (T.av_unc[s] * T.av_unc[s]) + T.av_g0[s];
                    }
                if (!isfinite(lnl)) bad = true;
            }
            if (bad) lnl = -d_inf();
        }
        if (A.lnpost) A.lnpost[i] = prior_ok ? lnp + lnl : -d_inf();
        if (A.lnprior) A.lnprior[i] = lnp;
        if (A.lnlike) A.lnlike[i] = lnl;
    }
}
