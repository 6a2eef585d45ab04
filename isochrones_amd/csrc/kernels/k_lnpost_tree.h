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

// Per-leaf values (6 model columns + one flux per band) live in LDS as [slot][lane] columns behind the staged axes - the
// lanes of a wave touch consecutive addresses - instead of per-lane arrays (star[8][6] + flux[8][16] were 1 664 B of scratch
// per lane: every value written to and read back from memory).  The host sizes the block for the tree at hand,
// n_leaves * (6 + n_bands) * lanes doubles, and launches workgroups of `lanes` = 256, 128 or 64 threads so that it fits;
// the parameters stay in global memory (L1 / L2 hits), as in the fused form.
struct TreeStore {
    double* base;       // this lane's column
    int stride;         // lanes per workgroup
    int per;            // 6 + n_bands
    __device__ __forceinline__ double& star(int l, int q) const { return base[(l * per + q) * stride]; }
    __device__ __forceinline__ double& flux(int l, int b) const { return base[(l * per + 6 + b) * stride]; }
};

__device__ __forceinline__ double tree_addmags(const TreeStore& S, uint32_t mask, int band, int n_leaves)
{
    double tot = 0.0;
    for (int l = 0; l < n_leaves; ++l)
        if (mask & (1u << l)) tot += S.flux(l, band);
    return -2.5 * log10(tot);
}

__global__ __launch_bounds__(BLOCK, 2) void k_lnpost_tree(const TreeArgs A, int lds_axes_doubles)
{
    extern __shared__ double lds[];
    stage_axes<3>(A.g3.ax, lds);
    stage_axes<4>(A.g4.ax, lds);
    __syncthreads();
    const DevTree& T = *A.T;
    // one sample per lane (no grid-stride loop: see k_lnpost)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < A.n) {
        const double* __restrict__ src = A.pars + i * A.stride_n;
        auto par = [&](int j) { return src[j * A.stride_p]; };
        TreeStore S;
        S.base = lds + lds_axes_doubles + threadIdx.x;
        S.stride = (int)blockDim.x;
        S.per = 6 + T.n_bands;
        // ---- every model star: model-table gather, then magnitudes as fluxes ----
        for (int l = 0; l < T.n_leaves; ++l) {
            const int s = T.leaf_system[l];
            const int base = T.sys_base[s], N = T.n_stars[s];
            const double eep = par(base + T.leaf_slot[l]), age = par(base + N), feh = par(base + N + 1);
            const double dist = par(base + N + 2), AV = par(base + N + 3);
            double v[6];
            Cell3 c3;
            if (locate3(A.g3, lds, age, feh, eep, c3)) {
                gather3<6>(A.g3, c3, v);
            } else {
#pragma unroll
                for (int q = 0; q < 6; ++q) v[q] = d_nan();
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) S.star(l, q) = v[q];
            Cell4 c4;
            const bool ok = locate4(A.g4, lds, v[0], v[1], v[2], AV, c4);
            const double dm = 5 * log10(dist / 10.0);
            for (int b = 0; b < T.n_bands; ++b) {
                const double bc = ok ? gather4_col(A.g4, c4, b) : d_nan();
                S.flux(l, b) = exp10(-0.4 * (v[3] + dm - bc));
            }
        }
        // ---- lnprior (starmodel.py:557-613) ----
        double lnp = 0.0;
        bool dead = false;
        for (int s = 0; s < T.n_systems && !dead; ++s) {
            const int base = T.sys_base[s], N = T.n_stars[s];
            const DevPrior* pri[4] = {&T.prior_age, &T.prior_feh, &T.prior_distance, &T.prior_AV};
            for (int j = 0; j < 4 && !dead; ++j) {
                const double val = par(base + N + j);
                if (val < T.bound_lo[j] || val > T.bound_hi[j]) { dead = true; break; }
                lnp += prior_lnpdf(*pri[j], val);
                if (!isfinite(lnp)) dead = true;
            }
            for (int j = 1; j < N && !dead; ++j)
                if (!(par(base + j) <= par(base + j - 1))) dead = true;
            if (dead) break;
            for (int l = 0; l < T.n_leaves; ++l) {
                if (T.leaf_system[l] != s) continue;
                const double eep = par(base + T.leaf_slot[l]);
                double term;
                if (eep < T.eep_lo || eep > T.eep_hi) {
                    term = -d_inf();
                } else {
                    const double pdf = prior_call(T.prior_mass, S.star(l, 4)) * S.star(l, 5);
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
                double mod = tree_addmags(S, tt.mask, tt.band, T.n_leaves);
                if (tt.relative) {
                    mod -= tree_addmags(S, tt.ref_mask, tt.band, T.n_leaves);
                    mag -= tt.ref_mag;
                }
                const double r = mag - mod;
                lnl += -0.5 * (r * r) / (tt.unc * tt.unc) + T.term_g0[t];
                if (!isfinite(lnl)) bad = true;
            }
            for (int k = 0; k < T.n_spec && !bad; ++k) {
                const iso_tree_prop& sp = T.spec[k];
                const double r = sp.a - S.star(sp.leaf, sp.prop);
                lnl += -0.5 * (r * r) / (sp.b * sp.b) + T.spec_g0[k];
                if (!isfinite(lnl)) bad = true;
            }
            for (int k = 0; k < T.n_limits && !bad; ++k) {
                const iso_tree_prop& lm = T.limits[k];
                const double mod = S.star(lm.leaf, lm.prop);
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
            if (bad) lnl = -d_inf();
        }
        if (A.lnpost) A.lnpost[i] = prior_ok ? lnp + lnl : -d_inf();
        if (A.lnprior) A.lnprior[i] = lnp;
        if (A.lnlike) A.lnlike[i] = lnl;
    }
}
