// Dispatcher of the fast fused lnpost kernels (instantiated in iso_fast_{track1,iso1,iso2,iso3}.hip).
#include "iso_fast_kernel.h"

namespace iso {

bool launch_lnpost_fast(int kind, int n_stars, int n_bands, bool multi, const FastArgs& A, hipStream_t s)
{
    if (!A.hotq || (!A.bcq && n_bands > 0)) return false;      // no corner-packed tables: the generic kernel's business
    if (n_bands > fastk::FAST_MAX_NB)        // 13-32 bands: band-tiled batch kernels (iso_fast_wide.hip)
        return !multi && !A.astq && launch_fast_wide(kind, n_stars, A, s);
    if (kind == ISO_KIND_TRACK) return n_stars == 1 && launch_fast_track1(n_bands, multi, A, s);
    switch (n_stars) {
    case 1: return launch_fast_iso1(n_bands, multi, A, s);
    case 2: return launch_fast_iso2(n_bands, multi, A, s);
    case 3: return launch_fast_iso3(n_bands, multi, A, s);
    }
    return false;
}

bool launch_stretch(int kind, int n_stars, int n_bands, const FastArgs& A, const StretchArgs& S, hipStream_t s)
{
    if (A.astq) {                     // the model has nu_max / delta_nu terms: ASTERO instantiations
        if (kind == ISO_KIND_TRACK) return n_stars == 1 && launch_stretch_ast_track1(n_bands, A, S, s);
        switch (n_stars) {
        case 1: return launch_stretch_ast_iso1(n_bands, A, S, s);
        case 2: return launch_stretch_ast_iso2(n_bands, A, S, s);
        case 3: return launch_stretch_ast_iso3(n_bands, A, S, s);
        }
        return false;
    }
    if (kind == ISO_KIND_TRACK) return n_stars == 1 && launch_stretch_track1(n_bands, A, S, s);
    switch (n_stars) {
    case 1: return launch_stretch_iso1(n_bands, A, S, s);
    case 2: return launch_stretch_iso2(n_bands, A, S, s);
    case 3: return launch_stretch_iso3(n_bands, A, S, s);
    }
    return false;
}

// start points of a catalog (fast/start_points.h); W walkers per star, oversample * W candidates at least
int launch_catalog_start(int kind, int n_stars, int n_bands, const FastArgs& A, double* best, double* best_lnp, int32_t* failed,
                         int64_t n_models, int W, int oversample, int max_tries, uint64_t seed, hipStream_t s)
{
    fastk::StartArgs T;
    T.best = best;
    T.best_lnp = best_lnp;
    T.failed = failed;
    T.n_stars = n_models;
    T.W = W;
    T.chunks_min = std::max(1, (oversample * W + BLOCK - 1) / BLOCK);
    T.chunks_max = std::max(T.chunks_min, T.chunks_min * std::max(1, max_tries));
    T.seed = seed;
    if (W < 1 || W > fastk::START_MAX_W) return 0;
    bool ok = false;
    if (kind == ISO_KIND_TRACK) ok = n_stars == 1 && launch_start_track1(n_bands, A, T, s);
    else if (n_stars == 1) ok = launch_start_iso1(n_bands, A, T, s);
    else if (n_stars == 2) ok = launch_start_iso2(n_bands, A, T, s);
    else if (n_stars == 3) ok = launch_start_iso3(n_bands, A, T, s);
    return ok ? 1 : 0;
}

size_t stretch_persist_lds(int n_bands, int axes_len, int W, int n_params, int* ensembles_per_workgroup)
{
    if (ensembles_per_workgroup) *ensembles_per_workgroup = fastk::persist_group(W);
    return fastk::stretch_persist_lds_bytes(axes_len, n_bands, W, n_params);
}

}  // namespace iso
