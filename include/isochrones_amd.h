/*
 * isochrones_amd — C ABI of the MI355X-native isochrones hot path.
 *
 * This is the drop-in boundary.  The reference (timothydmorton/isochrones, pure Python +
 * numba) has no FFI; its de-facto operator boundary is the argument list of the numba
 * functions below, which take nothing but dense float64 arrays, axis vectors and column
 * indices.  Each entry point here states which of those it replaces (paths relative to the
 * reference checkout):
 *
 *   iso_table_*      <- DFInterpolator.grid / .index_columns        isochrones/interp.py:571-614
 *   iso_interp       <- interp_values_{2,3,4}d (+ scalar forms)     isochrones/interp.py:208-392
 *   iso_ic_*         <- ModelGridInterpolator (grid+BC binding)     isochrones/models.py:253-445
 *   iso_interp_mag   <- interp_mags / interp_mag                    isochrones/mags.py:8-124
 *   iso_model_*      <- BasicStarModel.__init__ (obs, priors)       isochrones/starmodel.py:1370-1484
 *   iso_lnpost       <- StarModel.lnpost -> BasicStarModel.lnprior / .lnlike -> star_lnlike
 *                       isochrones/starmodel.py:538-542,1563-1635; isochrones/likelihood.py:16-147;
 *                       isochrones/priors.py (default prior lnpdf's)
 *   iso_unit_cube    <- BasicStarModel.mnest_prior                  isochrones/starmodel.py:1637-1640
 *   iso_eep_table_*, iso_interp_eep <- get_eep / interp_eeps      isochrones/models.py:501-542, interp.py:488-558
 *   iso_tree_*       <- generic StarModel + ObservationTree           isochrones/starmodel.py:544-613, observation.py:464-491,1181-1234
 *   iso_sampler_*    <- emcee.EnsembleSampler driven by lnpost        isochrones/starmodel.py:886-972
 *   iso_catalog_*    <- StarCatalog.iter_models + one lnpost per star  isochrones/catalog.py:126-139
 *
 * Conventions
 *   - All sample buffers (x, pars, outputs) are DEVICE pointers (HIP, the ctx's device); table
 *     and descriptor inputs of the *_create calls are HOST pointers and are copied.
 *   - Everything is float64 (the reference computes in float64; indices are int32/int64).
 *   - Calls are asynchronous on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream).  Tables are immutable after creation; a ctx is bound to one device.
 *   - Return value: ISO_OK or a negative error code; iso_last_error() gives a thread-local
 *     message.  Numeric problems are reported in-band exactly like the reference
 *     (NaN / -inf), never as an error code.
 *   - Strided parameter access: parameter p of sample i is pars[i*stride_n + p*stride_p]
 *     (SoA [n_par][N]: stride_n=1, stride_p=N;  row-major [N][n_par]: stride_n=n_par, stride_p=1).
 */
#ifndef ISOCHRONES_AMD_H
#define ISOCHRONES_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISO_OK            0
#define ISO_ERR_INVALID  -1   /* bad argument / shape */
#define ISO_ERR_HIP      -2   /* HIP runtime failure (message has the hipError string) */
#define ISO_ERR_NOMEM    -3

#define ISO_MAX_DIM       4
#define ISO_MAX_BANDS    32
#define ISO_MAX_STARS     3
#define ISO_MAX_PARAMS    7   /* n_stars + 4 */
#define ISO_MAX_COLS     64   /* columns selectable by one iso_interp call */

/* parametrisations (reference: isochrones/models.py:664-669, 691-696) */
#define ISO_KIND_TRACK    0   /* (mass, eep, feh, distance, AV); grid axes (feh, mass, eep) */
#define ISO_KIND_ISO      1   /* (eep[,eep_1[,eep_2]], age, feh, distance, AV); grid axes (age, feh, eep) */

/* prior families (reference: isochrones/priors.py) */
#define ISO_PRIOR_FLAT       1  /* FlatPrior        :283-293  */
#define ISO_PRIOR_FLATLOG    2  /* FlatLogPrior     :296-306  */
#define ISO_PRIOR_POWERLAW   3  /* PowerLawPrior    :309-342   a = alpha */
#define ISO_PRIOR_GAUSS      4  /* GaussianPrior    :235-257   a = mean, b = sigma, c = lognorm */
#define ISO_PRIOR_LOGNORMAL  5  /* LogNormalPrior   :260-280   a = mu, b = sigma */
#define ISO_PRIOR_CHABRIER   6  /* BrokenPrior(LogNormal, PowerLaw) :143-232,514-519
                                   a = mu, b = sigma, c = alpha, d = breakpoint,
                                   e,f = norms[0..1], g,h = power-law component bounds */
#define ISO_PRIOR_FEH        7  /* FehPrior         :345-381   a = halo_fraction, b = norm, c = local(1/0) */

typedef struct iso_prior {
    int32_t kind;       /* ISO_PRIOR_* */
    int32_t bounded;    /* 1: (lo,hi) are enforced by BoundedPrior.lnpdf/__call__; 0: bounds=None */
    double  lo, hi;     /* Prior.bounds */
    double  a, b, c, d, e, f, g, h;
} iso_prior;

/* One unresolved 1-3 star system: observations + priors.
 * (reference: BasicStarModel.__init__/lnlike/lnprior, isochrones/starmodel.py:1370-1635) */
typedef struct iso_model_desc {
    int32_t n_stars;                       /* 1, 2, 3 */
    int32_t n_bands;
    int32_t bc_cols[ISO_MAX_BANDS];        /* column of each observed band in the BC table */
    double  mag_val[ISO_MAX_BANDS];
    double  mag_unc[ISO_MAX_BANDS];
    double  spec_val[3];                   /* Teff, logg, feh; NaN = not observed */
    double  spec_unc[3];
    int32_t has_parallax;                  /* parallax [mas]; model = 1000/distance */
    int32_t has_numax;
    int32_t has_dnu;                       /* only honoured if has_numax (as the reference) */
    int32_t reserved0;
    double  plx_val, plx_unc;
    double  numax_val, numax_unc;
    double  dnu_val, dnu_unc;              /* reference uses unc := val (starmodel.py:1612); host decides */
    /* priors, one per parameter *name* */
    iso_prior prior_mass, prior_age, prior_feh, prior_distance, prior_AV;
    double  eep_lo, eep_hi;                /* EEP_prior bounds (priors.py:409-421) */
    /* BasicStarModel.bounds(par) per parameter, in param_names order (starmodel.py:1538-1558);
     * what mnest_prior maps the unit cube onto.  Usually equal to the priors' bounds, but the
     * reference lets them diverge (e.g. halo_fraction= replaces the feh prior, :1477-1478). */
    double  bound_lo[ISO_MAX_PARAMS];
    double  bound_hi[ISO_MAX_PARAMS];
} iso_model_desc;

/* ---- observation-tree models ("next" row f4) ---------------------------------------------
 * The reference's generic StarModel keeps an ObservationTree of Python objects
 * (isochrones/observation.py); its likelihood only needs, per observation node, the band, the
 * set of model stars blended into it, the optional reference node (relative photometry) and the
 * measurement.  iso_tree_desc is that flattened form (built on the host by
 * isochrones_amd/observation.py:ObservationTree.program).  Parameters: for every system s (in
 * sorted index order) n_stars[s] EEPs followed by age, feh, distance, AV
 * (observation.py:1116-1148 p2pardict / param_description).  Isochrone parametrisation only
 * (as the reference: starmodel.py:608-609). */
#define ISO_TREE_MAX_SYSTEMS 4
#define ISO_TREE_MAX_LEAVES  8
#define ISO_TREE_MAX_BANDS  16
#define ISO_TREE_MAX_TERMS  64
#define ISO_TREE_MAX_SPEC   24
#define ISO_TREE_MAX_PARAMS (ISO_TREE_MAX_LEAVES + 4 * ISO_TREE_MAX_SYSTEMS)

typedef struct iso_tree_term {      /* one ObsNode (observation.py:464-491) */
    int32_t  band;                  /* index into bc_cols */
    int32_t  relative;              /* 1: compare (node - reference) with (mag - ref_mag) */
    uint32_t mask;                  /* bit l set: leaf l lies below this node */
    uint32_t ref_mask;              /* leaves below the reference node */
    double   mag, unc, ref_mag;
} iso_tree_term;

typedef struct iso_tree_prop {      /* spectroscopy (a = value, b = sigma) or limit (a = min, b = max) */
    int32_t leaf;
    int32_t prop;                   /* 0 Teff, 1 logg, 2 feh */
    double  a, b;
} iso_tree_prop;

typedef struct iso_tree_desc {
    int32_t n_systems, n_leaves, n_bands, n_terms, n_spec, n_limits;
    int32_t n_stars[ISO_TREE_MAX_SYSTEMS];
    int32_t leaf_system[ISO_TREE_MAX_LEAVES];   /* leaves in parameter order */
    int32_t leaf_slot[ISO_TREE_MAX_LEAVES];     /* which EEP of its system */
    int32_t bc_cols[ISO_TREE_MAX_BANDS];
    iso_tree_term terms[ISO_TREE_MAX_TERMS];    /* in the reference's children-first summation order */
    iso_tree_prop spec[ISO_TREE_MAX_SPEC];
    iso_tree_prop limits[ISO_TREE_MAX_SPEC];
    int32_t has_plx[ISO_TREE_MAX_SYSTEMS], has_av[ISO_TREE_MAX_SYSTEMS];
    double  plx_val[ISO_TREE_MAX_SYSTEMS], plx_unc[ISO_TREE_MAX_SYSTEMS];
    double  av_val[ISO_TREE_MAX_SYSTEMS], av_unc[ISO_TREE_MAX_SYSTEMS];
    iso_prior prior_mass, prior_age, prior_feh, prior_distance, prior_AV;
    double  eep_lo, eep_hi;
    double  bound_lo[4], bound_hi[4];           /* StarModel.bounds of age, feh, distance, AV (starmodel.py:563-566) */
} iso_tree_desc;

typedef struct iso_ctx   iso_ctx;
typedef struct iso_table iso_table;   /* dense N-D table + axes, resident in HBM */
typedef struct iso_ic    iso_ic;      /* model table + BC table + column binding */
typedef struct iso_model iso_model;   /* iso_ic + iso_model_desc */
typedef struct iso_catalog iso_catalog; /* iso_ic + many iso_model_desc sharing bands/multiplicity */

const char* iso_last_error(void);
const char* iso_version(void);

int  iso_ctx_create(iso_ctx** out, int device);
void iso_ctx_destroy(iso_ctx* ctx);

/* shape[ndim+1] = (n_0..n_{ndim-1}, n_col); grid C-contiguous, last axis = column
 * (the layout of DFInterpolator.grid, isochrones/interp.py:607-609); axes[d] has shape[d]
 * strictly increasing values.  ndim in {2,3,4}. */
int  iso_table_create(iso_ctx* ctx, int ndim, const int64_t* shape, const double* grid,
                      const double* const* axes, iso_table** out);
/* The same for a table whose VALUES are already in this device's memory (d_grid: [n0]..[n_columns] float64, device
 * pointer; axes on the host): one device-to-device copy.  What a rank that received the tables through an RCCL broadcast
 * calls (isochrones_amd.catalog.broadcast_interpolator) instead of downloading them to a host array and uploading them
 * again.  The caller keeps ownership of d_grid. */
int  iso_table_create_from_device(iso_ctx* ctx, int ndim, const int64_t* shape, const double* d_grid, const double* const* axes,
                                  iso_table** out);
void iso_table_destroy(iso_table* t);

/* out[i*k + c] = multilinear interpolation of column icols[c] at (x[0][i],..,x[ndim-1][i]).
 * x = HOST array of ndim DEVICE pointers; icols = HOST array.  Replaces interp_values_2d/3d/4d and the
 * scalar interp_value_*d (isochrones/interp.py:208-392); 3-D tables switch to a [cell][column][corner] pack for
 * batches of >= 32768 rows. */
int  iso_interp(iso_table* t, const double* const* x, int64_t n, const int32_t* icols, int k,
                double* out, void* stream);

/* Bind a 3-D model table and a 4-D BC table.  cols[0..3] = (Teff, logg, feh, Mbol) columns of
 * the model table (models.py:416-428); prior_cols[0..1] = (age, dt_deep) for tracks or
 * (mass, dm_deep) for isochrones (priors.py:423-429), or -1,-1 if the table lacks them;
 * astero_cols[0..1] = (nu_max, delta_nu) or -1,-1.  Builds the packed hot-column table. */
int  iso_ic_create(iso_ctx* ctx, iso_table* model_grid, iso_table* bc_grid, int kind,
                   const int32_t cols[4], const int32_t prior_cols[2], const int32_t astero_cols[2],
                   iso_ic** out);
void iso_ic_destroy(iso_ic* ic);

/* interp_mags / interp_mag (isochrones/mags.py:8-124): pars = 5 parameters per sample in the ic's parametrisation
 * (strided, see top).
 * Teff/logg/feh [n], mags [n, nb] row-major; any output pointer may be NULL. */
int  iso_interp_mag(iso_ic* ic, const double* pars, int64_t stride_n, int64_t stride_p, int64_t n,
                    const int32_t* bc_cols, int nb,
                    double* Teff, double* logg, double* feh, double* mags, void* stream);

/* One 1-3 star system: observations + prior constants (BasicStarModel.__init__, isochrones/starmodel.py:1370-1484;
 * the descriptor is what isochrones_amd/starmodel.py:model_desc packs). */
int  iso_model_create(iso_ic* ic, const iso_model_desc* desc, iso_model** out);
void iso_model_destroy(iso_model* m);
int  iso_model_n_params(const iso_model* m);
/* Which kernel family evaluates this model (measurement / test helper): the generic kernel (any table shape, any
 * band count) or the fused kernel on the corner-packed tables.  (ISO_PATH_FUSED_COMPACT, the fused kernel on the
 * compact tables, is no longer produced: since round 4 the fused kernels read the corner-packed tables only and a
 * model without them - ISOCHRONES_AMD_PATH=compact, a pack that did not fit - runs the generic kernel.) */
#define ISO_PATH_GENERIC        0
#define ISO_PATH_FUSED_COMPACT  1
#define ISO_PATH_FUSED_PACKED   2
int  iso_model_kernel_path(const iso_model* m);
/* Test hooks: which kernel instantiations the library's launchers choose.  iso_debug_trace_kernels(1) clears the calling
 * thread's list and turns recording on (0: off; returns the previous state); every launch made by a thread is then noted
 * once per distinct instantiation, spelled as c++filt spells the kernel's symbol ("k_lnpost_fast<0, 1, 1, false, false>").
 * iso_debug_kernels copies the calling thread's list, one name per line, into buf (NUL-terminated, truncated to `size`)
 * and returns the number of bytes the whole list needs.  tests/test_gpu_dispatch_table.py ticks every kernel the build
 * compiled off against the oracle with these.  Recording costs one atomic load per launch when it is off. */
int     iso_debug_trace_kernels(int on);
int64_t iso_debug_kernels(char* buf, int64_t size);
/* What the calling thread's last iso_sampler_run decided for the BasicStarModel / catalog kernels: out8 = {persistent form
 * (1) or one launch per half-step (0), register-capped instantiation, threads per workgroup (256, or 192 for ensembles of 129-192
 * moves per half-step), default prior families compiled in, ensembles per workgroup, workgroups per CU by the occupancy
 * query, workgroups of the launch, 0}.  tests assert the launch shape of the reference-shape catalog with it. */
int     iso_debug_sampler_plan(int32_t* out8);

/* Host-side test helper, no device involved: the bracket index the fused kernels compute for every x[k] on axis
 * `which` of a set of axes - bucket table (planned for all `n_axes` axes together within `budget` bytes, exactly as
 * the library stages them in LDS) + windowed bisection.  It must equal what the reference's searchsorted /
 * find_indices give (interp.py:10-35,116-123): #{a_j <= x} - 1, clamped to [0, n - 2], for a0 <= x <= a_last.
 * plan_out (optional, 5 ints per axis): buckets, window, levels, shift, first bucket.  Returns 0 or an ISO_ERR code. */
int  iso_axis_bracket_host(const double* const* axes, const int32_t* n_nodes, int n_axes, int budget, int which,
                           const double* x, int64_t n, int32_t* index_out, int32_t* plan_out);

/* lnpost = lnprior + lnlike, or -inf where lnprior is not finite (starmodel.py:538-542).
 * lnprior_out / lnlike_out are optional (NULL to skip); when requested they hold the values
 * BasicStarModel.lnprior / .lnlike would return for every sample (lnlike is then evaluated even
 * where the prior is -inf). */
int  iso_lnpost(iso_model* m, const double* pars, int64_t stride_n, int64_t stride_p, int64_t n,
                double* lnpost_out, double* lnprior_out, double* lnlike_out, void* stream);

/* The same for HOST arrays (pars [n][n_params] row-major): the sampler-callback form — what emcee /
 * MultiNest do when they call StarModel.lnpost(p) with one parameter vector (starmodel.py:952,966,
 * 1642-1645) — and the vectorised form for host batches of any size.  Up to 32768 rows travel through a pinned,
 * device-mapped staging buffer owned by the model, one launch per 8192 rows; a call that fits one workgroup (<= 256
 * rows on the fused kernel) is completed by a flag the kernel raises in mapped memory, which the host spins on
 * (ISOCHRONES_AMD_HOST_SYNC=1: synchronise the stream instead).  Larger batches are cut into 131072-row chunks: the
 * calling thread uploads and launches chunk k (results land in pinned host memory) while a helper thread copies the
 * results of the chunks before it into the caller's arrays.
 * Calls on one model are serialised by the library (the staging areas are the model's); different models run
 * concurrently. */
int  iso_lnpost_host(iso_model* m, const double* pars, int64_t n, double* lnpost_out, double* lnprior_out,
                     double* lnlike_out);

/* mnest_prior: cube[i,p] <- lo_p + (hi_p - lo_p) * cube[i,p], in place (starmodel.py:1637-1640). */
int  iso_unit_cube(iso_model* m, double* cube, int64_t stride_n, int64_t stride_p, int64_t n, void* stream);

/* (mass, age, feh) -> EEP on the ragged per-track age arrays: the non-"accurate" get_eep of the
 * reference (isochrones/models.py:501-542 -> interp_eeps, isochrones/interp.py:488-558).
 * ages[n0*n1][n_eep] = log10 age along every (feh, mass) track (HOST, copied), lengths[n0*n1] =
 * populated points per track, eep0 = EEP of array index 0 (1 for MIST).  x = age, x0 = feh,
 * x1 = mass (DEVICE arrays). */
typedef struct iso_eep_table iso_eep_table;
int  iso_eep_table_create(iso_ctx* ctx, const double* ages, const int64_t* lengths, const double* ax0, int64_t n0,
                          const double* ax1, int64_t n1, int64_t n_eep, double eep0, iso_eep_table** out);
void iso_eep_table_destroy(iso_eep_table* t);
int  iso_interp_eep(iso_eep_table* t, const double* x, const double* x0, const double* x1, int64_t n, double* out,
                    void* stream);

/* HOST-array forms of the three interpolation entry points for scalar calls and small batches — how the reference's
 * API is used interactively (mist.interp_value(pars, props), mist.interp_mag(pars, bands), mist.get_eep(m, a, f):
 * isochrones/models.py:390-445, 501-542).  Inputs and outputs are host arrays; they travel through a pinned,
 * device-mapped staging buffer owned by the context: one launch + one synchronise per call (per 1 MiB chunk).
 * x [n][ndim] and pars [n][5] row-major; out [n][k]; mags [n][nb]; any of Teff / logg / feh / mags may be NULL. */
int  iso_interp_host(iso_table* t, const double* x, int64_t n, const int32_t* icols, int k, double* out);
int  iso_interp_mag_host(iso_ic* ic, const double* pars, int64_t n, const int32_t* bc_cols, int nb, double* Teff,
                         double* logg, double* feh, double* mags);
int  iso_interp_eep_host(iso_eep_table* t, const double* age, const double* feh, const double* mass, int64_t n,
                         double* out);

/* A catalog = many independent systems observed in the same bands with the same multiplicity
 * (reference: isochrones/catalog.py:19-139 StarCatalog.iter_models; scripts/batch_starfit shards
 * them over processes).  iso_catalog_lnpost evaluates a batch of rows where row i belongs to
 * star star_id[i] (DEVICE int32 array): S stars x W walkers in one launch.  Needs 1-12 bands. */
int  iso_catalog_create(iso_ic* ic, const iso_model_desc* descs, int64_t n_models, iso_catalog** out);
/* The same catalog from one template descriptor (priors, bands, flags, bounds: identical for every star) plus
 * plain HOST columns of what differs per star: magnitudes [n][n_bands] (NaN value = band not observed),
 * spectroscopic values [n][3] (NaN = absent), parallaxes (has_plx[n] 0/1) and the upper distance bound [n]
 * (NULL = the template's).  The per-star constant blocks are filled by a kernel on the device, so the
 * host moves ~20 doubles per star instead of building n descriptors. */
int  iso_catalog_create_columns(iso_ic* ic, const iso_model_desc* tmpl, int64_t n_models, const double* mag_val,
                                const double* mag_unc, const double* spec_val, const double* spec_unc,
                                const int32_t* has_plx, const double* plx_val, const double* plx_unc,
                                const double* dist_hi, iso_catalog** out);
void iso_catalog_destroy(iso_catalog* c);
int  iso_catalog_lnpost(iso_catalog* c, const int32_t* star_id, const double* pars, int64_t stride_n,
                        int64_t stride_p, int64_t n, double* lnpost_out, void* stream);

/* Start points of a catalog fit, found on the device (one workgroup per star; fast/start_points.h): every star draws
 * oversample x nwalkers candidates inside its parameter bounds - log-uniform in mass, distance within 4 sigma of the
 * parallax distance where the star has a positive parallax, the EEPs of a multiple system in descending order -
 * evaluates them with the catalog kernels' lnpost and keeps its best nwalkers; while fewer than nwalkers are finite it
 * draws again, up to max_tries times as many.  Replaces the reference's per-star `sample_from_prior` loop
 * (starmodel.py:903-949: one Python lnpost call per draw until nwalkers rows are valid) for S stars at once.
 * best [S][nwalkers][n_params], best_lnp [S][nwalkers] (descending per star), failed [S] (1 = fewer than nwalkers finite
 * candidates: that star's rows are NaN) are DEVICE arrays.  Random numbers: Philox4x32-10, counter
 * (4 chunk + call, star, lane, 0x57), key = seed - a given (seed, catalog) gives the same start points on every run.
 * nwalkers <= 256. */
int  iso_catalog_start_points(iso_catalog* c, int nwalkers, int oversample, int max_tries, uint64_t seed, double* best,
                              double* best_lnp, int32_t* failed, void* stream);

/* A catalog fit keeps its batch rectangular without a host round trip: every star with failed[s] != 0 (its start-point search
 * found fewer than nwalkers finite candidates) gets the walkers of the batch's first good star and lnpost 0 - on its own
 * posterior those walkers never move, and the caller blanks its result row (the reference isolates a failing star with
 * try / except around its fit, isochrones/starfit.py:155-159).  pos [S][nwalkers][n_params], lnp [S][nwalkers], failed [S]: DEVICE
 * arrays as iso_catalog_start_points wrote them.  With no good star at all the positions stay NaN. */
int  iso_catalog_patch_failed(iso_catalog* c, int nwalkers, double* pos, double* lnp, const int32_t* failed, void* stream);

/* Generic (observation-tree) StarModel: lnpost / lnprior / lnlike of starmodel.py:538-613 +
 * observation.py:1181-1234.  Outputs as iso_lnpost; lnlike is -inf (never NaN) when not finite,
 * as the reference. */
typedef struct iso_tree_model iso_tree_model;
int  iso_tree_model_create(iso_ic* ic, const iso_tree_desc* desc, iso_tree_model** out);
void iso_tree_model_destroy(iso_tree_model* m);
int  iso_tree_lnpost(iso_tree_model* m, const double* pars, int64_t stride_n, int64_t stride_p, int64_t n,
                     double* lnpost_out, double* lnprior_out, double* lnlike_out, void* stream);
/* The same for HOST arrays (pars [n][n_params] row-major), as iso_lnpost_host: the sampler-callback form. */
int  iso_tree_lnpost_host(iso_tree_model* m, const double* pars, int64_t n, double* lnpost_out, double* lnprior_out,
                          double* lnlike_out);

/* Device-resident affine-invariant ensemble sampler (stretch move, Goodman & Weare 2010) — what the
 * reference obtains from emcee.EnsembleSampler(nwalkers, npars, self.lnpost).run_mcmc(...)
 * (isochrones/starmodel.py:951-969), with the proposal, the fused lnpost and the accept step in one
 * kernel: either one launch per half-ensemble step, or — whenever an ensemble fits a workgroup's LDS — a single
 * persistent launch for all nsteps
 * iterations whose workgroups own their ensembles (same moves, same Philox numbers, identical
 * chains; environment ISOCHRONES_AMD_SAMPLER=auto|persistent|stepwise overrides the choice).  pos [n_ens*W, n_params] row-major and lnp [n_ens*W] are DEVICE arrays
 * updated in place (lnp must hold lnpost(pos) on entry); chain [nsteps][n_ens*W][n_params],
 * chain_lnp [nsteps][n_ens*W] and accepted [n_ens*W] (int32 counters) are optional DEVICE outputs.
 * n_ens = 1 for a model, = n_models for a catalog (row = star*W + walker).  The model / catalog
 * handle must outlive the sampler. */
typedef struct iso_sampler iso_sampler;
/* layouts of a stored chain */
#define ISO_CHAIN_ROW_MAJOR    0   /* chain [nsteps][n_ens*W][n_params] (the default; emcee's walker-major rows) */
#define ISO_CHAIN_PARAM_MAJOR  1   /* chain [nsteps][n_params][n_ens*W]: consecutive walkers store consecutive doubles, and
                                      the nsteps*W values of an (ensemble, parameter) pair are W contiguous doubles per
                                      step - the summaries below then fetch every line of the chain once */
int  iso_sampler_create_model(iso_model* m, int nwalkers, double a, uint64_t seed, iso_sampler** out);
/* n_ensembles independent ensembles of ONE model advanced in lock-step by the same launches (row = ensemble * W + walker
 * keys the random numbers, so ensemble e of such a sampler makes the moves star e of a catalog sampler would): a single
 * star's fit occupies one workgroup of the chip, so further chains of it - for convergence diagnostics across
 * independently started ensembles, or simply more samples - cost no extra time until the workgroups fill the chip. */
int  iso_sampler_create_model_ensembles(iso_model* m, int64_t n_ensembles, int nwalkers, double a, uint64_t seed,
                                        iso_sampler** out);
int  iso_sampler_create_catalog(iso_catalog* c, int nwalkers, double a, uint64_t seed, iso_sampler** out);
/* The same sampler for the model classes that are not a BasicStarModel with at most 12 bands - the reference fits every
 * StarModel through the one fit_mcmc (isochrones/starmodel.py:886-972):
 *   iso_sampler_create_model     also takes models with 13-32 bands (band-tiled evaluation);
 *   iso_sampler_create_tree      an observation tree (likelihood of observation.py:1181-1234; n_params = sum over systems
 *                                of (stars + 4), rows [n_ens*W][n_params]);
 *   iso_sampler_create_isotrack  the reference's IsoTrackModel (starmodel.py:2010-2104): parameters (eep, mass, age, feh,
 *                                distance, AV); iso_m / track_m are single-star models with the same bands on the
 *                                isochrone / evolution-track grid (track_m owns the priors and the parallax term), the age
 *                                prior is ln p(age) = age_lnorm + age ln 10 inside [age_lo, age_hi], -inf outside.
 * All three run every iteration of iso_sampler_run in ONE persistent launch, one workgroup per ensemble, positions in LDS
 * (a 300-walker ensemble of a 10-parameter tree takes 77 KB of the CU's 160 KB); same Philox stream and move arithmetic as
 * the kernels above.  n_ensembles independent ensembles as iso_sampler_create_model_ensembles.  ISO_ERR_INVALID when the
 * shape has no kernel (a tree off the corner-packed path or with more than 12 bands) or an ensemble does not fit a CU's LDS. */
int  iso_sampler_create_tree(iso_tree_model* m, int64_t n_ensembles, int nwalkers, double a, uint64_t seed, iso_sampler** out);
int  iso_sampler_create_isotrack(iso_model* iso_m, iso_model* track_m, double age_lo, double age_hi, double age_lnorm,
                                 int64_t n_ensembles, int nwalkers, double a, uint64_t seed, iso_sampler** out);
void iso_sampler_destroy(iso_sampler* s);
/* How iso_sampler_run lays out its `chain` output from now on (chain_lnp is [nsteps][n_ens*W] either way). */
int  iso_sampler_set_chain_layout(iso_sampler* s, int layout);
int  iso_sampler_run(iso_sampler* s, double* pos, double* lnp, int nsteps, double* chain, double* chain_lnp,
                     int32_t* accepted, void* stream);

/* Per-ensemble quantiles of a stored chain, on the device: the posterior summaries a catalog fit reports per
 * star (the reference takes them from the pandas samples of every star, isochrones/starfit.py + catalog
 * drivers).  chain is iso_sampler_run's chain output [nsteps][n_ens*W][n_params]; for every ensemble e and
 * parameter d the order statistics of the nsteps*W values are selected (one wavefront per pair with the values
 * in registers; a workgroup selection / LDS sort for more than 6656 values or heavy ties) and
 * out[(e*n_params + d)*nq + k] receives the q[k] quantile with linear interpolation between order statistics
 * (numpy.percentile's default, bit for bit).  q is a HOST array of nq <= 8 levels in [0, 1].  More than 8192 values per
 * pair (the reference's default fit keeps 300 walkers x 100 iterations) are selected by refinement passes streamed from
 * the chain - any length below 2^31.
 * ISOCHRONES_AMD_QUANTILES=workgroup|sort forces the older forms (tests, A/B runs). */
int  iso_chain_quantiles(iso_ctx* ctx, const double* chain, int64_t nsteps, int64_t n_ens, int W, int n_params,
                         const double* q, int nq, double* out, void* stream);
/* The same for a chain stored in `layout` (ISO_CHAIN_ROW_MAJOR = the call above). */
int  iso_chain_quantiles_layout(iso_ctx* ctx, const double* chain, int layout, int64_t nsteps, int64_t n_ens, int W,
                                int n_params, const double* q, int nq, double* out, void* stream);

/* Time `reps` back-to-back iso_lnpost launches with hipEvents on `stream`; returns the mean
 * milliseconds per launch in *ms_per_launch (measurement helper for bench.py). */
int  iso_time_lnpost(iso_model* m, const double* pars, int64_t stride_n, int64_t stride_p, int64_t n,
                     double* lnpost_out, int reps, void* stream, double* ms_per_launch);
/* The same over a rotation of `n_batches` distinct sample batches: launch r evaluates pars[r % n_batches] into
 * lnpost_out[r % n_batches] (HOST arrays of DEVICE pointers, every batch n rows with the same strides).  With enough
 * batches the table lines one launch touches have left the 256 MiB Infinity Cache before any launch needs them again,
 * so the measured time has no inter-launch reuse in it. */
int  iso_time_lnpost_rotating(iso_model* m, const double* const* pars, double* const* lnpost_out, int n_batches,
                              int64_t stride_n, int64_t stride_p, int64_t n, int reps, void* stream,
                              double* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* ISOCHRONES_AMD_H */
