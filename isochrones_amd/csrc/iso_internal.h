// Internal definitions shared by the translation units of libiso_hip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <memory>
#include <mutex>
#include <vector>
#include <atomic>
#include <thread>

#include "../../include/isochrones_amd.h"

namespace iso {

int fail(int code, const std::string& msg);   // sets the thread-local error string

constexpr int BLOCK = 256;
constexpr int MAX_LDS_AXIS_DOUBLES = 6144;   // 48 KiB of staged axes at most
constexpr int HOT_COLS = 8;                  // Teff logg feh Mbol prior_val prior_deriv nu_max delta_nu
constexpr int PACK_COLS = 6;                 // columns kept in the corner-packed table
constexpr int PACK_ENTRY = 8 * PACK_COLS;    // doubles per corner-packed cell (8 corners x 6 columns = 384 B)

struct AxisD {
    const double* g;   // device copy
    int n;
    int lds_off;       // offset (doubles) into the workgroup's LDS staging area, -1 = not staged
    int uniform;       // 1: a_i == a0 + i*step exactly (with and without FMA)
    double a0, step;
};

struct Grid3V {           // packed hot-column model table
    AxisD ax[3];
    const double* hot;    // [n0][n1][n2][HOT_COLS]
    const double* hotq;   // corner-packed [cell][8 corners x PACK_COLS] (k_pack_corners order), or null
    const double* astq;   // corner-packed (nu_max, delta_nu) [cell][8 corners x 2], or null
    int64_t s0, s1;       // cell strides of axes 0, 1 (axis 2 stride = 1)
};

struct Grid4V {           // BC table (all columns, or packed to the model's bands)
    AxisD ax[4];
    const double* tab;    // [nT][ng][nf][nA][ncol]
    const double* tabq;   // corner-packed [cell][16 corners x ncol] (k_pack_corners order) of the same columns, or null
    int ncol;
    int64_t s0, s1, s2;   // cell strides of axes 0..2 (axis 3 stride = 1)
};

// device-side prior record with everything constant pre-evaluated on the host
struct DevPrior {
    int kind, bounded;
    double lo, hi;
    double a, b, c, d, e, f, g, h;
    double k0, k1, k2, k3, k4, k5;
    double r0, r1, r2;    // reciprocals / log-constants used by the fast kernels
};

struct DevModel {
    int n_stars, n_bands, kind;
    int has_parallax, has_numax, has_dnu;
    double mag_val[ISO_MAX_BANDS];
    double mag_g0[ISO_MAX_BANDS];    // log(1/sqrt(2 pi)) + log(unc)
    double mag_unc2[ISO_MAX_BANDS];  // unc*unc
    double mag_hinv[ISO_MAX_BANDS];  // 0.5/(unc*unc)
    double spec_val[3], spec_g0[3], spec_unc2[3], spec_hinv[3];
    double plx_val, plx_g0, plx_unc2, plx_hinv;
    double numax_val, numax_g0, numax_unc2, numax_hinv;
    double dnu_val, dnu_g0, dnu_unc2, dnu_hinv;
    DevPrior prior_mass, prior_age, prior_feh, prior_distance, prior_AV;
    double eep_lo, eep_hi;
    double bound_lo[ISO_MAX_PARAMS], bound_hi[ISO_MAX_PARAMS];
};

// ---- fast-path descriptors ------------------------------------------------------------------
struct FastAxis {
    int off;   // LDS offset (doubles): values at [off, off+n), reciprocal spacings at [off+n, off+2n-1)
    int n;
    // bucket table (fast/axis_lut.h): the byte at blob byte offset lut + clamp((hi32(x + c) >> sh) - b0, 0, nbk - 1) is a
    // node at or below every x of that bucket, and the bracket lies within the `win` nodes from there (one bucket,
    // win = n: the plain bisection).  c is a double whose low word is zero (chi = its high word); six scalar registers
    // per axis.
    int lut;   // byte offset of the table inside the staged blob
    int shw;   // sh | win << 5 | (nbk - 1) << 17   (sh < 32, win < 4096, nbk <= 32768)
    int chi;
    int b0;
};

struct FastArgs {
    FastAxis m0, m1;                 // model axes 0 and 1 (bisection in LDS)
    double e_a0, e_step, e_inv;      // model axis 2 = EEP: exactly uniform -> O(1) index (e_axis == null)
    double e_last;                   // its last node (upper bound of the table)
    int e_n;
    FastAxis ec;                     // ... or not uniform: every 8th node (+ the last one) staged in LDS
    const double* e_axis;            //     and the axis itself on the device (null for a uniform axis)
    FastAxis b0, b1, b2, b3;         // BC axes
    const double* axes_blob;         // [values | 1/spacing] of the six LDS axes, concatenated
    int axes_len;                    // doubles
    const double* hot;               // compact hot table [n0][n1][n2][HOT_COLS]
    const double* hotq;              // corner-packed [n0][n1][n2][8 corners][PACK_COLS] (or null)
    const double* astq;              // corner-packed (nu_max, delta_nu) [cell][8 corners][2], null unless the
                                     // model has asteroseismic terms
    int64_t s0, s1;
    const double* bc;                // BC restricted to the model's bands [..][nb]
    const double* bcq;               // corner-packed BC [cells][16 corners][nb] (or null)
    int nb_total;                    // bands in a cell of bcq (the band-tiled kernels, 13-32 bands, read it)
    int64_t bs0, bs1, bs2;
    const DevModel* m;               // one model, or an array indexed by star_id (catalog kernels)
    int shared_priors;               // catalog kernels: every star's mass / age / [Fe/H] / A_V priors and EEP bounds equal
                                     // those of m[0] (only the distance prior is per star) - they are then read from m[0],
                                     // one cached block for the whole launch instead of five lines per star and half-step
    const int32_t* star_id;          // per-row model index (catalog kernels only)
    const double* pars;
    int64_t stride_n, stride_p, n;
    double* lnpost;
    double* lnprior;                 // optional: BasicStarModel.lnprior / .lnlike of every sample
    double* lnlike;                  // (lnlike is then evaluated even where the prior is not finite)
    // host-callback launches of a single workgroup (iso_lnpost_host): after its results are out the kernel
    // stores done_seq to done_flag (pinned, device-mapped) - the host spins on it instead of paying a
    // hipStreamSynchronize (13 -> 8 us per round trip, tools/sync_probe.hip); null for every other launch
    unsigned long long* done_flag;
    unsigned long long done_seq;
};

// flattened observation tree of a generic StarModel (constants pre-evaluated on the host)
// one photometric term of an observation tree as the fused evaluation reads it (constants folded on the host)
struct DevTreeTerm {
    uint32_t mask, ref_mask;   // leaves below the node / below its reference node
    int32_t relative, pad_;
    double dmag;               // observed magnitude (relative: minus the reference's)
    double g0, hinv;           // log(1/sqrt(2 pi)) + log(unc), 0.5 / unc^2
};

struct DevTree {
    int n_systems, n_leaves, n_bands, n_terms, n_spec, n_limits, n_params;
    int n_stars[ISO_TREE_MAX_SYSTEMS], sys_base[ISO_TREE_MAX_SYSTEMS];
    int leaf_system[ISO_TREE_MAX_LEAVES], leaf_slot[ISO_TREE_MAX_LEAVES];
    iso_tree_term terms[ISO_TREE_MAX_TERMS];
    double term_g0[ISO_TREE_MAX_TERMS];          // log(1/sqrt(2 pi)) + log(unc)
    double term_hinv[ISO_TREE_MAX_TERMS];        // 0.5 / unc^2 (the fused tree evaluation multiplies; the generic kernel divides)
    iso_tree_prop spec[ISO_TREE_MAX_SPEC], limits[ISO_TREE_MAX_SPEC];
    double spec_g0[ISO_TREE_MAX_SPEC], spec_hinv[ISO_TREE_MAX_SPEC];
    int has_plx[ISO_TREE_MAX_SYSTEMS], has_av[ISO_TREE_MAX_SYSTEMS];
    double plx_val[ISO_TREE_MAX_SYSTEMS], plx_unc[ISO_TREE_MAX_SYSTEMS], plx_g0[ISO_TREE_MAX_SYSTEMS], plx_hinv[ISO_TREE_MAX_SYSTEMS];
    double av_val[ISO_TREE_MAX_SYSTEMS], av_unc[ISO_TREE_MAX_SYSTEMS], av_g0[ISO_TREE_MAX_SYSTEMS], av_hinv[ISO_TREE_MAX_SYSTEMS];
    DevPrior prior_mass, prior_age, prior_feh, prior_distance, prior_AV;
    double eep_lo, eep_hi;
    double bound_lo[4], bound_hi[4];
    int std_priors;      // the five prior records are the reference's default families (fast/tree_eval.h takes them as constants)
    // the photometric terms once more, BAND-MAJOR (stable: the reference's order inside a band), for the register form of the
    // fused evaluation (fast/tree_eval.h): the band is then a compile-time loop index there and a node's flux sum selects
    // over leaves only; bterm_first[b] .. bterm_first[b + 1] - 1 are band b's terms
    int bterm_first[ISO_TREE_MAX_BANDS + 1];
    DevTreeTerm bterms[ISO_TREE_MAX_TERMS];
};

struct StretchArgs {
    double* pos;          // [n_rows][n_params] row-major, n_rows = n_stars_in_batch * W
    double* lnp;          // [n_rows]
    int32_t* accepted;    // [n_rows] acceptance counters, may be null
    int W;                // walkers per star (even)
    int half;             // 0: update walkers [0, W/2) against [W/2, W); 1: the other way round
    int multi;            // 1: A.m is an array indexed by the star of the row
    int64_t n_active;     // n_stars_in_batch * W/2
    double a;             // stretch scale
    uint64_t seed;
    uint32_t step;
    int nsteps;           // 0: step-wise kernel (one half-step per launch); > 0: persistent kernel, all
                          // iterations in one launch (chain slabs then advance by n_rows per iteration)
    int64_t chain_rs, chain_ps;   // element strides of the stored chain between rows / between parameters of a step
    double* chain_pos;    // optional: this step's [n_rows][n_params] (or [n_params][n_rows]) slab of the stored chain
    double* chain_lnp;    // optional: this step's [n_rows] slab
    int* occupancy_query; // host side only, persistent form: non-null = report workgroups/CU, do not launch
    int dense;            // host side only, persistent form: 1 = the register-capped (3 waves/SIMD) instantiation
    int std_priors;       // host side only: the single model's priors are the reference's default families
    int group;            // persistent form: ensembles per workgroup (0 or >= the most a workgroup holds: that many);
                          // fewer spread a small catalog over more CUs - LDS is laid out for the maximum either way
    int pair;             // host side only: a single binary may take the one-star-per-lane kernel (k_stretch_pair)
    int threads;          // host side only, persistent register-capped form: threads per workgroup (0 = BLOCK; 192 for ensembles of
                          // 129 ... 192 moves per half-step, fast/sampler.h)
    int dense_stdp;       // host side only, persistent register-capped form: 1 = the instantiation with the default prior families compiled in
    int triple_moves;     // host side only: a single triple takes the one-star-per-row kernel (k_stretch_triple) while a workgroup's
                          // half-step has at most this many moves (0: never)
};

// the moves' side of a run of the any-model persistent sampler (fast/sampler_any.h)
struct AnyStretchArgs {
    double* pos;          // [n_ens * W][NP] row-major
    double* lnp;          // [n_ens * W]
    int32_t* accepted;    // [n_ens * W] acceptance counters, may be null
    int W, NP;
    int lanes;            // lanes of a workgroup that take moves (64, 128, 192 or 256)
    int own_off;          // doubles of LDS in front of the sampler's own arrays (the evaluator's: axes, gather slots, leaf values)
    int64_t n_ens;
    double a;
    uint64_t seed;
    uint32_t step;
    int nsteps;
    int64_t chain_rs, chain_ps;   // as StretchArgs: element strides of a stored step between rows / between parameters
    double* chain_pos;    // optional [nsteps] slabs of n_ens * W * NP
    double* chain_lnp;    // optional [nsteps][n_ens * W]
};

// the per-point callback's mailbox (iso_fast_mailbox.hip): pinned host memory, device-mapped, 64-byte lines
constexpr int ISO_MAILBOX_ROWS = 128;
struct IsoMailbox {
    unsigned long long req[8];        // line 0, host -> device: req[0] = sequence word (checksum << 32 | counter << 16 | parts << 8 | rows - 1;
                                      //         checksum = mailbox_checksum of the words of a one-row request, so that a line
                                      //         whose eight words did not arrive together is seen as such and polled again),
                                      //         req[1..7] = the parameters of a one-row request
    unsigned long long done[8];       // line 1, device -> host: done[0] = sequence word of the last finished request,
                                      //         done[1..3] = lnpost, lnprior, lnlike of a one-row request
    unsigned long long ctl[8];        // line 2: ctl[0] = state (0 none, 1 running, 2 exited; the device writes 2),
                                      //         ctl[1] = quit (host -> device)
    double rows[ISO_MAILBOX_ROWS * ISO_MAX_PARAMS];     // requests of 2..128 rows, [row][parameter]
    double out[3 * ISO_MAILBOX_ROWS];                   // their results: lnpost | lnprior | lnlike
};

// 32-bit checksum of the parameter words of a one-row mailbox request (host and device compute the same number)
__host__ __device__ inline uint32_t mailbox_checksum(const unsigned long long* w, int n)
{
    uint32_t c = 0x9E3779B9u;
    for (int q = 0; q < n; ++q) {
        c = (c ^ (uint32_t)w[q]) * 0x85EBCA6Bu;
        c = (c ^ (uint32_t)(w[q] >> 32)) * 0xC2B2AE35u + (uint32_t)q;
    }
    return c;
}

// closed-form age prior of an IsoTrackModel (the reference's AgePrior, flat in linear age): lnorm + age ln 10 inside [lo, hi]
struct IsoTrackAge {
    double lo, hi, lnorm;
};

}  // namespace iso

struct iso_ctx {
    int device;
    // pinned, device-mapped staging for the *_host entry points (scalar / small-batch calls from Python: one launch
    // + one synchronise per call, kernels read and write host memory through PCIe); lazily allocated
    double* h_stage;
    std::mutex stage_mu;
    unsigned long long stage_seq;    // sequence number of the completion flag kept behind the staging area
};
constexpr int64_t ISO_CTX_STAGE_DOUBLES = 1 << 17;      // 1 MiB (+ 8 doubles for the completion flag)

struct iso_table {
    int device;
    iso_ctx* ctx;
    int ndim;
    int64_t shape[ISO_MAX_DIM + 1];
    int64_t ncells;
    double* d_grid;
    double* d_wide;          // 3-D tables: [cell][column][corner] pack for k_interp3_wide (lazy), may be null
    bool wide_failed;        // do not try to build it again
    std::mutex wide_mu;
    double* d_axes[ISO_MAX_DIM];
    std::vector<double> h_axes[ISO_MAX_DIM];
    iso::AxisD ax[ISO_MAX_DIM];
};

// corner-packed BC table for one band list, built on first use by iso_interp_mag (fast form)
struct MagPack {
    std::vector<int32_t> cols;
    double* d_bc_hot;
    double* d_bcq;
    double* d_axes_blob;
    iso::FastArgs fast;
    uint64_t last_use;
};

// Corner-packed BC + staged axes of one band list, shared by the catalogs an interpolator's fits build (a fit in slices /
// shards creates one catalog per slice; packing 54 MB per band and planning the bucket tables again for each cost 1.5 ms
// of a 10^4-star fit).  Reference-counted: the interpolator's cache and every catalog hold a reference, the device
// buffers go with the last one - a catalog never reaches back into its interpolator to release anything.
struct BandPack {
    int device;
    std::vector<int32_t> cols;
    double* d_bcq;
    double* d_axes_blob;
    iso::FastArgs fast;
    size_t bytes;            // of the corner-packed copy (what the interpolator's cache is bounded by)
    ~BandPack();
};

struct iso_ic {
    int device;
    iso_ctx* ctx;
    iso_table* model;
    iso_table* bc;
    int kind;
    int32_t cols[4], prior_cols[2], astero_cols[2];
    double* d_hot;
    double* d_hotq;          // corner-packed model table (fast path), may be null
    double* d_astq;          // corner-packed (nu_max, delta_nu) for the fast kernel, built by the first
                             // asteroseismic model (guarded by mag_mu)
    iso::Grid3V g3;
    iso::Grid4V g4;          // full BC table view
    int lds_doubles;         // LDS staging size for model + BC axes (generic kernels)
    std::vector<double> h_axes_model[3], h_axes_bc[4];
    std::mutex mag_mu;       // guards mag_packs
    std::vector<MagPack> mag_packs;
    std::vector<std::shared_ptr<BandPack>> band_packs;   // catalogs' packs, most recently used last (guarded by mag_mu)
    uint64_t mag_clock;
};

struct iso_catalog {
    int device;
    iso_ic* ic;
    int64_t n_models;
    int n_stars, n_bands;
    iso::DevModel* d_models;  // [n_models]
    size_t models_bytes;      // size of d_models' block when it came from the pool of per-fit buffers (0: plain hipMalloc)
    double* d_bc_hot;
    double* d_bcq;
    double* d_axes_blob;
    bool packed;
    iso::FastArgs fast;
    std::shared_ptr<BandPack> pack;   // the band list's shared pack (d_bcq / d_axes_blob above are then null)
    int std_priors;                   // the priors the stars share are the reference's default families (and they do share them)
};

struct iso_model {
    int device;              // copied: destroy order of handles is up to the caller / a GC
    iso_ic* ic;
    iso_model_desc desc;
    iso::DevModel* d_model;
    double* d_bc_hot;        // BC table restricted to the model's bands, [..][nb]
    double* d_bcq;           // corner-packed BC for the model's bands, may be null
    double* d_axes_blob;     // fast path: staged axes
    iso::Grid4V g4;          // view of d_bc_hot
    bool fast_ok;
    iso::FastArgs fast;      // template filled at create time (pars/outputs set per call)
    double* h_stage;         // pinned, device-mapped staging for iso_lnpost_host (lazy)
    int64_t stage_rows;
    unsigned long long stage_seq;   // sequence number of the completion flag at the end of h_stage
    // large host batches (iso_lnpost_host beyond the staging buffer): device rows + two streams, kept between calls
    double* d_pipe;          // device: the uploaded parameter rows
    double* h_pipe;          // pinned, device-mapped: the kernels write their results straight into host memory
    int64_t pipe_rows;
    hipStream_t pipe_stream[2];
    std::mutex host_mu;      // iso_lnpost_host: one caller at a time per model (the staging areas are the model's)
    // resident mailbox wave of the per-point callback (lazy; guarded by host_mu)
    iso::IsoMailbox* mbox;   // pinned, device-mapped
    iso::IsoMailbox* d_mbox; // its device address
    hipStream_t mbox_stream; // non-blocking: the resident wave must not order itself against the null stream
    unsigned long long mbox_count;   // requests posted
    int mbox_state;          // 0 untried, 1 usable, -1 not available for this model (no instantiation / allocation failed)
};

struct iso_sampler {
    int device;
    int kind, n_stars, n_bands, n_params;
    int64_t n_ensembles;     // stars sampled in lock-step (1 for a single model)
    int W;
    double a;
    uint64_t seed;
    uint32_t step;           // running step counter (keeps the RNG stream moving across runs)
    int multi;
    int std_priors;          // single model whose priors are the reference's default families (compile-time kinds)
    int chain_layout;        // ISO_CHAIN_ROW_MAJOR / ISO_CHAIN_PARAM_MAJOR
    iso::FastArgs fast;      // tables + model(s); copied at create time (owner must outlive the sampler)
    // the any-model persistent sampler (fast/sampler_any.h): 0 = the fused BasicStarModel kernels above,
    // ISO_SAMPLER_TREE / _ISOTRACK / _WIDE otherwise
    int form;
    const iso::DevTree* d_tree;   // tree: the model's device record (the tree model must outlive the sampler)
    int n_leaves;
    iso::FastArgs fast2;     // isotrack: the track-grid model's tables (fast = the isochrone-grid model's)
    iso::IsoTrackAge age;
};
constexpr int ISO_SAMPLER_TREE = 1, ISO_SAMPLER_ISOTRACK = 2, ISO_SAMPLER_WIDE = 3;

namespace iso {
struct MagOut {
    double *Teff, *logg, *feh, *mags;     // each may be null; mags is [n][nb]
};
// defined in iso_fast_mag.hip: interp_mag on the corner-packed tables (nb = 1..12)
bool launch_interp_mag_fast(int kind, int nb, const FastArgs& A, const MagOut& O, hipStream_t s);
// defined in iso_fast_tree.hip: observation-tree lnpost on the corner-packed tables (1..12 bands)
bool launch_tree_fast(int nb, int n_leaves, const FastArgs& A, const DevTree* T, hipStream_t s);
bool launch_stretch(int kind, int n_stars, int n_bands, const FastArgs& A, const StretchArgs& S, hipStream_t s);
// defined in iso_fast_mailbox.hip: start a model's resident mailbox wave (false: no instantiation for the shape)
bool launch_mailbox(int kind, int n_stars, int n_bands, const FastArgs& A, IsoMailbox* d_mb, unsigned long long idle_ticks,
                    unsigned long long life_ticks, hipStream_t s);
// defined in iso_fast_stretch_tree.hip / iso_fast_stretch_more.hip: the any-model persistent sampler (one workgroup per
// ensemble).  `query` non-null: report whether a kernel exists for the shape and its LDS fits a CU, launch nothing.
bool launch_stretch_tree(int nb, int n_leaves, const FastArgs& A, const DevTree* T, const AnyStretchArgs& S, int* query, hipStream_t s);
bool launch_stretch_isotrack(int nb, const FastArgs& Ai, const FastArgs& At, const IsoTrackAge& P, AnyStretchArgs S, int* query,
                             hipStream_t s);
bool launch_stretch_wide(int kind, int n_stars, const FastArgs& A, AnyStretchArgs S, int* query, hipStream_t s);
// dynamic LDS bytes one workgroup of the persistent sampler kernel needs for W-walker ensembles, and how
// many ensembles such a workgroup owns
size_t stretch_persist_lds(int n_bands, int axes_len, int W, int n_params, int* ensembles_per_workgroup);
// defined in iso_fast_*.hip: launch the specialised fused kernel; returns false if no
// specialisation exists for (kind, n_stars, n_bands)
bool launch_lnpost_fast(int kind, int n_stars, int n_bands, bool multi, const FastArgs& A, hipStream_t s);
// defined in iso_fast.hip: the start-point kernel of a catalog fit (fast/start_points.h); 0 = no instantiation for this shape
int launch_catalog_start(int kind, int n_stars, int n_bands, const FastArgs& A, double* best, double* best_lnp, int32_t* failed,
                         int64_t n_models, int W, int oversample, int max_tries, uint64_t seed, hipStream_t s);
// test hook (iso_debug_trace_kernels / iso_debug_kernels): launchers name the instantiation they chose, spelled as c++filt
// spells the kernel's symbol; a no-op unless tracing is on
void note_kernel(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
inline const char* tf(bool b) { return b ? "true" : "false"; }
}  // namespace iso
