// isochrones_amd — hand-written HIP (gfx950 / CDNA4) implementation of the isochrones hot path
// behind the C ABI of include/isochrones_amd.h.
//
// This file: the C ABI and its host-side bookkeeping.  Device code (all float64, gather/latency bound — no MFMA
// on purpose, this is interpolation) lives in
//   kernels/k_interp.h          K3  N-D bracket search + multilinear gather of an arbitrary column subset
//                                   (isochrones/interp.py:10-35, 63-338); column-parallel and wide-pack forms
//   kernels/k_interp_mag.h      K4  3-D model gather -> 4-D BC gather -> magnitudes (isochrones/mags.py:8-124)
//   kernels/k_lnpost_generic.h  K1+K2 fused: priors + 1-3 component stars + likelihood reduce, any table shape
//                                   (likelihood.py:16-147, starmodel.py:538-542,1563-1635, priors.py)
//   kernels/k_lnpost_tree.h     generic StarModel over a flattened ObservationTree
//   kernels/k_interp_eep.h, k_chain_quantiles.h, k_small_and_pack.h (mnest_prior, table packers)
//   kernels/dev_*.h             shared device helpers (axis staging, brackets, gathers, prior families)
//   iso_fast_kernel.h + iso_fast_*.hip   the fast fused kernels on the corner-packed tables (lnpost, sampler,
//                                   interp_mag, tree), one translation unit per parametrisation / star count
//
// Mapping: one lane = one sample, 256-thread workgroups (4 wave64), grid-stride over samples.
// Short irregular axes are staged in LDS once per workgroup and searched with a branch-free
// bisection; exactly-uniform axes (the integer EEP axis) are indexed in O(1).  The bracket rule
//   i = clamp(#{a_j <= x} - 1, 0, n-2),  t = (x - a_i) / (a_{i+1} - a_i)
// reproduces the reference's searchsorted/find_indices results bit-for-bit (an exact node hit
// gives t = 0 either way) and defines the reference's undefined exact-upper-edge query as
// (n-2, t=1).  Zero-weight corners are still accumulated so NaN padding propagates exactly
// like the reference (docs/interpolate.ipynb cell 14).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <exception>
#include <thread>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

// ======================================================================================
// host-side bookkeeping
// ======================================================================================
#include <cstdarg>
#include <cstdio>
#include "iso_internal.h"
#include "fast/tree_mailbox.h"
#include "fast/axis_lut.h"

using namespace iso;

namespace {
thread_local std::string g_err;
}

namespace iso {
int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
}  // namespace iso

BandPack::~BandPack()
{
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    (void)hipSetDevice(device);
    if (d_bcq) (void)hipFree(d_bcq);
    if (d_axes_blob) (void)hipFree(d_axes_blob);
    if (prev >= 0) (void)hipSetDevice(prev);
}

namespace {
// the reference's default prior families (starmodel.py:1459-1475, priors.py): the sampler kernels then have them as
// compile-time constants
bool default_prior_families(const iso_model_desc& d)
{
    return d.prior_mass.kind == ISO_PRIOR_CHABRIER && d.prior_age.kind == ISO_PRIOR_FLATLOG && d.prior_feh.kind == ISO_PRIOR_FEH &&
           d.prior_feh.c != 0.0 && d.prior_distance.kind == ISO_PRIOR_POWERLAW && d.prior_AV.kind == ISO_PRIOR_FLAT;
}
}  // namespace

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ISO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

// ======================================================================================
// device code
// ======================================================================================
namespace {

#include "kernels/dev_common.h"
#include "kernels/k_interp.h"
#include "kernels/dev_gather.h"
#include "kernels/k_interp_mag.h"
#include "kernels/dev_priors.h"
#include "kernels/k_lnpost_generic.h"
#include "kernels/k_interp_eep.h"
#include "kernels/k_lnpost_tree.h"
#include "kernels/k_chain_quantiles.h"
#include "kernels/k_small_and_pack.h"
#include "kernels/k_service.h"

// ======================================================================================
// host helpers
// ======================================================================================
int grid_blocks(int64_t n)
{
    int64_t b = (n + BLOCK - 1) / BLOCK;
    const int64_t cap = 256 * 8;   // 256 CUs x 8 workgroups, grid-stride beyond that
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

bool axis_uniform(const std::vector<double>& a, double& a0, double& step)
{
    if (a.size() < 2) return false;
    a0 = a[0];
    step = a[1] - a[0];
    if (!(step > 0)) return false;
    for (size_t i = 0; i < a.size(); ++i) {
        const double plain = (double)i * step + a0;
        const double fused = std::fma((double)i, step, a0);
        if (plain != a[i] || fused != a[i]) return false;
    }
    // the O(1) index also needs (x - a0)/step to be within one cell of the truth: guaranteed for
    // exact arithmetic nodes; keep the fast path to modest sizes
    return a.size() < (1u << 24);
}

// assign LDS offsets to the non-uniform axes of up to two tables; returns doubles used
int assign_lds(AxisD* a, int na, AxisD* b, int nb)
{
    int used = 0;
    AxisD* sets[2] = {a, b};
    int counts[2] = {na, nb};
    for (int s = 0; s < 2; ++s)
        for (int d = 0; d < counts[s]; ++d) {
            AxisD& A = sets[s][d];
            A.lds_off = -1;
            if (A.uniform) continue;
            if (used + A.n <= MAX_LDS_AXIS_DOUBLES) {
                A.lds_off = used;
                used += A.n;
            }
        }
    return used;
}

hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

double host_powerlaw_C(double alpha, double lo, double hi)
{
    return (1 + alpha) / (std::pow(hi, 1 + alpha) - std::pow(lo, 1 + alpha));
}

DevPrior make_dev_prior(const iso_prior& P)
{
    DevPrior D;
    std::memset(&D, 0, sizeof(D));
    D.kind = P.kind;
    D.bounded = P.bounded;
    D.lo = P.lo; D.hi = P.hi;
    D.a = P.a; D.b = P.b; D.c = P.c; D.d = P.d; D.e = P.e; D.f = P.f; D.g = P.g; D.h = P.h;
    switch (P.kind) {
    case ISO_PRIOR_FLAT:
        D.k0 = 1.0 / (P.hi - P.lo);
        D.k1 = std::log(D.k0);
        break;
    case ISO_PRIOR_FLATLOG:
        D.k0 = std::pow(10.0, P.hi) - std::pow(10.0, P.lo);
        D.k1 = std::log(std::log(10.0) / D.k0);
        break;
    case ISO_PRIOR_POWERLAW:
        D.k0 = host_powerlaw_C(P.a, P.lo, P.hi);
        D.k1 = std::log(D.k0);
        break;
    case ISO_PRIOR_GAUSS:
        D.k0 = std::exp(P.c);
        D.k1 = std::log(P.b);
        D.r0 = 1.0 / P.b;
        break;
    case ISO_PRIOR_LOGNORMAL:
        D.k0 = std::exp(P.a);
        D.k1 = std::log(P.b);
        D.r0 = 1.0 / D.k0;
        D.r1 = 1.0 / P.b;
        break;
    case ISO_PRIOR_CHABRIER:
        D.k0 = std::exp(P.a);
        D.k1 = std::log(P.b);
        D.k2 = host_powerlaw_C(P.c, P.g, P.h);
        D.k3 = std::log(P.e);
        D.k4 = std::log(P.f);
        D.k5 = std::log(D.k2);
        D.r0 = 1.0 / D.k0;
        D.r1 = 1.0 / P.b;
        break;
    case ISO_PRIOR_FEH:
        D.k0 = 1.0 / std::sqrt(2 * M_PI * 0.4 * 0.4);
        D.r0 = 1.0 / P.b;
        break;
    }
    return D;
}

bool prior_kind_ok(int k) { return k >= ISO_PRIOR_FLAT && k <= ISO_PRIOR_FEH; }

void gauss_consts(double unc, double& g0, double& unc2, double* hinv = nullptr)
{
    g0 = std::log(1.0 / std::sqrt(2 * M_PI)) + std::log(unc);
    unc2 = unc * unc;
    if (hinv) *hinv = 0.5 / unc2;
}

// "auto" (default): fast kernel on corner-packed tables when eligible; "compact": fast kernel on
// the compact tables; "generic": always the generic kernel.  For A/B measurements and tests.
enum PathMode { PATH_AUTO = 0, PATH_COMPACT = 1, PATH_GENERIC = 2 };

PathMode path_mode()
{
    const char* e = std::getenv("ISOCHRONES_AMD_PATH");
    if (!e) return PATH_AUTO;
    if (!std::strcmp(e, "generic")) return PATH_GENERIC;
    if (!std::strcmp(e, "compact")) return PATH_COMPACT;
    return PATH_AUTO;
}

constexpr int FAST_NB_MAX = 12;        // band count up to which every fused form exists (batch, catalog, samplers)
constexpr int FAST_MAX_BLOB = 4096;   // doubles (32 KiB of LDS) the fast kernel may stage

// third model axis (EEP): exactly uniform -> O(1) index; otherwise the fused kernels bisect it (every 8th node in LDS, a
// window of 9 nodes from the device copy of the axis), which needs at least 9 nodes
bool third_axis_ok(const iso_ic* ic) { return ic->model->ax[2].uniform || ic->model->ax[2].n >= 9; }


hipError_t pack_corners(const double* src, int ncol, int keep, int ndim, const int64_t* n, double** out, int col0 = 0)
{
    PackCornersArgs P;
    P.src = src;
    P.ncol = ncol;
    P.keep = keep;
    P.col0 = col0;
    P.ndim = ndim;
    P.ncells = 1;
    for (int d = 0; d < 4; ++d) P.n[d] = 1;
    for (int d = 0; d < ndim; ++d) {
        P.n[d] = n[d];
        P.ncells *= n[d];
    }
    if (P.ncells >= (int64_t(1) << 32)) return hipErrorInvalidValue;     // the fused kernels number cells in 32 bits
    const size_t bytes = (size_t)P.ncells * (size_t)(1 << ndim) * keep * sizeof(double);
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) return e;
    P.out = *out;
    note_kernel("k_pack_corners");
    hipLaunchKernelGGL(k_pack_corners, dim3(grid_blocks(P.ncells * (1 << ndim) * keep)), dim3(BLOCK), 0, 0, P);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(*out);
        *out = nullptr;
    }
    return e;
}

}  // namespace

// ======================================================================================
// C ABI
// ======================================================================================
struct iso_tree_model {
    int device;
    iso_ic* ic;
    int n_params;
    DevTree* d_tree;
    double* d_bc_hot;
    Grid4V g4;
    int n_bands, n_leaves;
    double* d_bcq;           // corner-packed BC for the tree's bands (fast form), may be null
    double* d_axes_blob;
    bool fast_ok;
    FastArgs fast;
    // resident mailbox wave of the per-point callback (lazy; calls are serialised by the context's stage_mu)
    iso::IsoTreeBox* mbox;   // pinned, device-mapped
    iso::IsoTreeBox* d_mbox;
    hipStream_t mbox_stream;
    unsigned long long mbox_count;
    int mbox_state;          // 0 untried, 1 usable, -1 not available
};

struct iso_eep_table {
    int device;
    iso_ctx* ctx;
    int64_t n0, n1, n_eep;
    double eep0;
    double *d_ages, *d_ax0, *d_ax1;
    int64_t* d_lengths;
    AxisD ax[2];
};

namespace {
// Wide pack of a 3-D table (see k_interp3_wide): built by the first batch of >= WIDE_BUILD_MIN_ROWS rows if
// it fits the budget; 1 = available, 0 = use the column-parallel kernel, < 0 = error.
const int64_t WIDE_USE_MIN_ROWS = 1024, WIDE_BUILD_MIN_ROWS = 32768;
const size_t WIDE_MAX_BYTES = (size_t)32 << 30;

int ensure_wide_pack(iso_table* t, int64_t n)
{
    std::lock_guard<std::mutex> lock(t->wide_mu);
    if (t->d_wide) return 1;
    if (t->wide_failed || n < WIDE_BUILD_MIN_ROWS) return 0;
    const size_t bytes = (size_t)t->ncells * (size_t)t->shape[3] * 8 * sizeof(double);
    if (bytes > WIDE_MAX_BYTES) {
        t->wide_failed = true;
        return 0;
    }
    double* w = nullptr;
    hipError_t e = hipMalloc(&w, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        t->wide_failed = true;        // no room: keep using the column-parallel kernel
        return 0;
    }
    PackWideArgs P;
    P.grid = t->d_grid;
    P.out = w;
    P.n0 = t->shape[0]; P.n1 = t->shape[1]; P.n2 = t->shape[2];
    P.ncol = (int)t->shape[3];
    note_kernel("k_pack_wide");
    hipLaunchKernelGGL(k_pack_wide, dim3(grid_blocks((int64_t)(bytes / sizeof(double)))), dim3(BLOCK), 0, 0, P);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(w);
        return fail(ISO_ERR_HIP, std::string("iso_interp: wide pack: ") + hipGetErrorString(e));
    }
    t->d_wide = w;
    return 1;
}

}  // namespace

namespace {
// the resident service wave of the scalar accessors (defined with the *_host entry points below)
void service_stop(iso_ctx* ctx, bool release);
void service_forget(const void* key);
void free_mag_pack(MagPack& mp);
int acquire_mag_pack(iso_ic* ic, const int32_t* bc_cols, int nb, int64_t n, iso::FastArgs& F);
hipError_t acquire_band_pack(iso_ic* ic, const int32_t* bc_cols, int nb, std::shared_ptr<BandPack>* out, bool* ok);
}  // namespace

// dynamic LDS a persistent sampler workgroup may ask for (a CU's whole LDS)
static constexpr size_t PERSIST_LDS_MAX = 160 * 1024;

// ---- test hook: which kernel instantiations the launchers chose (iso_debug_trace_kernels / iso_debug_kernels) ----------
namespace iso {
namespace {
std::atomic<int> g_trace_kernels{0};
thread_local std::vector<std::string> t_kernels;
}  // namespace

void note_kernel(const char* fmt, ...)
{
    if (!g_trace_kernels.load(std::memory_order_relaxed)) return;
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    for (const std::string& k : t_kernels)
        if (k == buf) return;
    if (t_kernels.size() < 4096) t_kernels.emplace_back(buf);
}
}  // namespace iso

// what the calling thread's last iso_sampler_run decided: {persistent, dense, threads, dense_stdp, group, workgroups per CU,
// workgroups, form} (tests assert the launch shape of the catalog routes with it; the kernel's NAME is in the trace above)
namespace iso {
namespace {
thread_local int32_t t_sampler_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
}
}  // namespace iso

extern "C" int iso_debug_sampler_plan(int32_t* out8)
{
    if (!out8) return ISO_ERR_INVALID;
    std::memcpy(out8, iso::t_sampler_plan, sizeof iso::t_sampler_plan);
    return ISO_OK;
}

extern "C" int iso_debug_trace_kernels(int on)
{
    iso::t_kernels.clear();
    return iso::g_trace_kernels.exchange(on ? 1 : 0);
}

extern "C" int64_t iso_debug_kernels(char* buf, int64_t size)
{
    std::string all;
    for (const std::string& k : iso::t_kernels) {
        all += k;
        all += '\n';
    }
    if (buf && size > 0) {
        const size_t n = std::min<size_t>((size_t)size - 1, all.size());
        std::memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return (int64_t)all.size() + 1;
}

extern "C" {

const char* iso_last_error(void) { return g_err.c_str(); }

const char* iso_version(void) { return "isochrones_amd 0.1.0 (gfx950 HIP)"; }

int iso_ctx_create(iso_ctx** out, int device)
{
    if (!out) return fail(ISO_ERR_INVALID, "iso_ctx_create: out is NULL");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(ISO_ERR_INVALID, "iso_ctx_create: no such device");
    iso_ctx* c = new (std::nothrow) iso_ctx;
    if (!c) return fail(ISO_ERR_NOMEM, "iso_ctx_create: out of host memory");
    c->device = device;
    c->h_stage = nullptr;
    *out = c;
    return ISO_OK;
}

void iso_ctx_destroy(iso_ctx* ctx)
{
    if (!ctx) return;
    {
        DeviceGuard guard(ctx->device);
        service_stop(ctx, true);
    }
    if (ctx->h_stage) {
        DeviceGuard guard(ctx->device);
        (void)hipHostFree(ctx->h_stage);
    }
    delete ctx;
}

namespace {
int table_create(iso_ctx* ctx, int ndim, const int64_t* shape, const double* grid, bool grid_on_device, const double* const* axes,
                 iso_table** out);
}

int iso_table_create(iso_ctx* ctx, int ndim, const int64_t* shape, const double* grid, const double* const* axes,
                     iso_table** out)
{
    return table_create(ctx, ndim, shape, grid, false, axes, out);
}

int iso_table_create_from_device(iso_ctx* ctx, int ndim, const int64_t* shape, const double* d_grid, const double* const* axes,
                                 iso_table** out)
{
    return table_create(ctx, ndim, shape, d_grid, true, axes, out);
}

namespace {
int table_create(iso_ctx* ctx, int ndim, const int64_t* shape, const double* grid, bool grid_on_device, const double* const* axes,
                 iso_table** out)
{
    if (!ctx || !shape || !grid || !axes || !out) return fail(ISO_ERR_INVALID, "iso_table_create: NULL argument");
    if (ndim < 2 || ndim > ISO_MAX_DIM) return fail(ISO_ERR_INVALID, "iso_table_create: ndim must be 2, 3 or 4");
    int64_t ncells = 1;
    for (int d = 0; d < ndim; ++d) {
        if (shape[d] < 2 || shape[d] > (1 << 30)) return fail(ISO_ERR_INVALID, "iso_table_create: axis length < 2");
        ncells *= shape[d];
    }
    if (shape[ndim] < 1) return fail(ISO_ERR_INVALID, "iso_table_create: no columns");
    for (int d = 0; d < ndim; ++d)
        for (int64_t j = 0; j < shape[d]; ++j) {
            const double v = axes[d][j];
            if (!(v == v) || (j > 0 && !(axes[d][j - 1] < v)))
                return fail(ISO_ERR_INVALID, "iso_table_create: axis values must be strictly increasing");
        }
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(ISO_ERR_HIP, "iso_table_create: hipSetDevice failed");
    iso_table* t = new (std::nothrow) iso_table();
    if (!t) return fail(ISO_ERR_NOMEM, "iso_table_create: out of host memory");
    t->ctx = ctx;
    t->device = ctx->device;
    t->ndim = ndim;
    t->ncells = ncells;
    t->d_grid = nullptr;
    t->d_wide = nullptr;
    t->wide_failed = false;
    for (int d = 0; d < ISO_MAX_DIM; ++d) t->d_axes[d] = nullptr;
    for (int d = 0; d <= ndim; ++d) t->shape[d] = shape[d];
    const size_t bytes = (size_t)ncells * (size_t)shape[ndim] * sizeof(double);
    hipError_t e = hipMalloc(&t->d_grid, bytes);
    if (e == hipSuccess) e = hipMemcpy(t->d_grid, grid, bytes, grid_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
    for (int d = 0; d < ndim && e == hipSuccess; ++d) {
        t->h_axes[d].assign(axes[d], axes[d] + shape[d]);
        e = hipMalloc(&t->d_axes[d], shape[d] * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(t->d_axes[d], axes[d], shape[d] * sizeof(double), hipMemcpyHostToDevice);
        AxisD& A = t->ax[d];
        A.g = t->d_axes[d];
        A.n = (int)shape[d];
        A.lds_off = -1;
        A.uniform = axis_uniform(t->h_axes[d], A.a0, A.step) ? 1 : 0;
    }
    if (e != hipSuccess) {
        std::string msg = std::string("iso_table_create: ") + hipGetErrorString(e);
        iso_table_destroy(t);
        return fail(e == hipErrorOutOfMemory ? ISO_ERR_NOMEM : ISO_ERR_HIP, msg);
    }
    *out = t;
    return ISO_OK;
}
}  // namespace

void iso_table_destroy(iso_table* t)
{
    if (!t) return;
    DeviceGuard guard(t->device);
    service_stop(t->ctx, false);     // a resident service wave may have this table's axes staged / be reading it
    service_forget(t);
    if (t->d_grid) (void)hipFree(t->d_grid);
    if (t->d_wide) (void)hipFree(t->d_wide);
    for (int d = 0; d < ISO_MAX_DIM; ++d)
        if (t->d_axes[d]) (void)hipFree(t->d_axes[d]);
    delete t;
}

int iso_interp(iso_table* t, const double* const* x, int64_t n, const int32_t* icols, int k, double* out,
               void* stream)
{
    if (!t || !x || !icols || (!out && n > 0)) return fail(ISO_ERR_INVALID, "iso_interp: NULL argument");
    if (k < 1 || k > ISO_MAX_COLS) return fail(ISO_ERR_INVALID, "iso_interp: k out of range");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_interp: n < 0");
    if (n == 0) return ISO_OK;
    InterpArgs A;
    std::memset(&A, 0, sizeof(A));
    for (int d = 0; d < t->ndim; ++d) {
        A.ax[d] = t->ax[d];
        A.x[d] = x[d];
        if (!x[d]) return fail(ISO_ERR_INVALID, "iso_interp: NULL coordinate pointer");
    }
    const int lds = assign_lds(A.ax, t->ndim, nullptr, 0);
    int64_t s = 1;
    for (int d = t->ndim - 1; d >= 0; --d) {
        A.stride[d] = s;
        s *= t->shape[d];
    }
    A.grid = t->d_grid;
    A.ncol = (int)t->shape[t->ndim];
    A.n = n;
    A.k = k;
    for (int c = 0; c < k; ++c) {
        if (icols[c] < 0 || icols[c] >= A.ncol) return fail(ISO_ERR_INVALID, "iso_interp: column index out of range");
        A.icols[c] = icols[c];
    }
    A.out = out;
    DeviceGuard guard(t->ctx->device);
    if (t->ndim == 3 && path_mode() == PATH_AUTO && n >= WIDE_USE_MIN_ROWS) {
        const int rc = ensure_wide_pack(t, n);
        if (rc < 0) return rc;
        if (rc == 1) {
            WideArgs W;
            std::memset(&W, 0, sizeof(W));
            for (int d = 0; d < 3; ++d) {
                W.ax[d] = A.ax[d];
                W.stride[d] = A.stride[d];
                W.x[d] = x[d];
            }
            W.wide = t->d_wide;
            W.ncol = A.ncol;
            W.n = n;
            W.k = k;
            W.kinv = ((uint64_t)1 << 32) / (uint64_t)k + 1;
            W.lds_axes = lds;
            for (int c = 0; c < k; ++c) W.icols[c] = A.icols[c];
            W.out = out;
            const size_t sh = (size_t)(lds + ISO_MAX_COLS / 2 + BLOCK * WIDE_SLOT) * sizeof(double);
            // groups of 64 samples per wave: as many as leave every CU ~8 workgroups of work (a workgroup's fixed part -
            // staging the axes, one barrier - is then paid once per 256 x groups samples); A/B switches for both choices
            const int64_t wgs1 = (n + BLOCK - 1) / BLOCK;
            // measured (tools/wide_form_ab.py, profiles/r04/wide_form_ab.jsonl; 10^6 samples): 1 column 31.8 -> 29.0 us with
            // the four-pass form, 28.2 with four groups; 2 columns 49.8 -> 45.1; 3 columns 69.1 -> 67.1; 18 columns 206 -> 208
            // (a wave already has 72 passes of work there); eight groups, or any grouping at 10^5 samples, leave CUs idle
            int groups = k <= 3 ? (int)std::min<int64_t>(4, std::max<int64_t>(1, wgs1 / 900)) : 1;
            if (const char* e = std::getenv("ISOCHRONES_AMD_WIDE_GROUPS")) groups = std::max(1, std::atoi(e));
            bool narrow = k == 1;
            if (const char* e = std::getenv("ISOCHRONES_AMD_WIDE_NARROW")) narrow = narrow && std::atoi(e) != 0;
            W.groups = groups;
            const int64_t per_wg = (int64_t)BLOCK * groups;
            const dim3 grid((unsigned)((n + per_wg - 1) / per_wg));
            note_kernel("k_interp3_wide<%d>", narrow ? 4 : WIDE_UNROLL);
            if (narrow) hipLaunchKernelGGL(k_interp3_wide<4>, grid, dim3(BLOCK), sh, as_stream(stream), W);
            else hipLaunchKernelGGL(k_interp3_wide<WIDE_UNROLL>, grid, dim3(BLOCK), sh, as_stream(stream), W);
            HIP_TRY(hipGetLastError());
            return ISO_OK;
        }
    }
    const int lanes_per_sample = (k + 1) / 2, samples_per_wave = 64 / lanes_per_sample;
    const int64_t waves = (n + samples_per_wave - 1) / samples_per_wave;
    const dim3 g(grid_blocks(waves * 64)), b(BLOCK);
    const size_t shmem = (size_t)lds * sizeof(double);
    switch (t->ndim) {
    case 2: note_kernel("k_interp<2>"); hipLaunchKernelGGL(k_interp<2>, g, b, shmem, as_stream(stream), A); break;
    case 3: note_kernel("k_interp<3>"); hipLaunchKernelGGL(k_interp<3>, g, b, shmem, as_stream(stream), A); break;
    default: note_kernel("k_interp<4>"); hipLaunchKernelGGL(k_interp<4>, g, b, shmem, as_stream(stream), A); break;
    }
    HIP_TRY(hipGetLastError());
    return ISO_OK;
}

int iso_ic_create(iso_ctx* ctx, iso_table* model_grid, iso_table* bc_grid, int kind, const int32_t cols[4],
                  const int32_t prior_cols[2], const int32_t astero_cols[2], iso_ic** out)
{
    if (!ctx || !model_grid || !bc_grid || !cols || !prior_cols || !astero_cols || !out)
        return fail(ISO_ERR_INVALID, "iso_ic_create: NULL argument");
    if (model_grid->ndim != 3) return fail(ISO_ERR_INVALID, "iso_ic_create: model table must be 3-D");
    if (bc_grid->ndim != 4) return fail(ISO_ERR_INVALID, "iso_ic_create: BC table must be 4-D");
    if (kind != ISO_KIND_TRACK && kind != ISO_KIND_ISO) return fail(ISO_ERR_INVALID, "iso_ic_create: bad kind");
    if (model_grid->ctx->device != ctx->device || bc_grid->ctx->device != ctx->device)
        return fail(ISO_ERR_INVALID, "iso_ic_create: tables live on another device");
    const int ncol = (int)model_grid->shape[3];
    for (int q = 0; q < 4; ++q)
        if (cols[q] < 0 || cols[q] >= ncol) return fail(ISO_ERR_INVALID, "iso_ic_create: column index out of range");
    for (int q = 0; q < 2; ++q) {
        if (prior_cols[q] < -1 || prior_cols[q] >= ncol || astero_cols[q] < -1 || astero_cols[q] >= ncol)
            return fail(ISO_ERR_INVALID, "iso_ic_create: column index out of range");
    }
    DeviceGuard guard(ctx->device);
    iso_ic* ic = new (std::nothrow) iso_ic();
    if (!ic) return fail(ISO_ERR_NOMEM, "iso_ic_create: out of host memory");
    ic->ctx = ctx;
    ic->device = ctx->device;
    ic->model = model_grid;
    ic->bc = bc_grid;
    ic->kind = kind;
    std::memcpy(ic->cols, cols, sizeof(ic->cols));
    std::memcpy(ic->prior_cols, prior_cols, sizeof(ic->prior_cols));
    std::memcpy(ic->astero_cols, astero_cols, sizeof(ic->astero_cols));
    ic->d_hot = nullptr;
    const size_t bytes = (size_t)model_grid->ncells * HOT_COLS * sizeof(double);
    hipError_t e = hipMalloc(&ic->d_hot, bytes);
    if (e != hipSuccess) {
        delete ic;
        return fail(e == hipErrorOutOfMemory ? ISO_ERR_NOMEM : ISO_ERR_HIP,
                    std::string("iso_ic_create: hipMalloc(hot table): ") + hipGetErrorString(e));
    }
    PackHotArgs P;
    P.grid = model_grid->d_grid;
    P.ncol = ncol;
    P.ncells = model_grid->ncells;
    const int32_t src[HOT_COLS] = {cols[0], cols[1], cols[2], cols[3], prior_cols[0], prior_cols[1],
                                   astero_cols[0], astero_cols[1]};
    std::memcpy(P.src, src, sizeof(src));
    P.hot = ic->d_hot;
    note_kernel("k_pack_hot");
    hipLaunchKernelGGL(k_pack_hot, dim3(grid_blocks(P.ncells * HOT_COLS)), dim3(BLOCK), 0, 0, P);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(ic->d_hot);
        delete ic;
        return fail(ISO_ERR_HIP, std::string("iso_ic_create: pack kernel: ") + hipGetErrorString(e));
    }
    for (int d = 0; d < 3; ++d) ic->g3.ax[d] = model_grid->ax[d];
    ic->g3.hot = ic->d_hot;
    ic->g3.s1 = model_grid->shape[2];
    ic->g3.s0 = model_grid->shape[1] * model_grid->shape[2];
    for (int d = 0; d < 4; ++d) ic->g4.ax[d] = bc_grid->ax[d];
    ic->g4.tab = bc_grid->d_grid;
    ic->g4.ncol = (int)bc_grid->shape[4];
    ic->g4.s2 = bc_grid->shape[3];
    ic->g4.s1 = bc_grid->shape[2] * bc_grid->shape[3];
    ic->g4.s0 = bc_grid->shape[1] * bc_grid->shape[2] * bc_grid->shape[3];
    ic->lds_doubles = assign_lds(ic->g3.ax, 3, ic->g4.ax, 4);
    for (int d = 0; d < 3; ++d) ic->h_axes_model[d] = model_grid->h_axes[d];
    for (int d = 0; d < 4; ++d) ic->h_axes_bc[d] = bc_grid->h_axes[d];
    ic->d_hotq = nullptr;
    ic->d_astq = nullptr;
    ic->g3.hotq = ic->g3.astq = nullptr;
    ic->g4.tabq = nullptr;
    // (cell indices travel as 32-bit integers through the cooperative gather)
    // Built for every table the generic kernel can meet as well (non-uniform EEP axis, ISOCHRONES_AMD_PATH=generic):
    // its lane-per-sample gathers read the same pack.  "compact" keeps every kernel on the compact tables.
    if (path_mode() != PATH_COMPACT && model_grid->ncells < (int64_t(1) << 31) &&
        bc_grid->ncells < (int64_t(1) << 31)) {
        // corner-packed copy for the fast kernel: 8 corners x 6 columns per cell (384 B)
        e = pack_corners(ic->d_hot, HOT_COLS, PACK_COLS, 3, model_grid->shape, &ic->d_hotq);
        if (e != hipSuccess) {
            ic->d_hotq = nullptr;      // not fatal: the compact table serves the fast kernel too
            (void)hipGetLastError();
        }
    }
    *out = ic;
    return ISO_OK;
}

void iso_ic_destroy(iso_ic* ic)
{
    if (!ic) return;
    DeviceGuard guard(ic->device);
    service_stop(ic->ctx, false);
    service_forget(ic);
    if (ic->d_hot) (void)hipFree(ic->d_hot);
    if (ic->d_hotq) (void)hipFree(ic->d_hotq);
    if (ic->d_astq) (void)hipFree(ic->d_astq);
    for (MagPack& mp : ic->mag_packs) free_mag_pack(mp);
    delete ic;
}

int iso_interp_mag(iso_ic* ic, const double* pars, int64_t stride_n, int64_t stride_p, int64_t n,
                   const int32_t* bc_cols, int nb, double* Teff, double* logg, double* feh, double* mags,
                   void* stream)
{
    if (!ic || (!pars && n > 0)) return fail(ISO_ERR_INVALID, "iso_interp_mag: NULL argument");
    if (nb < 0 || nb > ISO_MAX_BANDS) return fail(ISO_ERR_INVALID, "iso_interp_mag: nb out of range");
    if (nb > 0 && !bc_cols) return fail(ISO_ERR_INVALID, "iso_interp_mag: bc_cols is NULL");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_interp_mag: n < 0");
    if (n == 0) return ISO_OK;
    MagArgs A;
    std::memset(&A, 0, sizeof(A));
    A.g3 = ic->g3;
    A.g4 = ic->g4;
    A.kind = ic->kind;
    A.pars = pars;
    A.stride_n = stride_n;
    A.stride_p = stride_p;
    A.n = n;
    A.nb = nb;
    for (int b = 0; b < nb; ++b) {
        if (bc_cols[b] < 0 || bc_cols[b] >= ic->g4.ncol)
            return fail(ISO_ERR_INVALID, "iso_interp_mag: band column out of range");
        A.bc_cols[b] = bc_cols[b];
    }
    A.Teff = Teff; A.logg = logg; A.feh = feh;
    A.mags = nb > 0 ? mags : nullptr;
    DeviceGuard guard(ic->ctx->device);
    if (nb >= 1 && nb <= 12 && mags && ic->d_hotq && third_axis_ok(ic) && path_mode() == PATH_AUTO) {
        // large batches: corner-packed tables + wave-cooperative gathers (the pack for this band list
        // is built once and kept); small ones are not worth building a pack for
        FastArgs F;
        const int rc = acquire_mag_pack(ic, bc_cols, nb, n, F);
        if (rc < 0) return rc;
        if (rc == 1) {
            F.pars = pars;
            F.stride_n = stride_n;
            F.stride_p = stride_p;
            F.n = n;
            MagOut O{Teff, logg, feh, mags};
            if (launch_interp_mag_fast(ic->kind, nb, F, O, as_stream(stream))) {
                HIP_TRY(hipGetLastError());
                return ISO_OK;
            }
        }
    }
    const int lanes_per_sample = nb > 1 ? (nb + 1) / 2 : 1, samples_per_wave = 64 / lanes_per_sample;
    const int64_t waves = (n + samples_per_wave - 1) / samples_per_wave;
    const dim3 g(grid_blocks(waves * 64)), b(BLOCK);
    const size_t shmem = (size_t)ic->lds_doubles * sizeof(double);
    note_kernel("k_interp_mag<%d>", ic->kind == ISO_KIND_TRACK ? ISO_KIND_TRACK : ISO_KIND_ISO);
    if (ic->kind == ISO_KIND_TRACK) hipLaunchKernelGGL(k_interp_mag<ISO_KIND_TRACK>, g, b, shmem, as_stream(stream), A);
    else hipLaunchKernelGGL(k_interp_mag<ISO_KIND_ISO>, g, b, shmem, as_stream(stream), A);
    HIP_TRY(hipGetLastError());
    return ISO_OK;
}

}  // extern "C" (helpers follow)

namespace {

int validate_desc(const iso_ic* ic, const iso_model_desc* desc, const char* who)
{
    const std::string w(who);
    if (desc->n_stars < 1 || desc->n_stars > ISO_MAX_STARS) return fail(ISO_ERR_INVALID, w + ": n_stars");
    if (desc->n_stars > 1 && ic->kind == ISO_KIND_TRACK)
        return fail(ISO_ERR_INVALID, w + ": multiple stars need the isochrone parametrisation");
    if (desc->n_bands < 0 || desc->n_bands > ISO_MAX_BANDS) return fail(ISO_ERR_INVALID, w + ": n_bands");
    if (ic->prior_cols[0] < 0 || ic->prior_cols[1] < 0)
        return fail(ISO_ERR_INVALID, w + ": the model table has no EEP-prior columns");
    if (desc->has_numax && (ic->astero_cols[0] < 0 || ic->astero_cols[1] < 0))
        return fail(ISO_ERR_INVALID, w + ": the model table has no nu_max/delta_nu columns");
    for (int b = 0; b < desc->n_bands; ++b)
        if (desc->bc_cols[b] < 0 || desc->bc_cols[b] >= ic->g4.ncol)
            return fail(ISO_ERR_INVALID, w + ": band column out of range");
    const iso_prior* pr[5] = {&desc->prior_mass, &desc->prior_age, &desc->prior_feh, &desc->prior_distance,
                              &desc->prior_AV};
    for (int j = 0; j < 5; ++j)
        if (!prior_kind_ok(pr[j]->kind)) return fail(ISO_ERR_INVALID, w + ": unknown prior family");
    return ISO_OK;
}

// observations + prior constants of one system, everything constant pre-evaluated on the host
void fill_dev_model(const iso_model_desc* desc, int kind, DevModel& H)
{
    std::memset(&H, 0, sizeof(H));
    H.n_stars = desc->n_stars;
    H.n_bands = desc->n_bands;
    H.kind = kind;
    H.has_parallax = desc->has_parallax;
    H.has_numax = desc->has_numax;
    H.has_dnu = desc->has_numax ? desc->has_dnu : 0;
    for (int b = 0; b < desc->n_bands; ++b) {
        H.mag_val[b] = desc->mag_val[b];
        gauss_consts(desc->mag_unc[b], H.mag_g0[b], H.mag_unc2[b], &H.mag_hinv[b]);
    }
    for (int q = 0; q < 3; ++q) {
        H.spec_val[q] = desc->spec_val[q];
        gauss_consts(desc->spec_unc[q], H.spec_g0[q], H.spec_unc2[q], &H.spec_hinv[q]);
    }
    H.plx_val = desc->plx_val;
    gauss_consts(desc->plx_unc, H.plx_g0, H.plx_unc2, &H.plx_hinv);
    H.numax_val = desc->numax_val;
    gauss_consts(desc->numax_unc, H.numax_g0, H.numax_unc2, &H.numax_hinv);
    H.dnu_val = desc->dnu_val;
    gauss_consts(desc->dnu_unc, H.dnu_g0, H.dnu_unc2, &H.dnu_hinv);
    H.prior_mass = make_dev_prior(desc->prior_mass);
    H.prior_age = make_dev_prior(desc->prior_age);
    H.prior_feh = make_dev_prior(desc->prior_feh);
    H.prior_distance = make_dev_prior(desc->prior_distance);
    H.prior_AV = make_dev_prior(desc->prior_AV);
    H.eep_lo = desc->eep_lo;
    H.eep_hi = desc->eep_hi;
    for (int j = 0; j < ISO_MAX_PARAMS; ++j) {
        H.bound_lo[j] = desc->bound_lo[j];
        H.bound_hi[j] = desc->bound_hi[j];
    }
}

// BC table restricted to `nb` bands (observation order), contiguous per cell
hipError_t pack_bands(const iso_ic* ic, const int32_t* bc_cols, int nb, double** out)
{
    const int64_t ncells = ic->bc->ncells;
    hipError_t e = hipMalloc(out, (size_t)ncells * nb * sizeof(double));
    if (e != hipSuccess) return e;
    PackBcArgs P;
    P.grid = ic->bc->d_grid;
    P.ncol = ic->g4.ncol;
    P.nb = nb;
    P.ncells = ncells;
    for (int b = 0; b < nb; ++b) P.src[b] = bc_cols[b];
    P.out = *out;
    note_kernel("k_pack_bc");
    hipLaunchKernelGGL(k_pack_bc, dim3(grid_blocks(ncells * nb)), dim3(BLOCK), 0, 0, P);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e;
}

bool fast_eligible(const iso_ic* ic, const iso_model_desc* desc)
{
    // asteroseismic terms are only on the corner-packed form of the fast path
    // 13-32 bands: the band-tiled batch kernel, on the corner-packed tables only and without asteroseismic terms
    const bool bands_ok = desc->n_bands >= 0 && (desc->n_bands <= FAST_NB_MAX ||
                                                (desc->n_bands <= ISO_MAX_BANDS && path_mode() == PATH_AUTO && ic->d_hotq && !desc->has_numax));
    return path_mode() != PATH_GENERIC && bands_ok &&
           (!desc->has_numax || (path_mode() == PATH_AUTO && ic->d_hotq)) && third_axis_ok(ic);
}

// staged axes (+ reciprocal spacings) and, when the interpolator has a corner-packed model table,
// the corner-packed BC; fills F (without m / pars / outputs).  *ok = false if not representable.
hipError_t build_fast(const iso_ic* ic, int nb, const double* d_bc_hot, double** d_axes_blob, double** d_bcq,
                      FastArgs& F, bool* ok)
{
    *ok = false;
    std::vector<double> blob;
    FastAxis fa[6];
    std::memset(fa, 0, sizeof(fa));
    const std::vector<double>* src[6] = {&ic->h_axes_model[0], &ic->h_axes_model[1], &ic->h_axes_bc[0],
                                         &ic->h_axes_bc[1], &ic->h_axes_bc[2], &ic->h_axes_bc[3]};
    for (int a = 0; a < 6; ++a) {
        const std::vector<double>& v = *src[a];
        fa[a].off = (int)blob.size();
        fa[a].n = (int)v.size();
        blob.insert(blob.end(), v.begin(), v.end());
        for (size_t j = 0; j + 1 < v.size(); ++j) blob.push_back(1.0 / (v[j + 1] - v[j]));
        blob.push_back(0.0);
    }
    FastAxis coarse;
    std::memset(&coarse, 0, sizeof(coarse));
    const bool e_uniform = ic->model->ax[2].uniform != 0;
    if (!e_uniform) {
        const std::vector<double>& v = ic->h_axes_model[2];
        if (v.size() < 9) return hipSuccess;
        coarse.off = (int)blob.size();
        for (size_t j = 0; j < v.size(); j += 8) blob.push_back(v[j]);
        // the bucket table of this axis is entered with any x of the full axis: close it with the last node (a
        // coarse bracket there is clamped to the last window anyway)
        if ((v.size() - 1) % 8 != 0) blob.push_back(v.back());
        coarse.n = (int)blob.size() - coarse.off;
    }
    if ((int)blob.size() > FAST_MAX_BLOB) return hipSuccess;          // (the tables below add at most 1 KB to that)
    // bucket tables of the seven staged axes (fast/axis_lut.h), bytes behind the doubles.  Budget: what the
    // four-workgroups-per-CU persistent sampler has left of its 40 KB (axes + 256 gather slots of 7 doubles + 512
    // positions of 5 doubles; 736 B with the MIST axes), at most 1 KB.
    {
        std::vector<const std::vector<double>*> planned(src, src + 6);
        std::vector<double> coarse_nodes;
        if (!e_uniform) {
            coarse_nodes.assign(blob.begin() + coarse.off, blob.begin() + coarse.off + coarse.n);
            planned.push_back(&coarse_nodes);
        }
        const int spare = 6144 - (int)(blob.size() * sizeof(double));
        const int budget = spare >= 64 ? std::min(spare & ~7, 1024) : 512;
        const std::vector<AxisLutChoice> plan = axis_lut_plan(planned, budget);
        std::vector<uint8_t> bytes;
        for (size_t a = 0; a < planned.size(); ++a) {
            FastAxis& ax = a < 6 ? fa[a] : coarse;
            const AxisLutChoice& ch = plan[a];
            ax.lut = (int)(blob.size() * sizeof(double) + bytes.size());
            ax.shw = ch.sh | (ch.win << 5) | ((ch.nbk - 1) << 17);          // sh < 32, win <= n <= 2048, nbk <= 1024
            ax.chi = lut_hi32(ch.c);
            ax.b0 = ch.b0;
            bytes.resize(bytes.size() + (size_t)ch.nbk);
            axis_lut_fill(*planned[a], ch, bytes.data() + bytes.size() - (size_t)ch.nbk);
        }
        bytes.resize((bytes.size() + 7) & ~(size_t)7, 0);
        const size_t at = blob.size();
        blob.resize(at + bytes.size() / sizeof(double));
        std::memcpy(blob.data() + at, bytes.data(), bytes.size());
    }
    hipError_t e = hipMalloc(d_axes_blob, blob.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(*d_axes_blob, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    if (path_mode() == PATH_AUTO && ic->d_hotq && nb > 0) {
        hipError_t e2 = pack_corners(d_bc_hot, nb, nb, 4, ic->bc->shape, d_bcq);
        if (e2 != hipSuccess) {
            *d_bcq = nullptr;
            (void)hipGetLastError();
        }
    }
    std::memset(&F, 0, sizeof(F));
    F.m0 = fa[0]; F.m1 = fa[1];
    F.b0 = fa[2]; F.b1 = fa[3]; F.b2 = fa[4]; F.b3 = fa[5];
    F.e_n = ic->model->ax[2].n;
    if (e_uniform) {
        F.e_a0 = ic->model->ax[2].a0;
        F.e_step = ic->model->ax[2].step;
        F.e_inv = 1.0 / F.e_step;
        F.e_last = std::fma((double)(F.e_n - 1), F.e_step, F.e_a0);
        F.e_axis = nullptr;
    } else {
        F.e_a0 = ic->h_axes_model[2].front();
        F.e_last = ic->h_axes_model[2].back();
        F.e_step = F.e_inv = 0.0;
        F.ec = coarse;
        F.e_axis = ic->model->d_axes[2];
    }
    F.axes_blob = *d_axes_blob;
    F.axes_len = (int)blob.size();
    F.hot = ic->d_hot;
    F.hotq = ic->d_hotq;
    F.astq = nullptr;          // set by iso_model_create for asteroseismic models
    F.s0 = ic->g3.s0; F.s1 = ic->g3.s1;
    F.bc = d_bc_hot;
    F.bcq = *d_bcq;
    F.nb_total = nb;
    if (nb > FAST_NB_MAX && !*d_bcq) return hipSuccess;      // the band-tiled kernel exists on the packed tables only
    F.bs2 = ic->bc->shape[3];
    F.bs1 = ic->bc->shape[2] * ic->bc->shape[3];
    F.bs0 = ic->bc->shape[1] * ic->bc->shape[2] * ic->bc->shape[3];
    *ok = true;
    return hipSuccess;
}


// the shared pack of a band list (BandPack, iso_internal.h): found in the interpolator's cache or built and cached.
// The cache holds references to the most recently used packs up to a byte bound (ISOCHRONES_AMD_PACK_CACHE_MB, default 2048:
// a 12-band pack of the MIST BC grid is 654 MB) and at most 6 of them; a pack the cache lets go lives on with the catalogs
// that hold it and is freed with the last one.  The pack is built OUTSIDE the lock (a pass over the BC table and a device
// synchronise): concurrent catalog creation on other band lists does not queue behind it; two callers that build the same
// list at once both succeed and the second one's copy is dropped in favour of the cached one.
size_t band_pack_cache_bytes()
{
    if (const char* e = std::getenv("ISOCHRONES_AMD_PACK_CACHE_MB")) {
        const long mb = std::atol(e);
        if (mb >= 0) return (size_t)mb << 20;
    }
    return (size_t)2048 << 20;
}

static bool same_bands(const BandPack& bp, const int32_t* bc_cols, int nb)
{
    return (int)bp.cols.size() == nb && std::equal(bp.cols.begin(), bp.cols.end(), bc_cols);
}

hipError_t acquire_band_pack(iso_ic* ic, const int32_t* bc_cols, int nb, std::shared_ptr<BandPack>* out, bool* ok)
{
    *ok = false;
    auto lookup = [&]() -> bool {              // (with mag_mu held) most recently used last
        for (size_t k = 0; k < ic->band_packs.size(); ++k) {
            std::shared_ptr<BandPack> bp = ic->band_packs[k];
            if (same_bands(*bp, bc_cols, nb)) {
                ic->band_packs.erase(ic->band_packs.begin() + (long)k);
                ic->band_packs.push_back(bp);
                *out = bp;
                *ok = true;
                return true;
            }
        }
        return false;
    };
    {
        std::lock_guard<std::mutex> lock(ic->mag_mu);
        if (lookup()) return hipSuccess;
    }
    std::shared_ptr<BandPack> bp(new BandPack());
    bp->device = ic->device;
    bp->cols.assign(bc_cols, bc_cols + nb);
    bp->d_bcq = bp->d_axes_blob = nullptr;
    bp->bytes = (size_t)ic->bc->ncells * 16 * (size_t)nb * sizeof(double);
    double* d_bc_hot = nullptr;
    hipError_t e = pack_bands(ic, bc_cols, nb, &d_bc_hot);
    if (e == hipSuccess) e = build_fast(ic, nb, d_bc_hot, &bp->d_axes_blob, &bp->d_bcq, bp->fast, ok);
    if (d_bc_hot) (void)hipFree(d_bc_hot);               // the fused kernels read the corner-packed copy only
    bp->fast.bc = nullptr;
    if (e != hipSuccess || !*ok || !bp->d_bcq) {
        *ok = false;
        return e;
    }
    std::lock_guard<std::mutex> lock(ic->mag_mu);
    if (lookup()) return hipSuccess;                     // somebody else cached the same list meanwhile: theirs is used, ours goes
    const size_t bound = band_pack_cache_bytes();
    ic->band_packs.push_back(bp);
    size_t total = 0;
    for (const auto& q : ic->band_packs) total += q->bytes;
    while (ic->band_packs.size() > 1 && (ic->band_packs.size() > 6 || total > bound)) {       // the cache lets go; holders keep theirs
        total -= ic->band_packs.front()->bytes;
        ic->band_packs.erase(ic->band_packs.begin());
    }
    if (bp->bytes > bound) ic->band_packs.clear();       // a pack beyond the bound is not cached at all
    *out = bp;
    return hipSuccess;
}

void free_mag_pack(MagPack& mp)
{
    if (mp.d_bc_hot) (void)hipFree(mp.d_bc_hot);
    if (mp.d_bcq) (void)hipFree(mp.d_bcq);
    if (mp.d_axes_blob) (void)hipFree(mp.d_axes_blob);
    mp.d_bc_hot = mp.d_bcq = mp.d_axes_blob = nullptr;
}

// Corner-packed BC table of a band list for iso_interp_mag.  Returns 1 and fills F when a pack exists (or
// the batch is large enough to pay for building one: a pack costs one pass over the BC table), 0 when
// the caller should use the generic kernel, < 0 on error.  At most MAG_PACK_SLOTS band lists are kept
// (least recently used one is dropped; hipFree synchronises with kernels still reading it).
const size_t MAG_PACK_SLOTS = 6;
const int64_t MAG_PACK_BUILD_MIN_ROWS = 32768, MAG_PACK_USE_MIN_ROWS = 1024;

int acquire_mag_pack(iso_ic* ic, const int32_t* bc_cols, int nb, int64_t n, FastArgs& F)
{
    if (n < MAG_PACK_USE_MIN_ROWS) return 0;
    std::lock_guard<std::mutex> lock(ic->mag_mu);
    for (MagPack& mp : ic->mag_packs)
        if ((int)mp.cols.size() == nb && std::equal(mp.cols.begin(), mp.cols.end(), bc_cols)) {
            mp.last_use = ++ic->mag_clock;
            F = mp.fast;
            return 1;
        }
    if (n < MAG_PACK_BUILD_MIN_ROWS) return 0;
    MagPack mp;
    mp.cols.assign(bc_cols, bc_cols + nb);
    mp.d_bc_hot = mp.d_bcq = mp.d_axes_blob = nullptr;
    bool ok = false;
    hipError_t e = pack_bands(ic, bc_cols, nb, &mp.d_bc_hot);
    if (e == hipSuccess) e = build_fast(ic, nb, mp.d_bc_hot, &mp.d_axes_blob, &mp.d_bcq, mp.fast, &ok);
    if (e != hipSuccess || !ok || !mp.d_bcq) {
        free_mag_pack(mp);
        if (e == hipErrorOutOfMemory || e == hipSuccess) {     // no room / not representable: generic kernel
            (void)hipGetLastError();
            return 0;
        }
        return fail(ISO_ERR_HIP, std::string("iso_interp_mag: ") + hipGetErrorString(e));
    }
    (void)hipFree(mp.d_bc_hot);                                  // only the corner-packed copy is read
    mp.d_bc_hot = nullptr;
    mp.fast.bc = nullptr;
    if (ic->mag_packs.size() >= MAG_PACK_SLOTS) {
        size_t lru = 0;
        for (size_t k = 1; k < ic->mag_packs.size(); ++k)
            if (ic->mag_packs[k].last_use < ic->mag_packs[lru].last_use) lru = k;
        free_mag_pack(ic->mag_packs[lru]);
        ic->mag_packs.erase(ic->mag_packs.begin() + lru);
    }
    mp.last_use = ++ic->mag_clock;
    F = mp.fast;
    ic->mag_packs.push_back(mp);
    return 1;
}
}  // namespace

extern "C" {

int iso_model_create(iso_ic* ic, const iso_model_desc* desc, iso_model** out)
{
    if (!ic || !desc || !out) return fail(ISO_ERR_INVALID, "iso_model_create: NULL argument");
    const int rc = validate_desc(ic, desc, "iso_model_create");
    if (rc != ISO_OK) return rc;

    DeviceGuard guard(ic->ctx->device);
    iso_model* m = new (std::nothrow) iso_model();
    if (!m) return fail(ISO_ERR_NOMEM, "iso_model_create: out of host memory");
    m->ic = ic;
    m->device = ic->device;
    m->desc = *desc;
    m->d_model = nullptr;
    m->d_bc_hot = nullptr;
    m->d_bcq = nullptr;
    m->d_axes_blob = nullptr;
    m->fast_ok = false;
    m->h_stage = nullptr;
    m->mbox = m->d_mbox = nullptr;
    m->mbox_stream = nullptr;
    m->mbox_count = 0;
    m->mbox_state = 0;
    m->stage_rows = 0;
    m->stage_seq = 0;
    m->d_pipe = nullptr;
    m->h_pipe = nullptr;
    m->pipe_rows = 0;
    m->pipe_stream[0] = m->pipe_stream[1] = nullptr;

    DevModel H;
    fill_dev_model(desc, ic->kind, H);
    hipError_t e = hipMalloc(&m->d_model, sizeof(DevModel));
    if (e == hipSuccess) e = hipMemcpy(m->d_model, &H, sizeof(DevModel), hipMemcpyHostToDevice);

    m->g4 = ic->g4;
    if (e == hipSuccess && desc->n_bands > 0) {
        e = pack_bands(ic, desc->bc_cols, desc->n_bands, &m->d_bc_hot);
        m->g4.tab = m->d_bc_hot;
        m->g4.ncol = desc->n_bands;
    }
    if (e == hipSuccess && fast_eligible(ic, desc)) {
        e = build_fast(ic, desc->n_bands, m->d_bc_hot, &m->d_axes_blob, &m->d_bcq, m->fast, &m->fast_ok);
        m->fast.m = m->d_model;
        if (e == hipSuccess && m->fast_ok && desc->has_numax) {
            // (nu_max, delta_nu) = hot columns 6, 7, corner-packed once per interpolator (128 B per cell)
            std::lock_guard<std::mutex> lock(ic->mag_mu);
            if (!ic->d_astq && (m->d_bcq || desc->n_bands == 0)) {
                hipError_t e2 = pack_corners(ic->d_hot, HOT_COLS, 2, 3, ic->model->shape, &ic->d_astq, 6);
                if (e2 != hipSuccess) {
                    ic->d_astq = nullptr;
                    (void)hipGetLastError();
                }
            }
            m->fast.astq = ic->d_astq;
            if (!m->fast.astq || (!m->d_bcq && desc->n_bands > 0)) m->fast_ok = false;     // generic kernel
        }
    }
    // the fused kernels exist on the corner-packed tables only (round 4: the compact-table instantiations, 52 kernels that
    // nothing but ISOCHRONES_AMD_PATH=compact selected, are gone): without the packs the generic kernel evaluates the model
    if (m->fast_ok && (!m->fast.hotq || (!m->fast.bcq && desc->n_bands > 0))) m->fast_ok = false;
    if (e == hipSuccess && !m->fast_ok && ic->d_hotq) {
        // generic kernel: give its lane-per-sample gathers the corner-packed forms too (whole-line reads), whatever
        // made the model miss the fast path (> 12 bands, ISOCHRONES_AMD_PATH=generic)
        const size_t bcq_bytes = (size_t)ic->bc->ncells * 16 * (size_t)std::max(desc->n_bands, 1) * sizeof(double);
        if (desc->n_bands > 0 && !m->d_bcq && bcq_bytes <= (size_t(4) << 30)) {
            hipError_t e2 = pack_corners(m->d_bc_hot, desc->n_bands, desc->n_bands, 4, ic->bc->shape, &m->d_bcq);
            if (e2 != hipSuccess) {
                m->d_bcq = nullptr;
                (void)hipGetLastError();
            }
        }
        if (desc->has_numax) {
            std::lock_guard<std::mutex> lock(ic->mag_mu);
            if (!ic->d_astq) {
                hipError_t e2 = pack_corners(ic->d_hot, HOT_COLS, 2, 3, ic->model->shape, &ic->d_astq, 6);
                if (e2 != hipSuccess) {
                    ic->d_astq = nullptr;
                    (void)hipGetLastError();
                }
            }
        }
    }
    m->g4.tabq = (!m->fast_ok && ic->d_hotq) ? m->d_bcq : nullptr;
    if (e != hipSuccess) {
        std::string msg = std::string("iso_model_create: ") + hipGetErrorString(e);
        iso_model_destroy(m);
        return fail(e == hipErrorOutOfMemory ? ISO_ERR_NOMEM : ISO_ERR_HIP, msg);
    }
    *out = m;
    return ISO_OK;
}

void mailbox_stop(iso_model* m, bool release);

void iso_model_destroy(iso_model* m)
{
    if (!m) return;
    DeviceGuard guard(m->device);
    mailbox_stop(m, true);            // the resident wave reads the tables below (and hipFree would wait for it anyway)
    if (m->d_model) (void)hipFree(m->d_model);
    if (m->d_bc_hot) (void)hipFree(m->d_bc_hot);
    if (m->d_bcq) (void)hipFree(m->d_bcq);
    if (m->d_axes_blob) (void)hipFree(m->d_axes_blob);
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    if (m->d_pipe) (void)hipFree(m->d_pipe);
    if (m->h_pipe) (void)hipHostFree(m->h_pipe);
    for (hipStream_t st : m->pipe_stream)
        if (st) (void)hipStreamDestroy(st);
    delete m;
}

int iso_model_n_params(const iso_model* m) { return m ? m->desc.n_stars + 4 : ISO_ERR_INVALID; }

int iso_axis_bracket_host(const double* const* axes, const int32_t* n_nodes, int n_axes, int budget, int which,
                          const double* x, int64_t n, int32_t* index_out, int32_t* plan_out)
{
    if (!axes || !n_nodes || n_axes < 1 || which < 0 || which >= n_axes || (n > 0 && (!x || !index_out)))
        return fail(ISO_ERR_INVALID, "iso_axis_bracket_host: bad arguments");
    std::vector<std::vector<double>> ax(n_axes);
    std::vector<const std::vector<double>*> ptr(n_axes);
    for (int a = 0; a < n_axes; ++a) {
        if (n_nodes[a] < 2) return fail(ISO_ERR_INVALID, "iso_axis_bracket_host: an axis needs two nodes");
        ax[a].assign(axes[a], axes[a] + n_nodes[a]);
        ptr[a] = &ax[a];
    }
    const std::vector<AxisLutChoice> plan = axis_lut_plan(ptr, budget);
    for (int a = 0; plan_out && a < n_axes; ++a) {
        plan_out[5 * a + 0] = plan[a].nbk;
        plan_out[5 * a + 1] = plan[a].win;
        plan_out[5 * a + 2] = plan[a].levels;
        plan_out[5 * a + 3] = plan[a].sh;
        plan_out[5 * a + 4] = plan[a].b0;
    }
    const AxisLutChoice& ch = plan[which];
    const std::vector<double>& v = ax[which];
    std::vector<uint8_t> tab((size_t)ch.nbk);
    axis_lut_fill(v, ch, tab.data());
    const int nn = (int)v.size();
    for (int64_t k = 0; k < n; ++k) {                 // the statements of lut_start() + lds_bracket() (fast/brackets.h)
        const int b = lut_bucket(x[k], ch.c, ch.sh, ch.b0);
        int base = tab[(size_t)std::max(0, std::min(b, ch.nbk - 1))], len = ch.win;
        while (len > 1) {
            const int half = len >> 1;
            base = (v[(size_t)(base + half)] <= x[k]) ? base + half : base;
            len -= half;
        }
        index_out[k] = std::min(base, nn - 2);
    }
    return ISO_OK;
}

int iso_model_kernel_path(const iso_model* m)
{
    if (!m) return ISO_ERR_INVALID;
    if (!m->fast_ok) return ISO_PATH_GENERIC;
    return ISO_PATH_FUSED_PACKED;       // (ISO_PATH_FUSED_COMPACT: rounds 1-3; the fused kernels read the corner-packed tables only now)
}

}  // extern "C"

namespace {

template <int KIND, int NS>
void launch_lnpost_nb(int nb, dim3 g, dim3 b, size_t shmem, hipStream_t s, const PostArgs& A)
{
    // The generic kernel is the FALL-BACK for what the fused families do not take (a model table whose third axis is not
    // uniform, tables too large for 32-bit cell numbers, ISOCHRONES_AMD_PATH=generic / compact) and the reference-order
    // arithmetic the fused kernels are debugged against - not a throughput path (151 us per 10^6 rows where the fused
    // kernel takes 73).  Round 6: one form per (parametrisation, stars), the band loop at run time; the eight compile-time
    // band counts per shape (32 instantiations, 8-10 % on a path nobody times) are gone.
    (void)nb;
    note_kernel("k_lnpost<%d, %d, 0>", KIND, NS);
    hipLaunchKernelGGL((k_lnpost<KIND, NS, 0>), g, b, shmem, s, A);
}

void launch_lnpost(const iso_model* m, dim3 g, dim3 b, size_t shmem, hipStream_t s, const PostArgs& A)
{
    const int nb = m->desc.n_bands;
    if (m->ic->kind == ISO_KIND_TRACK) {
        launch_lnpost_nb<ISO_KIND_TRACK, 1>(nb, g, b, shmem, s, A);
    } else {
        switch (m->desc.n_stars) {
        case 1: launch_lnpost_nb<ISO_KIND_ISO, 1>(nb, g, b, shmem, s, A); break;
        case 2: launch_lnpost_nb<ISO_KIND_ISO, 2>(nb, g, b, shmem, s, A); break;
        default: launch_lnpost_nb<ISO_KIND_ISO, 3>(nb, g, b, shmem, s, A); break;
        }
    }
}

int enqueue_lnpost(iso_model* m, const double* pars, int64_t stride_n, int64_t stride_p, int64_t n,
                   double* lnpost_out, double* lnprior_out, double* lnlike_out, hipStream_t s,
                   unsigned long long* done_flag = nullptr, unsigned long long done_seq = 0)
{
    if (m->fast_ok) {
        FastArgs F = m->fast;
        F.done_flag = done_flag;        // single-workgroup host callbacks only (iso_lnpost_host)
        F.done_seq = done_seq;
        F.pars = pars;
        F.stride_n = stride_n;
        F.stride_p = stride_p;
        F.n = n;
        F.lnpost = lnpost_out;
        F.lnprior = lnprior_out;
        // lnprior alone still needs the likelihood flag off; lnlike requested -> evaluate everywhere
        F.lnlike = lnlike_out;
        if (launch_lnpost_fast(m->ic->kind, m->desc.n_stars, m->desc.n_bands, false, F, s)) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(ISO_ERR_HIP, std::string("iso_lnpost (fast) launch: ") + hipGetErrorString(e));
            return ISO_OK;
        }
    }
    PostArgs A;
    A.g3 = m->ic->g3;
    A.g3.hotq = m->ic->d_hotq;
    A.g3.astq = m->ic->d_astq;
    A.g4 = m->g4;
    A.m = m->d_model;
    A.pars = pars;
    A.stride_n = stride_n;
    A.stride_p = stride_p;
    A.n = n;
    A.lnpost = lnpost_out;
    A.lnprior = lnprior_out;
    A.lnlike = lnlike_out;
    const dim3 g((unsigned)((n + BLOCK - 1) / BLOCK)), b(BLOCK);        // one sample per lane (k_lnpost has no grid-stride loop)
    const size_t shmem = (size_t)m->ic->lds_doubles * sizeof(double);
    launch_lnpost(m, g, b, shmem, s, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ISO_ERR_HIP, std::string("iso_lnpost launch: ") + hipGetErrorString(e));
    return ISO_OK;
}

}  // namespace

extern "C" {

int iso_lnpost(iso_model* m, const double* pars, int64_t stride_n, int64_t stride_p, int64_t n,
               double* lnpost_out, double* lnprior_out, double* lnlike_out, void* stream)
{
    if (!m || (!pars && n > 0)) return fail(ISO_ERR_INVALID, "iso_lnpost: NULL argument");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_lnpost: n < 0");
    if (!lnpost_out && !lnprior_out && !lnlike_out) return fail(ISO_ERR_INVALID, "iso_lnpost: no output requested");
    if (n == 0) return ISO_OK;
    DeviceGuard guard(m->ic->ctx->device);
    return enqueue_lnpost(m, pars, stride_n, stride_p, n, lnpost_out, lnprior_out, lnlike_out, as_stream(stream));
}

// Large host batches: the rows are cut into chunks; the calling thread moves chunk k to the device and launches its
// kernel, whose results go straight into pinned, device-mapped host memory (no separate download); a second thread
// copies the results of finished chunks into the caller's arrays - so those copies (and the page faults of a freshly
// allocated result array) overlap the uploads of the later chunks.  The call costs about the 40 B/row upload alone.
static int lnpost_host_pipelined(iso_model* m, const double* pars, int64_t n, double* lnpost_out, double* lnprior_out,
                                 double* lnlike_out)
{
    const int np_ = m->desc.n_stars + 4;
    int64_t CH = int64_t(1) << 17;
    if (m->pipe_rows < n) {
        if (m->d_pipe) (void)hipFree(m->d_pipe);
        if (m->h_pipe) (void)hipHostFree(m->h_pipe);
        m->d_pipe = m->h_pipe = nullptr;
        m->pipe_rows = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_pipe), sizeof(double) * (size_t)n * np_));
        // coherent (fine-grained): the copy-out thread reads results a kernel wrote while later kernels are still running
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_pipe), sizeof(double) * (size_t)n * 3,
                              hipHostMallocMapped | hipHostMallocCoherent));
        m->pipe_rows = n;
    }
    if (!m->pipe_stream[0]) HIP_TRY(hipStreamCreateWithFlags(&m->pipe_stream[0], hipStreamNonBlocking));
    double* d_pars = m->d_pipe;
    double* d_res = nullptr;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_res), m->h_pipe, 0));
    double* d_out[3] = {d_res, d_res + m->pipe_rows, d_res + 2 * m->pipe_rows};
    const double* staged[3] = {m->h_pipe, m->h_pipe + m->pipe_rows, m->h_pipe + 2 * m->pipe_rows};
    double* h_out[3] = {lnpost_out, lnprior_out, lnlike_out};
    const int64_t nchunks = (n + CH - 1) / CH;
    std::vector<hipEvent_t> ev((size_t)nchunks, nullptr);
    for (hipEvent_t& e : ev) {
        const hipError_t ce = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (ce != hipSuccess) {
            for (hipEvent_t& made : ev)
                if (made) (void)hipEventDestroy(made);
            return fail(ISO_ERR_HIP, std::string("iso_lnpost_host: hipEventCreate: ") + hipGetErrorString(ce));
        }
    }
    std::atomic<int64_t> issued{0};
    std::atomic<int> worker_err{(int)hipSuccess};
    std::atomic<bool> abort_flag{false};
    const int device = m->device;
    std::thread worker;
    auto copy_out = [&] {
        (void)hipSetDevice(device);
        for (int64_t k = 0; k < nchunks; ++k) {
            while (issued.load(std::memory_order_acquire) <= k) {
                if (abort_flag.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            const hipError_t e = hipEventSynchronize(ev[(size_t)k]);
            if (e != hipSuccess) {
                worker_err.store((int)e);
                return;
            }
            const int64_t off = k * CH, c = std::min<int64_t>(CH, n - off);
            for (int o = 0; o < 3; ++o)
                if (h_out[o]) std::memcpy(h_out[o] + off, staged[o] + off, sizeof(double) * c);
        }
    };
    try {
        worker = std::thread(copy_out);
    } catch (const std::exception& ex) {                   // no thread to be had: nothing crosses the C ABI as an exception
        for (hipEvent_t& ev_k : ev) (void)hipEventDestroy(ev_k);
        return fail(ISO_ERR_NOMEM, std::string("iso_lnpost_host: cannot start the copy-out thread: ") + ex.what());
    }
    int rc = ISO_OK;
    hipError_t e = hipSuccess;
    for (int64_t k = 0; k < nchunks && rc == ISO_OK && e == hipSuccess; ++k) {
        const int64_t off = k * CH, c = std::min<int64_t>(CH, n - off);
        // blocking copy from pageable memory: measured faster than hipMemcpyAsync on the same rows (chunked 10^6 x 5:
        // 0.98 against 1.16 ms per call); the kernels run on a non-blocking stream, so nothing else is serialised
        e = hipMemcpy(d_pars + off * np_, pars + off * np_, sizeof(double) * c * np_, hipMemcpyHostToDevice);
        if (e != hipSuccess) break;
        rc = enqueue_lnpost(m, d_pars + off * np_, np_, 1, c, lnpost_out ? d_out[0] + off : nullptr,
                            lnprior_out ? d_out[1] + off : nullptr, lnlike_out ? d_out[2] + off : nullptr, m->pipe_stream[0]);
        if (rc != ISO_OK) break;
        e = hipEventRecord(ev[(size_t)k], m->pipe_stream[0]);
        if (e == hipSuccess) issued.store(k + 1, std::memory_order_release);
    }
    if (rc != ISO_OK || e != hipSuccess) abort_flag.store(true, std::memory_order_release);
    worker.join();
    (void)hipStreamSynchronize(m->pipe_stream[0]);
    for (hipEvent_t& ev_k : ev) (void)hipEventDestroy(ev_k);
    // the staging areas of an ordinary batch stay with the model for the next call; those of a very large one
    // (> 4 Mi rows: 160 MB of device + 96 MB of pinned memory and up) go back at once
    if (m->pipe_rows > (int64_t(1) << 22)) {
        (void)hipFree(m->d_pipe);
        (void)hipHostFree(m->h_pipe);
        m->d_pipe = m->h_pipe = nullptr;
        m->pipe_rows = 0;
    }
    if (rc != ISO_OK) return rc;
    if (e == hipSuccess) e = (hipError_t)worker_err.load();
    if (e != hipSuccess) return fail(ISO_ERR_HIP, std::string("iso_lnpost_host: ") + hipGetErrorString(e));
    return ISO_OK;
}

// ---- the per-point callback through the model's resident mailbox wave (iso_fast_mailbox.hip) --------------------------------
// ISOCHRONES_AMD_MAILBOX=0 keeps every call on the launch path; ISOCHRONES_AMD_MAILBOX_IDLE_US (default 1000) is how long the
// wave stays without a request - the longest a device-wide synchronise elsewhere in the process can be held up by it.
namespace {
constexpr double WALL_CLOCK_HZ = 1.0e8;          // wall_clock64(): the constant 100 MHz counter

inline unsigned long long mb_load(const volatile unsigned long long* p)
{
    return __atomic_load_n(p, __ATOMIC_ACQUIRE);
}

bool mailbox_enabled()
{
    const char* e = std::getenv("ISOCHRONES_AMD_MAILBOX");
    return !(e && e[0] == '0');
}

bool mailbox_launch(iso_model* m)
{
    double idle_us = 1000.0;
    if (const char* e = std::getenv("ISOCHRONES_AMD_MAILBOX_IDLE_US")) idle_us = std::max(10.0, std::atof(e));
    const unsigned long long idle = (unsigned long long)(idle_us * 1e-6 * WALL_CLOCK_HZ);
    const unsigned long long life = (unsigned long long)(30.0 * WALL_CLOCK_HZ);       // 30 s whatever happens
    __atomic_store_n(&m->mbox->ctl[1], 0ull, __ATOMIC_RELAXED);
    __atomic_store_n(&m->mbox->ctl[0], 1ull, __ATOMIC_RELEASE);                    // running (the wave writes 2 when it leaves)
    if (!launch_mailbox(m->ic->kind, m->desc.n_stars, m->desc.n_bands, m->fast, m->d_mbox, idle, life, m->mbox_stream) ||
        hipGetLastError() != hipSuccess) {
        __atomic_store_n(&m->mbox->ctl[0], 2ull, __ATOMIC_RELEASE);
        return false;
    }
    return true;
}

// lazily: the pinned mailbox, its stream; false = this model has no mailbox (the caller takes the launch path)
bool mailbox_ready(iso_model* m)
{
    if (m->mbox_state < 0) return false;
    if (m->mbox_state > 0) return true;
    m->mbox_state = -1;
    if (!m->fast_ok || !m->fast.hotq || (!m->fast.bcq && m->desc.n_bands > 0) || m->fast.astq || m->desc.n_bands > FAST_NB_MAX) return false;
    if (hipHostMalloc(reinterpret_cast<void**>(&m->mbox), sizeof(IsoMailbox), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        m->mbox = nullptr;
        return false;
    }
    std::memset(m->mbox, 0, sizeof(IsoMailbox));
    m->mbox->ctl[0] = 2;                             // no wave yet
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&m->d_mbox), m->mbox, 0) != hipSuccess ||
        hipStreamCreateWithFlags(&m->mbox_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(m->mbox);
        m->mbox = nullptr;
        return false;
    }
    m->mbox_state = 1;
    return true;
}

// n <= ISO_MAILBOX_ROWS rows through the resident wave; ISO_OK, or 1 = not served (the caller launches instead)
int mailbox_call(iso_model* m, const double* pars, int n, double* lnpost_out, double* lnprior_out, double* lnlike_out)
{
    IsoMailbox* mb = m->mbox;
    const int np_ = m->desc.n_stars + 4;
    const bool parts = lnprior_out || lnlike_out;
    unsigned long long seq = ((++m->mbox_count & 0xFFFFull) << 16) | ((unsigned long long)parts << 8) | (unsigned long long)(n - 1);
    if (n == 1) {
        unsigned long long words[ISO_MAX_PARAMS];
        for (int q = 0; q < np_; ++q) {
            std::memcpy(&words[q], pars + q, 8);
            __atomic_store_n(&mb->req[1 + q], words[q], __ATOMIC_RELAXED);
        }
        // the wave accepts the request only with parameter words that give this number: a line read in pieces (new sequence
        // word, old parameters) is polled again instead of evaluated
        seq |= (unsigned long long)mailbox_checksum(words, np_) << 32;
    } else {
        std::memcpy(mb->rows, pars, sizeof(double) * (size_t)n * np_);
    }
    __atomic_store_n(&mb->req[0], seq, __ATOMIC_RELEASE);        // the sequence word last
    if (mb_load(&mb->ctl[0]) != 1 && !mailbox_launch(m)) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 1; mb_load(&mb->done[0]) != seq; ++spins) {
        if ((spins & 255) == 0) {
            // the wave may have left (idle / lifetime) between our look at the state and its last poll: start another one,
            // which finds the request waiting.  A wave that neither answers nor leaves within 2 s is a fault.
            if (mb_load(&mb->ctl[0]) == 2 && mb_load(&mb->done[0]) != seq) {
                if (!mailbox_launch(m)) return 1;
            } else if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                __atomic_store_n(&mb->ctl[1], 1ull, __ATOMIC_RELEASE);
                (void)hipStreamSynchronize(m->mbox_stream);
                m->mbox_state = -1;
                return 1;
            }
        }
    }
    if (n == 1) {
        double r[3];
        for (int k = 0; k < 3; ++k) {
            const unsigned long long w = __atomic_load_n(&mb->done[1 + k], __ATOMIC_RELAXED);
            std::memcpy(&r[k], &w, 8);
        }
        if (lnpost_out) *lnpost_out = r[0];
        if (lnprior_out) *lnprior_out = r[1];
        if (lnlike_out) *lnlike_out = r[2];
#ifdef ISO_MAILBOX_CLOCK        // (variant builds: the wave's own time from seeing a request to its results, 100 MHz ticks)
        {
            static unsigned long long calls = 0, ticks = 0;
            ticks += __atomic_load_n(&mb->done[4], __ATOMIC_RELAXED);
            if (++calls % 2000 == 0) {
                std::fprintf(stderr, "model mailbox: %.2f us on the device per call (%llu calls)\n", ticks * 0.01 / (double)calls, calls);
                calls = ticks = 0;
            }
        }
#endif
    } else {
        if (lnpost_out) std::memcpy(lnpost_out, mb->out, sizeof(double) * n);
        if (lnprior_out) std::memcpy(lnprior_out, mb->out + ISO_MAILBOX_ROWS, sizeof(double) * n);
        if (lnlike_out) std::memcpy(lnlike_out, mb->out + 2 * ISO_MAILBOX_ROWS, sizeof(double) * n);
    }
    return ISO_OK;
}
}  // namespace

// ask the model's resident wave to leave and wait until it has (before its tables go, or before a device-wide synchronise
// that should not wait for the idle timeout); frees the mailbox when `release`
void mailbox_stop(iso_model* m, bool release)
{
    if (!m->mbox) return;
    if (mb_load(&m->mbox->ctl[0]) == 1) {
        __atomic_store_n(&m->mbox->ctl[1], 1ull, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(m->mbox_stream);
    }
    if (release) {
        (void)hipStreamSynchronize(m->mbox_stream);
        (void)hipStreamDestroy(m->mbox_stream);
        (void)hipHostFree(m->mbox);
        m->mbox = m->d_mbox = nullptr;
        m->mbox_stream = nullptr;
        m->mbox_state = 0;
    }
}

int iso_lnpost_host(iso_model* m, const double* pars, int64_t n, double* lnpost_out, double* lnprior_out,
                    double* lnlike_out)
{
    if (!m || (!pars && n > 0)) return fail(ISO_ERR_INVALID, "iso_lnpost_host: NULL argument");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_lnpost_host: n < 0");
    if (!lnpost_out && !lnprior_out && !lnlike_out) return fail(ISO_ERR_INVALID, "iso_lnpost_host: no output requested");
    if (n == 0) return ISO_OK;
    DeviceGuard guard(m->device);
    std::lock_guard<std::mutex> lock(m->host_mu);       // ctypes drops the GIL: two Python threads may call one model
    const int np_ = m->desc.n_stars + 4;
    // a sampler's per-point callback (one row, or a few): the model's resident mailbox wave - no launch
    if (n <= ISO_MAILBOX_ROWS && mailbox_enabled() && mailbox_ready(m)) {
        const int rc = mailbox_call(m, pars, (int)n, lnpost_out, lnprior_out, lnlike_out);
        if (rc <= 0) return rc;
    }
    constexpr int64_t CAP = 8192;
    if (n > 4 * CAP) return lnpost_host_pipelined(m, pars, n, lnpost_out, lnprior_out, lnlike_out);
    if (!m->h_stage) {
        // pinned + mapped: the kernel reads the parameters and writes the results straight through
        // PCIe — one launch + one synchronise per call, no separate copies (+ 8 doubles for the completion flag)
        // coherent (fine-grained) host memory: the host spins on a flag the kernel raises behind its results while the
        // kernel may still be running - visibility and ordering of both must not depend on HIP_HOST_COHERENT
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_stage), sizeof(double) * (CAP * (np_ + 3) + 8),
                              hipHostMallocMapped | hipHostMallocCoherent));
        m->stage_rows = CAP;
        m->stage_seq = 0;
        m->h_stage[CAP * (np_ + 3)] = 0.0;
    }
    double* h_pars = m->h_stage;
    double* h_post = h_pars + CAP * np_;
    double* h_prior = h_post + CAP;
    double* h_like = h_prior + CAP;
    double *d_pars = nullptr;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_pars), h_pars, 0));
    double* d_post = d_pars + CAP * np_;
    double* d_prior = d_post + CAP;
    double* d_like = d_prior + CAP;
    volatile unsigned long long* h_flag = reinterpret_cast<volatile unsigned long long*>(h_like + CAP);
    unsigned long long* d_flag = reinterpret_cast<unsigned long long*>(d_like + CAP);
    for (int64_t done = 0; done < n; done += CAP) {
        const int64_t c = std::min<int64_t>(CAP, n - done);
        std::memcpy(h_pars, pars + done * np_, sizeof(double) * c * np_);
        // one workgroup on the fused kernel (a sampler's scalar / half-ensemble callback): the kernel raises a flag in
        // mapped memory after its results and the host spins on it; everything else synchronises the stream
        const bool flagged = m->fast_ok && c <= BLOCK && !getenv("ISOCHRONES_AMD_HOST_SYNC");
        const unsigned long long seq = ++m->stage_seq;
        const int rc = enqueue_lnpost(m, d_pars, np_, 1, c, lnpost_out ? d_post : nullptr, lnprior_out ? d_prior : nullptr,
                                      lnlike_out ? d_like : nullptr, nullptr, flagged ? d_flag : nullptr, seq);
        if (rc != ISO_OK) return rc;
        bool seen = false;
        if (flagged) {
            const auto t0 = std::chrono::steady_clock::now();
            for (uint64_t spins = 0; !(seen = (*h_flag == seq)); ++spins) {
                // a kernel that never raises the flag (a fault) must not hang the caller: after 2 ms fall back to the
                // stream synchronise, which also reports the error
                if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) HIP_TRY(hipStreamSynchronize(nullptr));
        if (lnpost_out) std::memcpy(lnpost_out + done, h_post, sizeof(double) * c);
        if (lnprior_out) std::memcpy(lnprior_out + done, h_prior, sizeof(double) * c);
        if (lnlike_out) std::memcpy(lnlike_out + done, h_like, sizeof(double) * c);
    }
    return ISO_OK;
}

int iso_unit_cube(iso_model* m, double* cube, int64_t stride_n, int64_t stride_p, int64_t n, void* stream)
{
    if (!m || (!cube && n > 0)) return fail(ISO_ERR_INVALID, "iso_unit_cube: NULL argument");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_unit_cube: n < 0");
    if (n == 0) return ISO_OK;
    DeviceGuard guard(m->ic->ctx->device);
    note_kernel("k_unit_cube");
    hipLaunchKernelGGL(k_unit_cube, dim3(grid_blocks(n * (m->desc.n_stars + 4))), dim3(BLOCK), 0, as_stream(stream),
                       m->d_model, cube, stride_n, stride_p, n);
    HIP_TRY(hipGetLastError());
    return ISO_OK;
}

int iso_eep_table_create(iso_ctx* ctx, const double* ages, const int64_t* lengths, const double* ax0, int64_t n0,
                         const double* ax1, int64_t n1, int64_t n_eep, double eep0, iso_eep_table** out)
{
    if (!ctx || !ages || !lengths || !ax0 || !ax1 || !out) return fail(ISO_ERR_INVALID, "iso_eep_table_create: NULL argument");
    if (n0 < 2 || n1 < 2 || n_eep < 1) return fail(ISO_ERR_INVALID, "iso_eep_table_create: bad shape");
    for (int64_t j = 1; j < n0; ++j)
        if (!(ax0[j - 1] < ax0[j])) return fail(ISO_ERR_INVALID, "iso_eep_table_create: axis 0 not increasing");
    for (int64_t j = 1; j < n1; ++j)
        if (!(ax1[j - 1] < ax1[j])) return fail(ISO_ERR_INVALID, "iso_eep_table_create: axis 1 not increasing");
    for (int64_t j = 0; j < n0 * n1; ++j)
        if (lengths[j] < 0 || lengths[j] > n_eep) return fail(ISO_ERR_INVALID, "iso_eep_table_create: bad track length");
    DeviceGuard guard(ctx->device);
    iso_eep_table* t = new (std::nothrow) iso_eep_table();
    if (!t) return fail(ISO_ERR_NOMEM, "iso_eep_table_create: out of host memory");
    t->device = ctx->device;
    t->ctx = ctx;
    t->n0 = n0; t->n1 = n1; t->n_eep = n_eep; t->eep0 = eep0;
    t->d_ages = t->d_ax0 = t->d_ax1 = nullptr;
    t->d_lengths = nullptr;
    const size_t nt = (size_t)(n0 * n1);
    hipError_t e = hipMalloc(&t->d_ages, nt * n_eep * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(t->d_ages, ages, nt * n_eep * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&t->d_lengths, nt * sizeof(int64_t));
    if (e == hipSuccess) e = hipMemcpy(t->d_lengths, lengths, nt * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&t->d_ax0, n0 * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(t->d_ax0, ax0, n0 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&t->d_ax1, n1 * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(t->d_ax1, ax1, n1 * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        std::string msg = std::string("iso_eep_table_create: ") + hipGetErrorString(e);
        iso_eep_table_destroy(t);
        return fail(e == hipErrorOutOfMemory ? ISO_ERR_NOMEM : ISO_ERR_HIP, msg);
    }
    t->ax[0] = AxisD{t->d_ax0, (int)n0, -1, 0, 0.0, 0.0};
    t->ax[1] = AxisD{t->d_ax1, (int)n1, -1, 0, 0.0, 0.0};
    (void)assign_lds(t->ax, 2, nullptr, 0);
    *out = t;
    return ISO_OK;
}

void iso_eep_table_destroy(iso_eep_table* t)
{
    if (!t) return;
    DeviceGuard guard(t->device);
    service_stop(t->ctx, false);
    service_forget(t);
    if (t->d_ages) (void)hipFree(t->d_ages);
    if (t->d_lengths) (void)hipFree(t->d_lengths);
    if (t->d_ax0) (void)hipFree(t->d_ax0);
    if (t->d_ax1) (void)hipFree(t->d_ax1);
    delete t;
}

int iso_interp_eep(iso_eep_table* t, const double* x, const double* x0, const double* x1, int64_t n, double* out,
                   void* stream)
{
    if (!t || ((!x || !x0 || !x1 || !out) && n > 0)) return fail(ISO_ERR_INVALID, "iso_interp_eep: NULL argument");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_interp_eep: n < 0");
    if (n == 0) return ISO_OK;
    EepArgs A;
    A.ax[0] = t->ax[0];
    A.ax[1] = t->ax[1];
    A.ages = t->d_ages;
    A.lengths = t->d_lengths;
    A.n1 = (int)t->n1;
    A.n_eep = t->n_eep;
    A.eep0 = t->eep0;
    A.x = x; A.x0 = x0; A.x1 = x1;
    A.n = n;
    A.out = out;
    int lds = 0;
    for (int d = 0; d < 2; ++d)
        if (A.ax[d].lds_off >= 0) lds += A.ax[d].n;
    DeviceGuard guard(t->device);
    note_kernel("k_interp_eep");
    hipLaunchKernelGGL(k_interp_eep, dim3(grid_blocks(n)), dim3(BLOCK), (size_t)lds * sizeof(double), as_stream(stream), A);
    HIP_TRY(hipGetLastError());
    return ISO_OK;
}


// ---- the scalar accessors through the context's resident service wave (kernels/k_service.h) ------------------------------------
// ISOCHRONES_AMD_MAILBOX=0 keeps every call on the launch path (as for the per-point lnpost callback).
namespace {
struct SvcTargetRec {
    SvcTarget* d_target;
};
struct iso_service {
    IsoSvcBox* box = nullptr;      // pinned, device-mapped
    IsoSvcBox* d_box = nullptr;
    hipStream_t stream = nullptr;
    unsigned long long count = 0;
    int state = 0;                 // 0 untried, 1 usable, -1 not available
};
std::mutex g_svc_mu;                                             // guards the two maps below (calls are serialised per context
std::unordered_map<iso_ctx*, iso_service*> g_services;           // by ctx->stage_mu, which every *_host entry point holds)
std::unordered_map<const void*, SvcTargetRec> g_svc_targets;     // iso_table* / iso_ic* / iso_eep_table* -> its device record

inline unsigned long long svc_host_load(const volatile unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }

bool service_enabled()
{
    const char* e = std::getenv("ISOCHRONES_AMD_MAILBOX");
    return !(e && e[0] == '0');
}

iso_service* service_of(iso_ctx* ctx)
{
    std::lock_guard<std::mutex> lock(g_svc_mu);
    iso_service*& sv = g_services[ctx];
    if (!sv) sv = new iso_service();
    if (sv->state != 0) return sv->state > 0 ? sv : nullptr;
    sv->state = -1;
    if (hipHostMalloc(reinterpret_cast<void**>(&sv->box), sizeof(IsoSvcBox), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        sv->box = nullptr;
        return nullptr;
    }
    std::memset(sv->box, 0, sizeof(IsoSvcBox));
    sv->box->ctl[0] = 2;                                         // no wave yet
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&sv->d_box), sv->box, 0) != hipSuccess ||
        hipStreamCreateWithFlags(&sv->stream, hipStreamNonBlocking) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_service, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ISO_SVC_LDS_DOUBLES * sizeof(double))) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(sv->box);
        sv->box = nullptr;
        return nullptr;
    }
    sv->state = 1;
    return sv;
}

bool service_launch(iso_service* sv)
{
    double idle_us = 1000.0;
    if (const char* e = std::getenv("ISOCHRONES_AMD_MAILBOX_IDLE_US")) idle_us = std::max(10.0, std::atof(e));
    const unsigned long long idle = (unsigned long long)(idle_us * 1e-6 * 1.0e8);       // wall_clock64(): 100 MHz
    const unsigned long long life = (unsigned long long)(30.0 * 1.0e8);
    __atomic_store_n(&sv->box->ctl[1], 0ull, __ATOMIC_RELAXED);
    __atomic_store_n(&sv->box->ctl[0], 1ull, __ATOMIC_RELEASE);
    note_kernel("k_service");
    hipLaunchKernelGGL(k_service, dim3(1), dim3(64), (size_t)ISO_SVC_LDS_DOUBLES * sizeof(double), sv->stream, sv->d_box, idle, life);
    if (hipGetLastError() != hipSuccess) {
        __atomic_store_n(&sv->box->ctl[0], 2ull, __ATOMIC_RELEASE);
        return false;
    }
    return true;
}

// ask the context's wave to leave and wait until it has; `release` frees the mailbox
void service_stop(iso_ctx* ctx, bool release)
{
    iso_service* sv = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_svc_mu);
        auto it = g_services.find(ctx);
        if (it == g_services.end()) return;
        sv = it->second;
        if (release) g_services.erase(it);
    }
    if (sv->box) {
        if (svc_host_load(&sv->box->ctl[0]) == 1) __atomic_store_n(&sv->box->ctl[1], 1ull, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(sv->stream);
    }
    if (release) {
        if (sv->box) {
            (void)hipStreamDestroy(sv->stream);
            (void)hipHostFree(sv->box);
        }
        delete sv;
    }
}

// the device record of a target (created on first use); 0 = could not be created
bool service_target(const void* key, const SvcTarget& host, SvcTargetRec* out)
{
    std::lock_guard<std::mutex> lock(g_svc_mu);
    auto it = g_svc_targets.find(key);
    if (it != g_svc_targets.end()) {
        *out = it->second;
        return true;
    }
    SvcTargetRec rec;
    if (hipMalloc(reinterpret_cast<void**>(&rec.d_target), sizeof(SvcTarget)) != hipSuccess ||
        hipMemcpy(rec.d_target, &host, sizeof(SvcTarget), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    g_svc_targets[key] = rec;
    *out = rec;
    return true;
}

// (before the object's tables are freed)
void service_forget(const void* key)
{
    SvcTargetRec rec{nullptr};
    {
        std::lock_guard<std::mutex> lock(g_svc_mu);
        auto it = g_svc_targets.find(key);
        if (it == g_svc_targets.end()) return;
        rec = it->second;
        g_svc_targets.erase(it);
    }
    (void)hipFree(rec.d_target);      // (a device-wide wait: a resident wave has left by its idle time-out)
}

// one request; ISO_OK, or 1 = not served (the caller launches instead).  Caller holds ctx->stage_mu.
int service_call(iso_ctx* ctx, int op, const SvcTargetRec& tgt, const double* x, int nx, const int32_t* cols, int k, double* out,
                 int nout)
{
    iso_service* sv = service_of(ctx);
    if (!sv) return 1;
    IsoSvcBox* mb = sv->box;
    unsigned long long words[10];
    std::memset(words, 0, sizeof words);
    words[0] = (unsigned long long)(uintptr_t)tgt.d_target;
    for (int q = 0; q < nx; ++q) std::memcpy(&words[1 + q], x + q, 8);
    for (int c = 0; c < k; ++c) words[6 + (c >> 3)] |= (unsigned long long)(cols[c] & 0xFF) << (8 * (c & 7));
    const unsigned long long seq = ((unsigned long long)mailbox_checksum(words, 10) << 32) | ((++sv->count & 0xFFFFull) << 16) |
                                   ((unsigned long long)k << 8) | (unsigned long long)op;
    if (k > 8)
        for (int q = 7; q < 10; ++q) __atomic_store_n(&mb->req[1 + q], words[q], __ATOMIC_RELAXED);    // behind the line, before it
    for (int q = 0; q < 7; ++q) __atomic_store_n(&mb->req[1 + q], words[q], __ATOMIC_RELAXED);
    __atomic_store_n(&mb->req[0], seq, __ATOMIC_RELEASE);        // the sequence word last
    if (svc_host_load(&mb->ctl[0]) != 1 && !service_launch(sv)) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 1; svc_host_load(&mb->done[0]) != seq; ++spins) {
        if ((spins & 255) == 0) {
            if (svc_host_load(&mb->ctl[0]) == 2 && svc_host_load(&mb->done[0]) != seq) {
                if (!service_launch(sv)) return 1;               // the wave left between our look at its state and its last poll
            } else if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                __atomic_store_n(&mb->ctl[1], 1ull, __ATOMIC_RELEASE);
                (void)hipStreamSynchronize(sv->stream);
                sv->state = -1;
                return 1;
            }
        }
    }
    for (int q = 0; q < nout; ++q) {
        const unsigned long long w = __atomic_load_n(reinterpret_cast<unsigned long long*>(&mb->out[q]), __ATOMIC_RELAXED);
        std::memcpy(out + q, &w, 8);
    }
    return ISO_OK;
}
}  // namespace

namespace {
// the context's pinned, device-mapped staging area (host view + device view); caller holds ctx->stage_mu
int ctx_stage(iso_ctx* ctx, double** host, double** dev)
{
    if (!ctx->h_stage) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), sizeof(double) * (ISO_CTX_STAGE_DOUBLES + 8),
                              hipHostMallocMapped | hipHostMallocCoherent));
        ctx->h_stage[ISO_CTX_STAGE_DOUBLES] = 0.0;
        ctx->stage_seq = 0;
    }
    *host = ctx->h_stage;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(dev), ctx->h_stage, 0));
    return ISO_OK;
}

// Completion of the launches a *_host entry point has put on the default stream: a one-thread kernel behind them
// stores a sequence number into the mapped staging area and the host spins on it - 10 us instead of the 13 us of a
// hipStreamSynchronize per round trip (tools/sync_probe.hip).  After 2 ms without the flag (a fault, a very large
// batch) the stream is synchronised, which also surfaces the error.
__global__ void k_signal_done(volatile unsigned long long* flag, unsigned long long seq)
{
    __threadfence_system();
    *flag = seq;
    __threadfence_system();
}

int ctx_wait(iso_ctx* ctx, double* h, double* d)
{
    if (getenv("ISOCHRONES_AMD_HOST_SYNC")) {
        HIP_TRY(hipStreamSynchronize(nullptr));
        return ISO_OK;
    }
    const unsigned long long seq = ++ctx->stage_seq;
    volatile unsigned long long* h_flag = reinterpret_cast<volatile unsigned long long*>(h + ISO_CTX_STAGE_DOUBLES);
    note_kernel("k_signal_done");
    hipLaunchKernelGGL(k_signal_done, dim3(1), dim3(1), 0, nullptr,
                       reinterpret_cast<volatile unsigned long long*>(d + ISO_CTX_STAGE_DOUBLES), seq);
    HIP_TRY(hipGetLastError());
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (uint64_t spins = 0; !(seen = (*h_flag == seq)); ++spins)
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) HIP_TRY(hipStreamSynchronize(nullptr));
    return ISO_OK;
}
}  // namespace

int iso_interp_host(iso_table* t, const double* x, int64_t n, const int32_t* icols, int k, double* out)
{
    if (!t || !icols || ((!x || !out) && n > 0)) return fail(ISO_ERR_INVALID, "iso_interp_host: NULL argument");
    if (k < 1 || k > ISO_MAX_COLS) return fail(ISO_ERR_INVALID, "iso_interp_host: k out of range");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_interp_host: n < 0");
    if (n == 0) return ISO_OK;
    DeviceGuard guard(t->device);
    std::lock_guard<std::mutex> lock(t->ctx->stage_mu);
    if (n == 1 && k <= ISO_SVC_MAX_COLS && t->shape[t->ndim] <= 256 && service_enabled()) {
        // one point: the context's resident service wave - no launch (kernels/k_service.h)
        bool cols_ok = true;
        for (int c = 0; c < k; ++c) cols_ok = cols_ok && icols[c] >= 0 && icols[c] < t->shape[t->ndim];
        if (!cols_ok) return fail(ISO_ERR_INVALID, "iso_interp: column index out of range");
        SvcTarget T;
        std::memset(&T, 0, sizeof T);
        T.op = ISO_SVC_INTERP;
        T.ndim = t->ndim;
        for (int dd = 0; dd < ISO_MAX_DIM; ++dd) {
            if (dd < t->ndim) T.I.ax[dd] = t->ax[dd];
            T.I.ax[dd].lds_off = -1;
        }
        (void)assign_lds(T.I.ax, t->ndim, nullptr, 0);            // exactly as iso_interp stages them
        int64_t st = 1;
        for (int dd = t->ndim - 1; dd >= 0; --dd) {
            T.I.stride[dd] = st;
            st *= t->shape[dd];
        }
        T.I.grid = t->d_grid;
        T.I.ncol = (int)t->shape[t->ndim];
        SvcTargetRec rec;
        if (service_target(t, T, &rec) && service_call(t->ctx, ISO_SVC_INTERP, rec, x, t->ndim, icols, k, out, k) == ISO_OK)
            return ISO_OK;
    }
    double *h = nullptr, *d = nullptr;
    int rc = ctx_stage(t->ctx, &h, &d);
    if (rc != ISO_OK) return rc;
    const int nd = t->ndim;
    const int64_t cap = ISO_CTX_STAGE_DOUBLES / (nd + k);
    for (int64_t done = 0; done < n; done += cap) {
        const int64_t c = std::min<int64_t>(cap, n - done);
        const double* xp[ISO_MAX_DIM];
        for (int dd = 0; dd < nd; ++dd) {                       // row-major host rows -> one contiguous vector per axis
            for (int64_t i = 0; i < c; ++i) h[dd * c + i] = x[(done + i) * nd + dd];
            xp[dd] = d + dd * c;
        }
        rc = iso_interp(t, xp, c, icols, k, d + nd * c, nullptr);
        if (rc != ISO_OK) return rc;
        rc = ctx_wait(t->ctx, h, d);
        if (rc != ISO_OK) return rc;
        std::memcpy(out + done * k, h + nd * c, sizeof(double) * c * k);
    }
    return ISO_OK;
}

int iso_interp_mag_host(iso_ic* ic, const double* pars, int64_t n, const int32_t* bc_cols, int nb, double* Teff,
                        double* logg, double* feh, double* mags)
{
    if (!ic || (!pars && n > 0)) return fail(ISO_ERR_INVALID, "iso_interp_mag_host: NULL argument");
    if (nb < 0 || nb > ISO_MAX_BANDS) return fail(ISO_ERR_INVALID, "iso_interp_mag_host: nb out of range");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_interp_mag_host: n < 0");
    if (n == 0) return ISO_OK;
    DeviceGuard guard(ic->device);
    std::lock_guard<std::mutex> lock(ic->ctx->stage_mu);
    if (n == 1 && ic->g4.ncol <= 256 && service_enabled()) {
        // one point: the context's resident service wave (the generic column-parallel evaluation, which is also what a
        // one-point launch runs: no band pack is built for a small call)
        if (nb > 0 && !bc_cols) return fail(ISO_ERR_INVALID, "iso_interp_mag: bc_cols is NULL");
        for (int b = 0; b < nb; ++b)
            if (bc_cols[b] < 0 || bc_cols[b] >= ic->g4.ncol) return fail(ISO_ERR_INVALID, "iso_interp_mag: band column out of range");
        SvcTarget T;
        std::memset(&T, 0, sizeof T);
        T.op = ISO_SVC_MAG;
        T.kind = ic->kind;
        T.M.g3 = ic->g3;
        T.M.g4 = ic->g4;
        T.M.kind = ic->kind;
        SvcTargetRec rec;
        double o[3 + ISO_MAX_BANDS];
        const int want = (mags && nb > 0) ? nb : 0;
        if (service_target(ic, T, &rec) && service_call(ic->ctx, ISO_SVC_MAG, rec, pars, 5, bc_cols, want, o, 3 + want) == ISO_OK) {
            if (Teff) Teff[0] = o[0];
            if (logg) logg[0] = o[1];
            if (feh) feh[0] = o[2];
            for (int b = 0; b < want; ++b) mags[b] = o[3 + b];
            return ISO_OK;
        }
    }
    double *h = nullptr, *d = nullptr;
    int rc = ctx_stage(ic->ctx, &h, &d);
    if (rc != ISO_OK) return rc;
    const int64_t cap = ISO_CTX_STAGE_DOUBLES / (5 + 3 + (nb > 0 ? nb : 1));
    for (int64_t done = 0; done < n; done += cap) {
        const int64_t c = std::min<int64_t>(cap, n - done);
        std::memcpy(h, pars + done * 5, sizeof(double) * c * 5);
        double *dT = d + 5 * c, *dg = dT + c, *df = dg + c, *dm = df + c;
        rc = iso_interp_mag(ic, d, 5, 1, c, bc_cols, nb, dT, dg, df, (mags && nb > 0) ? dm : nullptr, nullptr);
        if (rc != ISO_OK) return rc;
        rc = ctx_wait(ic->ctx, h, d);
        if (rc != ISO_OK) return rc;
        if (Teff) std::memcpy(Teff + done, h + 5 * c, sizeof(double) * c);
        if (logg) std::memcpy(logg + done, h + 6 * c, sizeof(double) * c);
        if (feh) std::memcpy(feh + done, h + 7 * c, sizeof(double) * c);
        if (mags && nb > 0) std::memcpy(mags + done * nb, h + 8 * c, sizeof(double) * c * nb);
    }
    return ISO_OK;
}

int iso_interp_eep_host(iso_eep_table* t, const double* age, const double* feh, const double* mass, int64_t n, double* out)
{
    if (!t || ((!age || !feh || !mass || !out) && n > 0)) return fail(ISO_ERR_INVALID, "iso_interp_eep_host: NULL argument");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_interp_eep_host: n < 0");
    if (n == 0) return ISO_OK;
    DeviceGuard guard(t->device);
    std::lock_guard<std::mutex> lock(t->ctx->stage_mu);
    if (n == 1 && service_enabled()) {
        SvcTarget T;
        std::memset(&T, 0, sizeof T);
        T.op = ISO_SVC_EEP;
        T.E.ax[0] = t->ax[0];
        T.E.ax[1] = t->ax[1];
        T.E.ages = t->d_ages;
        T.E.lengths = t->d_lengths;
        T.E.n1 = (int)t->n1;
        T.E.n_eep = t->n_eep;
        T.E.eep0 = t->eep0;
        SvcTargetRec rec;
        const double xin[3] = {age[0], feh[0], mass[0]};
        if (service_target(t, T, &rec) && service_call(t->ctx, ISO_SVC_EEP, rec, xin, 3, nullptr, 0, out, 1) == ISO_OK) return ISO_OK;
    }
    double *h = nullptr, *d = nullptr;
    int rc = ctx_stage(t->ctx, &h, &d);
    if (rc != ISO_OK) return rc;
    const int64_t cap = ISO_CTX_STAGE_DOUBLES / 4;
    for (int64_t done = 0; done < n; done += cap) {
        const int64_t c = std::min<int64_t>(cap, n - done);
        std::memcpy(h, age + done, sizeof(double) * c);
        std::memcpy(h + c, feh + done, sizeof(double) * c);
        std::memcpy(h + 2 * c, mass + done, sizeof(double) * c);
        rc = iso_interp_eep(t, d, d + c, d + 2 * c, c, d + 3 * c, nullptr);
        if (rc != ISO_OK) return rc;
        rc = ctx_wait(t->ctx, h, d);
        if (rc != ISO_OK) return rc;
        std::memcpy(out + done, h + 3 * c, sizeof(double) * c);
    }
    return ISO_OK;
}


// ---- a small pool of device blocks for the per-fit buffers of a catalog -------------------------------------------------------
// A catalog fit of a few thousand stars is a few milliseconds, and it used to start with two hipMalloc and end with two hipFree
// (each a device-wide synchronise plus ~0.1 ms in the driver): a tenth of the 3.3 ms of a 1 250-star fit.  Blocks given back are
// kept (at most POOL_BLOCKS of them, POOL_BYTES in total, per device) and handed out again to the next request they fit;
// giving a block back waits for the device first - whatever stream read it has finished, as with hipFree.
namespace {
struct PoolBlock {
    void* p;
    size_t bytes;
    int device;
};
std::mutex g_pool_mu;
std::vector<PoolBlock> g_pool;
constexpr size_t POOL_BLOCKS = 8, POOL_BYTES = (size_t)512 << 20;

hipError_t pool_alloc(void** out, size_t bytes, int device, size_t* got)
{
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        size_t best = g_pool.size();
        for (size_t k = 0; k < g_pool.size(); ++k)
            if (g_pool[k].device == device && g_pool[k].bytes >= bytes && g_pool[k].bytes <= 2 * bytes + 4096 &&
                (best == g_pool.size() || g_pool[k].bytes < g_pool[best].bytes))
                best = k;
        if (best < g_pool.size()) {
            *out = g_pool[best].p;
            *got = g_pool[best].bytes;
            g_pool.erase(g_pool.begin() + (long)best);
            return hipSuccess;
        }
    }
    *got = bytes;
    return hipMalloc(out, bytes);
}

void pool_free(void* p, size_t bytes, int device)
{
    if (!p) return;
    (void)hipDeviceSynchronize();
    void* drop = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        size_t total = bytes;
        for (const PoolBlock& b : g_pool) total += b.bytes;
        if (bytes > POOL_BYTES / 2 || total > POOL_BYTES || g_pool.size() >= POOL_BLOCKS) {
            if (bytes <= POOL_BYTES / 2 && !g_pool.empty()) {      // make room: the oldest block goes
                drop = g_pool.front().p;
                g_pool.erase(g_pool.begin());
                g_pool.push_back(PoolBlock{p, bytes, device});
            } else {
                drop = p;
            }
        } else {
            g_pool.push_back(PoolBlock{p, bytes, device});
        }
    }
    if (drop) (void)hipFree(drop);
}
}  // namespace

int iso_catalog_create(iso_ic* ic, const iso_model_desc* descs, int64_t n_models, iso_catalog** out)
{
    if (!ic || !descs || !out || n_models < 1) return fail(ISO_ERR_INVALID, "iso_catalog_create: bad argument");
    const iso_model_desc& d0 = descs[0];
    for (int64_t k = 0; k < n_models; ++k) {
        const int rc = validate_desc(ic, &descs[k], "iso_catalog_create");
        if (rc != ISO_OK) return rc;
        if (descs[k].n_stars != d0.n_stars || descs[k].n_bands != d0.n_bands ||
            std::memcmp(descs[k].bc_cols, d0.bc_cols, sizeof(int32_t) * d0.n_bands) != 0)
            return fail(ISO_ERR_INVALID, "iso_catalog_create: every star needs the same multiplicity and bands");
        if (descs[k].has_numax) return fail(ISO_ERR_INVALID, "iso_catalog_create: asteroseismic terms are not batched");
    }
    if (!fast_eligible(ic, &d0) || !ic->d_hotq || d0.n_bands < 1 || d0.n_bands > FAST_NB_MAX)
        return fail(ISO_ERR_INVALID, "iso_catalog_create: needs 1-12 bands, a third model axis of at least 9 nodes (or an exactly uniform one) and the corner-packed "
                                     "tables (ISOCHRONES_AMD_PATH=auto)");
    DeviceGuard guard(ic->device);
    iso_catalog* c = new (std::nothrow) iso_catalog();
    if (!c) return fail(ISO_ERR_NOMEM, "iso_catalog_create: out of host memory");
    c->device = ic->device;
    c->ic = ic;
    c->n_models = n_models;
    c->n_stars = d0.n_stars;
    c->n_bands = d0.n_bands;
    c->d_models = nullptr;
    c->models_bytes = 0;
    c->d_bc_hot = c->d_bcq = c->d_axes_blob = nullptr;
    std::vector<DevModel> H((size_t)n_models);
    for (int64_t k = 0; k < n_models; ++k) fill_dev_model(&descs[k], ic->kind, H[(size_t)k]);
    hipError_t e = hipMalloc(&c->d_models, sizeof(DevModel) * (size_t)n_models);
    if (e == hipSuccess) e = hipMemcpy(c->d_models, H.data(), sizeof(DevModel) * (size_t)n_models, hipMemcpyHostToDevice);
    bool ok = false;
    if (e == hipSuccess) e = acquire_band_pack(ic, d0.bc_cols, d0.n_bands, &c->pack, &ok);
    if (e == hipSuccess && !ok) {
        iso_catalog_destroy(c);
        return fail(ISO_ERR_INVALID, "iso_catalog_create: tables not representable on the fast path");
    }
    if (e == hipSuccess) c->fast = c->pack->fast;
    if (e != hipSuccess) {
        std::string msg = std::string("iso_catalog_create: ") + hipGetErrorString(e);
        iso_catalog_destroy(c);
        return fail(e == hipErrorOutOfMemory ? ISO_ERR_NOMEM : ISO_ERR_HIP, msg);
    }
    c->fast.m = c->d_models;
    // do all stars share the priors other than the distance prior (and the EEP bounds)?  Then the kernels read them from
    // the first star's block.
    {
        bool shared = true;
        const DevModel& h0 = H[0];
        for (int64_t k = 1; k < n_models && shared; ++k) {
            const DevModel& h = H[(size_t)k];
            shared = std::memcmp(&h.prior_mass, &h0.prior_mass, sizeof(DevPrior)) == 0 &&
                     std::memcmp(&h.prior_age, &h0.prior_age, sizeof(DevPrior)) == 0 &&
                     std::memcmp(&h.prior_feh, &h0.prior_feh, sizeof(DevPrior)) == 0 &&
                     std::memcmp(&h.prior_AV, &h0.prior_AV, sizeof(DevPrior)) == 0 &&
                     h.eep_lo == h0.eep_lo && h.eep_hi == h0.eep_hi;
        }
        c->fast.shared_priors = shared ? 1 : 0;
        if (const char* env = std::getenv("ISOCHRONES_AMD_SHARED_PRIORS")) c->fast.shared_priors = shared && std::atoi(env) != 0;
        bool all_default = true;                       // (the distance prior is per star: every star's family counts)
        for (int64_t k = 0; k < n_models && all_default; ++k) all_default = default_prior_families(descs[k]);
        c->std_priors = all_default ? 1 : 0;
    }
    c->packed = true;
    *out = c;
    return ISO_OK;
}

int iso_catalog_create_columns(iso_ic* ic, const iso_model_desc* tmpl, int64_t n_models, const double* mag_val,
                               const double* mag_unc, const double* spec_val, const double* spec_unc,
                               const int32_t* has_plx, const double* plx_val, const double* plx_unc,
                               const double* dist_hi, iso_catalog** out)
{
    if (!ic || !tmpl || !out || n_models < 1 || !mag_val || !mag_unc || !spec_val || !spec_unc || !has_plx || !plx_val ||
        !plx_unc)
        return fail(ISO_ERR_INVALID, "iso_catalog_create_columns: bad argument");
    const int rc = validate_desc(ic, tmpl, "iso_catalog_create_columns");
    if (rc != ISO_OK) return rc;
    if (tmpl->has_numax) return fail(ISO_ERR_INVALID, "iso_catalog_create_columns: asteroseismic terms are not batched");
    if (!fast_eligible(ic, tmpl) || !ic->d_hotq || tmpl->n_bands < 1 || tmpl->n_bands > FAST_NB_MAX)
        return fail(ISO_ERR_INVALID, "iso_catalog_create_columns: needs 1-12 bands, a third model axis of at least 9 nodes (or an exactly uniform one) and the "
                                     "corner-packed tables (ISOCHRONES_AMD_PATH=auto)");
    DeviceGuard guard(ic->device);
    iso_catalog* c = new (std::nothrow) iso_catalog();
    if (!c) return fail(ISO_ERR_NOMEM, "iso_catalog_create_columns: out of host memory");
    c->device = ic->device;
    c->ic = ic;
    c->n_models = n_models;
    c->n_stars = tmpl->n_stars;
    c->n_bands = tmpl->n_bands;
    c->d_models = nullptr;
    c->models_bytes = 0;
    c->d_bc_hot = c->d_bcq = c->d_axes_blob = nullptr;
    const int nb = tmpl->n_bands;
    DevModel H;
    fill_dev_model(tmpl, ic->kind, H);
    // one device block, one copy: [template block | columns | parallax flags]; the per-star blocks are filled from it by
    // two kernels on the null stream (the callers' work on any blocking stream is ordered behind them)
    const size_t n = (size_t)n_models;
    const size_t col_doubles = n * (2 * (size_t)nb + 6 + 2 + (dist_hi ? 1 : 0));
    const size_t tmpl_doubles = sizeof(DevModel) / sizeof(double);
    const size_t stage_bytes = (tmpl_doubles + col_doubles) * sizeof(double) + n * sizeof(int32_t);
    std::vector<double> stage((stage_bytes + 7) / 8);
    std::memcpy(stage.data(), &H, sizeof(DevModel));
    double* d_stage = nullptr;
    size_t stage_got = 0;
    hipError_t e = pool_alloc(reinterpret_cast<void**>(&c->d_models), sizeof(DevModel) * n, c->device, &c->models_bytes);
    if (e == hipSuccess) e = pool_alloc(reinterpret_cast<void**>(&d_stage), stage.size() * sizeof(double), c->device, &stage_got);
    FillCatalogArgs F;
    std::memset(&F, 0, sizeof(F));
    if (e == hipSuccess) {
        size_t off = tmpl_doubles;
        auto put = [&](const double* src, size_t count, const double** slot) {
            std::memcpy(stage.data() + off, src, count * sizeof(double));
            *slot = d_stage + off;
            off += count;
        };
        put(mag_val, n * nb, &F.mag_val);
        put(mag_unc, n * nb, &F.mag_unc);
        put(spec_val, n * 3, &F.spec_val);
        put(spec_unc, n * 3, &F.spec_unc);
        put(plx_val, n, &F.plx_val);
        put(plx_unc, n, &F.plx_unc);
        if (dist_hi) put(dist_hi, n, &F.dist_hi);
        std::memcpy(stage.data() + off, has_plx, n * sizeof(int32_t));
        F.has_plx = reinterpret_cast<const int32_t*>(d_stage + off);
        e = hipMemcpy(d_stage, stage.data(), stage.size() * sizeof(double), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
        F.models = c->d_models;
        F.tmpl = reinterpret_cast<const DevModel*>(d_stage);
        F.n = n_models;
        F.nb = nb;
        F.i_dist = tmpl->n_stars + 2;
        note_kernel("k_catalog_copy_template");
        hipLaunchKernelGGL(k_catalog_copy_template, dim3(grid_blocks(n_models * (int64_t)(sizeof(DevModel) / 8))), dim3(BLOCK),
                           0, 0, F);
        note_kernel("k_catalog_fill");
        hipLaunchKernelGGL(k_catalog_fill, dim3(grid_blocks(n_models)), dim3(BLOCK), 0, 0, F);
        e = hipGetLastError();
    }
    if (d_stage) pool_free(d_stage, stage_got, c->device);      // (waits for the kernels that read it)
    bool ok = false;
    if (e == hipSuccess) e = acquire_band_pack(ic, tmpl->bc_cols, nb, &c->pack, &ok);
    if (e == hipSuccess && !ok) {
        iso_catalog_destroy(c);
        return fail(ISO_ERR_INVALID, "iso_catalog_create_columns: tables not representable on the fast path");
    }
    if (e != hipSuccess) {
        std::string msg = std::string("iso_catalog_create_columns: ") + hipGetErrorString(e);
        iso_catalog_destroy(c);
        return fail(e == hipErrorOutOfMemory ? ISO_ERR_NOMEM : ISO_ERR_HIP, msg);
    }
    c->fast = c->pack->fast;
    c->fast.m = c->d_models;
    c->fast.shared_priors = 1;          // every block is the template's but for observations and the distance prior
    if (const char* env = std::getenv("ISOCHRONES_AMD_SHARED_PRIORS")) c->fast.shared_priors = std::atoi(env) != 0;   // A/B switch
    c->std_priors = default_prior_families(*tmpl) ? 1 : 0;
    c->packed = true;
    *out = c;
    return ISO_OK;
}

void iso_catalog_destroy(iso_catalog* c)
{
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->d_models) {
        if (c->models_bytes) pool_free(c->d_models, c->models_bytes, c->device);     // (iso_catalog_create_columns' block)
        else (void)hipFree(c->d_models);
    }
    if (c->d_bc_hot) (void)hipFree(c->d_bc_hot);
    if (c->d_bcq) (void)hipFree(c->d_bcq);
    if (c->d_axes_blob) (void)hipFree(c->d_axes_blob);
    delete c;
}

int iso_catalog_lnpost(iso_catalog* c, const int32_t* star_id, const double* pars, int64_t stride_n,
                       int64_t stride_p, int64_t n, double* lnpost_out, void* stream)
{
    if (!c || ((!star_id || !pars || !lnpost_out) && n > 0)) return fail(ISO_ERR_INVALID, "iso_catalog_lnpost: NULL argument");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_catalog_lnpost: n < 0");
    if (n == 0) return ISO_OK;
    DeviceGuard guard(c->device);
    FastArgs F = c->fast;
    F.star_id = star_id;
    F.pars = pars;
    F.stride_n = stride_n;
    F.stride_p = stride_p;
    F.n = n;
    F.lnpost = lnpost_out;
    if (!launch_lnpost_fast(c->ic->kind, c->n_stars, c->n_bands, true, F, as_stream(stream)))
        return fail(ISO_ERR_INVALID, "iso_catalog_lnpost: no kernel specialisation");
    HIP_TRY(hipGetLastError());
    return ISO_OK;
}

int iso_catalog_start_points(iso_catalog* c, int nwalkers, int oversample, int max_tries, uint64_t seed, double* best,
                             double* best_lnp, int32_t* failed, void* stream)
{
    if (!c || !best || !best_lnp || !failed) return fail(ISO_ERR_INVALID, "iso_catalog_start_points: NULL argument");
    if (nwalkers < 1 || oversample < 1 || max_tries < 1)
        return fail(ISO_ERR_INVALID, "iso_catalog_start_points: nwalkers, oversample and max_tries must be positive");
    if ((int64_t)oversample * nwalkers * max_tries > (int64_t(1) << 24))
        return fail(ISO_ERR_INVALID, "iso_catalog_start_points: more than 2^24 candidates per star");
    DeviceGuard guard(c->device);
    if (!launch_catalog_start(c->ic->kind, c->n_stars, c->n_bands, c->fast, best, best_lnp, failed, c->n_models, nwalkers,
                              oversample, max_tries, seed, as_stream(stream)))
        return fail(ISO_ERR_INVALID, "iso_catalog_start_points: no kernel for this shape (more than 1024 walkers, or an "
                                         "ensemble whose records do not fit a CU's LDS)");
    HIP_TRY(hipGetLastError());
    return ISO_OK;
}

int iso_catalog_patch_failed(iso_catalog* c, int nwalkers, double* pos, double* lnp, const int32_t* failed, void* stream)
{
    if (!c || !pos || !lnp || !failed) return fail(ISO_ERR_INVALID, "iso_catalog_patch_failed: NULL argument");
    if (nwalkers < 1) return fail(ISO_ERR_INVALID, "iso_catalog_patch_failed: nwalkers must be positive");
    DeviceGuard guard(c->device);
    note_kernel("k_catalog_patch_failed");
    hipLaunchKernelGGL(k_catalog_patch_failed, dim3((unsigned)c->n_models), dim3(BLOCK), 0, as_stream(stream), pos, lnp, failed,
                       c->n_models, nwalkers, c->n_stars + 4);
    HIP_TRY(hipGetLastError());
    return ISO_OK;
}

int iso_tree_model_create(iso_ic* ic, const iso_tree_desc* d, iso_tree_model** out)
{
    if (!ic || !d || !out) return fail(ISO_ERR_INVALID, "iso_tree_model_create: NULL argument");
    if (ic->kind != ISO_KIND_ISO) return fail(ISO_ERR_INVALID, "iso_tree_model_create: isochrone parametrisation only");
    if (ic->prior_cols[0] < 0 || ic->prior_cols[1] < 0)
        return fail(ISO_ERR_INVALID, "iso_tree_model_create: the model table has no EEP-prior columns");
    if (d->n_systems < 1 || d->n_systems > ISO_TREE_MAX_SYSTEMS || d->n_leaves < 1 || d->n_leaves > ISO_TREE_MAX_LEAVES ||
        d->n_bands < 0 || d->n_bands > ISO_TREE_MAX_BANDS || d->n_terms < 0 || d->n_terms > ISO_TREE_MAX_TERMS ||
        d->n_spec < 0 || d->n_spec > ISO_TREE_MAX_SPEC || d->n_limits < 0 || d->n_limits > ISO_TREE_MAX_SPEC)
        return fail(ISO_ERR_INVALID, "iso_tree_model_create: counts out of range");
    int total = 0;
    for (int s = 0; s < d->n_systems; ++s) {
        if (d->n_stars[s] < 1) return fail(ISO_ERR_INVALID, "iso_tree_model_create: empty system");
        total += d->n_stars[s];
    }
    if (total != d->n_leaves) return fail(ISO_ERR_INVALID, "iso_tree_model_create: n_leaves != sum(n_stars)");
    for (int l = 0; l < d->n_leaves; ++l)
        if (d->leaf_system[l] < 0 || d->leaf_system[l] >= d->n_systems || d->leaf_slot[l] < 0 ||
            d->leaf_slot[l] >= d->n_stars[d->leaf_system[l]])
            return fail(ISO_ERR_INVALID, "iso_tree_model_create: bad leaf placement");
    for (int b = 0; b < d->n_bands; ++b)
        if (d->bc_cols[b] < 0 || d->bc_cols[b] >= ic->g4.ncol) return fail(ISO_ERR_INVALID, "iso_tree_model_create: band column out of range");
    const uint32_t all = (d->n_leaves >= 32) ? 0xFFFFFFFFu : ((1u << d->n_leaves) - 1u);
    for (int t = 0; t < d->n_terms; ++t) {
        const iso_tree_term& tt = d->terms[t];
        if (tt.band < 0 || tt.band >= d->n_bands || (tt.mask & ~all) || (tt.ref_mask & ~all))
            return fail(ISO_ERR_INVALID, "iso_tree_model_create: bad observation term");
    }
    for (int k = 0; k < d->n_spec; ++k)
        if (d->spec[k].leaf < 0 || d->spec[k].leaf >= d->n_leaves || d->spec[k].prop < 0 || d->spec[k].prop > 2)
            return fail(ISO_ERR_INVALID, "iso_tree_model_create: bad spectroscopy entry");
    for (int k = 0; k < d->n_limits; ++k)
        if (d->limits[k].leaf < 0 || d->limits[k].leaf >= d->n_leaves || d->limits[k].prop < 0 || d->limits[k].prop > 2)
            return fail(ISO_ERR_INVALID, "iso_tree_model_create: bad limit entry");
    const iso_prior* pr[5] = {&d->prior_mass, &d->prior_age, &d->prior_feh, &d->prior_distance, &d->prior_AV};
    for (int j = 0; j < 5; ++j)
        if (!prior_kind_ok(pr[j]->kind)) return fail(ISO_ERR_INVALID, "iso_tree_model_create: unknown prior family");

    DeviceGuard guard(ic->device);
    iso_tree_model* m = new (std::nothrow) iso_tree_model();
    if (!m) return fail(ISO_ERR_NOMEM, "iso_tree_model_create: out of host memory");
    m->device = ic->device;
    m->ic = ic;
    m->d_tree = nullptr;
    m->d_bc_hot = nullptr;
    m->d_bcq = nullptr;
    m->d_axes_blob = nullptr;
    m->fast_ok = false;
    m->mbox = m->d_mbox = nullptr;
    m->mbox_stream = nullptr;
    m->mbox_count = 0;
    m->mbox_state = 0;
    m->n_bands = d->n_bands;
    m->n_leaves = d->n_leaves;
    DevTree* H = new DevTree();
    std::memset(H, 0, sizeof(DevTree));
    H->n_systems = d->n_systems; H->n_leaves = d->n_leaves; H->n_bands = d->n_bands;
    H->n_terms = d->n_terms; H->n_spec = d->n_spec; H->n_limits = d->n_limits;
    int base = 0;
    for (int s = 0; s < d->n_systems; ++s) {
        H->n_stars[s] = d->n_stars[s];
        H->sys_base[s] = base;
        base += d->n_stars[s] + 4;
        H->has_plx[s] = d->has_plx[s]; H->has_av[s] = d->has_av[s];
        H->plx_val[s] = d->plx_val[s]; H->plx_unc[s] = d->plx_unc[s];
        H->av_val[s] = d->av_val[s]; H->av_unc[s] = d->av_unc[s];
        double u2;
        gauss_consts(d->plx_unc[s], H->plx_g0[s], u2, &H->plx_hinv[s]);
        gauss_consts(d->av_unc[s], H->av_g0[s], u2, &H->av_hinv[s]);
    }
    H->n_params = base;
    m->n_params = base;
    for (int l = 0; l < d->n_leaves; ++l) {
        H->leaf_system[l] = d->leaf_system[l];
        H->leaf_slot[l] = d->leaf_slot[l];
    }
    for (int t = 0; t < d->n_terms; ++t) {
        H->terms[t] = d->terms[t];
        double u2;
        gauss_consts(d->terms[t].unc, H->term_g0[t], u2, &H->term_hinv[t]);
    }
    {   // band-major copy of the terms (stable inside a band), constants folded
        int n = 0;
        for (int b = 0; b <= ISO_TREE_MAX_BANDS; ++b) {
            H->bterm_first[b] = n;
            if (b == ISO_TREE_MAX_BANDS) break;
            for (int t = 0; t < d->n_terms; ++t) {
                if (d->terms[t].band != b) continue;
                DevTreeTerm& bt = H->bterms[n++];
                bt.mask = d->terms[t].mask;
                bt.ref_mask = d->terms[t].ref_mask;
                bt.relative = d->terms[t].relative;
                bt.pad_ = 0;
                bt.dmag = d->terms[t].relative ? d->terms[t].mag - d->terms[t].ref_mag : d->terms[t].mag;
                bt.g0 = H->term_g0[t];
                bt.hinv = H->term_hinv[t];
            }
        }
    }
    for (int k = 0; k < d->n_spec; ++k) {
        H->spec[k] = d->spec[k];
        double u2;
        gauss_consts(d->spec[k].b, H->spec_g0[k], u2, &H->spec_hinv[k]);
    }
    for (int k = 0; k < d->n_limits; ++k) H->limits[k] = d->limits[k];
    H->prior_mass = make_dev_prior(d->prior_mass);
    H->prior_age = make_dev_prior(d->prior_age);
    H->prior_feh = make_dev_prior(d->prior_feh);
    H->prior_distance = make_dev_prior(d->prior_distance);
    H->prior_AV = make_dev_prior(d->prior_AV);
    H->std_priors = (d->prior_mass.kind == ISO_PRIOR_CHABRIER && d->prior_age.kind == ISO_PRIOR_FLATLOG && d->prior_feh.kind == ISO_PRIOR_FEH &&
                     d->prior_feh.c != 0.0 && d->prior_distance.kind == ISO_PRIOR_POWERLAW && d->prior_AV.kind == ISO_PRIOR_FLAT) ? 1 : 0;
    if (const char* e = std::getenv("ISOCHRONES_AMD_STD_PRIORS")) H->std_priors = H->std_priors && std::atoi(e) != 0;   // A/B switch
    H->eep_lo = d->eep_lo; H->eep_hi = d->eep_hi;
    for (int j = 0; j < 4; ++j) {
        H->bound_lo[j] = d->bound_lo[j];
        H->bound_hi[j] = d->bound_hi[j];
    }
    hipError_t e = hipMalloc(&m->d_tree, sizeof(DevTree));
    if (e == hipSuccess) e = hipMemcpy(m->d_tree, H, sizeof(DevTree), hipMemcpyHostToDevice);
    delete H;
    m->g4 = ic->g4;
    if (e == hipSuccess && d->n_bands > 0) {
        e = pack_bands(ic, d->bc_cols, d->n_bands, &m->d_bc_hot);
        m->g4.tab = m->d_bc_hot;
        m->g4.ncol = d->n_bands;
    }
    if (e == hipSuccess && path_mode() == PATH_AUTO && ic->d_hotq && d->n_bands >= 1 && d->n_bands <= ISO_TREE_MAX_BANDS &&
        third_axis_ok(ic)) {
        bool ok = false;
        e = build_fast(ic, d->n_bands, m->d_bc_hot, &m->d_axes_blob, &m->d_bcq, m->fast, &ok);
        m->fast_ok = ok && m->d_bcq != nullptr;
    }
    if (e != hipSuccess) {
        std::string msg = std::string("iso_tree_model_create: ") + hipGetErrorString(e);
        iso_tree_model_destroy(m);
        return fail(e == hipErrorOutOfMemory ? ISO_ERR_NOMEM : ISO_ERR_HIP, msg);
    }
    *out = m;
    return ISO_OK;
}

namespace {
inline unsigned long long tmb_load(const volatile unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }

bool tree_mailbox_launch(iso_tree_model* m)
{
    double idle_us = 1000.0;
    if (const char* e = std::getenv("ISOCHRONES_AMD_MAILBOX_IDLE_US")) idle_us = std::max(10.0, std::atof(e));
    const unsigned long long idle = (unsigned long long)(idle_us * 1e-6 * 1.0e8);       // wall_clock64(): 100 MHz
    const unsigned long long life = (unsigned long long)(30.0 * 1.0e8);
    __atomic_store_n(&m->mbox->ctl[1], 0ull, __ATOMIC_RELAXED);
    __atomic_store_n(&m->mbox->ctl[0], 1ull, __ATOMIC_RELEASE);
    FastArgs F = m->fast;
    F.pars = nullptr;
    F.n = 0;
    F.lnpost = F.lnprior = F.lnlike = nullptr;
    if (!launch_tree_mailbox(m->n_bands, m->n_leaves, F, m->d_tree, m->d_mbox, idle, life, m->mbox_stream) ||
        hipGetLastError() != hipSuccess) {
        __atomic_store_n(&m->mbox->ctl[0], 2ull, __ATOMIC_RELEASE);
        return false;
    }
    return true;
}

bool tree_mailbox_ready(iso_tree_model* m)
{
    if (m->mbox_state < 0) return false;
    if (m->mbox_state > 0) return true;
    m->mbox_state = -1;
    if (!m->fast_ok || m->n_params > 24) return false;
    if (hipHostMalloc(reinterpret_cast<void**>(&m->mbox), sizeof(IsoTreeBox), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        m->mbox = nullptr;
        return false;
    }
    std::memset(m->mbox, 0, sizeof(IsoTreeBox));
    m->mbox->ctl[0] = 2;                             // no wave yet
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&m->d_mbox), m->mbox, 0) != hipSuccess ||
        hipStreamCreateWithFlags(&m->mbox_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(m->mbox);
        m->mbox = nullptr;
        return false;
    }
    m->mbox_state = 1;
    return true;
}

// one row through the resident wave; ISO_OK, or 1 = not served (the caller launches instead)
int tree_mailbox_call(iso_tree_model* m, const double* pars, double* lnpost_out, double* lnprior_out, double* lnlike_out)
{
    IsoTreeBox* mb = m->mbox;
    const int np_ = m->n_params;
    const bool parts = lnprior_out || lnlike_out;
    unsigned long long words[24];
    for (int q = 0; q < np_; ++q) {
        std::memcpy(&words[q], pars + q, 8);
        __atomic_store_n(&mb->req[1 + q], words[q], __ATOMIC_RELAXED);
    }
    const unsigned long long seq = ((unsigned long long)mailbox_checksum(words, np_) << 32) | ((++m->mbox_count & 0xFFFFull) << 16) |
                                   ((unsigned long long)parts << 8);
    __atomic_store_n(&mb->req[0], seq, __ATOMIC_RELEASE);        // the sequence word last
    if (tmb_load(&mb->ctl[0]) != 1 && !tree_mailbox_launch(m)) {
        m->mbox_state = -1;                                      // no instantiation for this shape: the launch path from now on
        return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 1; tmb_load(&mb->done[0]) != seq; ++spins) {
        if ((spins & 255) == 0) {
            if (tmb_load(&mb->ctl[0]) == 2 && tmb_load(&mb->done[0]) != seq) {
                if (!tree_mailbox_launch(m)) return 1;
            } else if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                __atomic_store_n(&mb->ctl[1], 1ull, __ATOMIC_RELEASE);
                (void)hipStreamSynchronize(m->mbox_stream);
                m->mbox_state = -1;
                return 1;
            }
        }
    }
    double r[3];
    for (int k = 0; k < 3; ++k) {
        const unsigned long long w = __atomic_load_n(&mb->done[1 + k], __ATOMIC_RELAXED);
        std::memcpy(&r[k], &w, 8);
    }
    if (lnpost_out) *lnpost_out = r[0];
    if (lnprior_out) *lnprior_out = r[1];
    if (lnlike_out) *lnlike_out = r[2];
#ifdef ISO_MAILBOX_CLOCK        // (variant builds: the wave's own time from seeing a request to its results, 100 MHz ticks)
    {
        static unsigned long long calls = 0, ticks = 0, ph[5] = {0, 0, 0, 0, 0};
        ticks += __atomic_load_n(&mb->done[4], __ATOMIC_RELAXED);
#ifdef ISO_PHASE_CLOCK
        for (int k = 0; k < 3; ++k) ph[k] += __atomic_load_n(&mb->done[5 + k], __ATOMIC_RELAXED);
        for (int k = 0; k < 2; ++k) ph[3 + k] += __atomic_load_n(&mb->ctl[4 + k], __ATOMIC_RELAXED);
#endif
        if (++calls % 2000 == 0) {
            std::fprintf(stderr, "tree mailbox: %.2f us on the device per call (%llu calls)\n", ticks * 0.01 / (double)calls, calls);
#ifdef ISO_PHASE_CLOCK
            std::fprintf(stderr, "tree mailbox: shader clocks from the request: first model cell %.0f, leaves done %.0f, priors %.0f, likelihood %.0f, results written %.0f\n",
                         ph[0] / (double)calls, ph[1] / (double)calls, ph[2] / (double)calls, ph[3] / (double)calls, ph[4] / (double)calls);
            for (auto& v : ph) v = 0;
#endif
            calls = ticks = 0;
        }
    }
#endif
    return ISO_OK;
}

void tree_mailbox_stop(iso_tree_model* m, bool release)
{
    if (!m->mbox) return;
    if (tmb_load(&m->mbox->ctl[0]) == 1) {
        __atomic_store_n(&m->mbox->ctl[1], 1ull, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(m->mbox_stream);
    }
    if (release) {
        (void)hipStreamSynchronize(m->mbox_stream);
        (void)hipStreamDestroy(m->mbox_stream);
        (void)hipHostFree(m->mbox);
        m->mbox = m->d_mbox = nullptr;
        m->mbox_stream = nullptr;
        m->mbox_state = 0;
    }
}
}  // namespace

void iso_tree_model_destroy(iso_tree_model* m)
{
    if (!m) return;
    DeviceGuard guard(m->device);
    tree_mailbox_stop(m, true);        // the resident wave reads the tables below
    if (m->d_tree) (void)hipFree(m->d_tree);
    if (m->d_bc_hot) (void)hipFree(m->d_bc_hot);
    if (m->d_bcq) (void)hipFree(m->d_bcq);
    if (m->d_axes_blob) (void)hipFree(m->d_axes_blob);
    delete m;
}

int iso_tree_lnpost(iso_tree_model* m, const double* pars, int64_t stride_n, int64_t stride_p, int64_t n,
                    double* lnpost_out, double* lnprior_out, double* lnlike_out, void* stream)
{
    if (!m || (!pars && n > 0)) return fail(ISO_ERR_INVALID, "iso_tree_lnpost: NULL argument");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_tree_lnpost: n < 0");
    if (!lnpost_out && !lnprior_out && !lnlike_out) return fail(ISO_ERR_INVALID, "iso_tree_lnpost: no output requested");
    if (n == 0) return ISO_OK;
    if (m->fast_ok) {
        FastArgs F = m->fast;
        F.pars = pars;
        F.stride_n = stride_n;
        F.stride_p = stride_p;
        F.n = n;
        F.lnpost = lnpost_out;
        F.lnprior = lnprior_out;
        F.lnlike = lnlike_out;
        DeviceGuard guard(m->device);
        if (launch_tree_fast(m->n_bands, m->n_leaves, F, m->d_tree, as_stream(stream))) {
            HIP_TRY(hipGetLastError());
            return ISO_OK;
        }
    }
    TreeArgs A;
    A.g3 = m->ic->g3;
    A.g4 = m->g4;
    A.T = m->d_tree;
    A.pars = pars;
    A.stride_n = stride_n;
    A.stride_p = stride_p;
    A.n = n;
    A.lnpost = lnpost_out;
    A.lnprior = lnprior_out;
    A.lnlike = lnlike_out;
    DeviceGuard guard(m->device);
    // per-leaf values in LDS behind the staged axes, [slot][lane]: as many lanes per workgroup as fit the CU's 160 KB
    const size_t axes = (size_t)m->ic->lds_doubles, per_lane = (size_t)m->n_leaves * (6 + (size_t)m->n_bands);
    int lanes = BLOCK;
    while (lanes > 64 && (axes + per_lane * lanes) * sizeof(double) > 160 * 1024) lanes >>= 1;
    const size_t bytes = (axes + per_lane * lanes) * sizeof(double);
    if (bytes > 160 * 1024) return fail(ISO_ERR_INVALID, "iso_tree_lnpost: the tree's per-star values do not fit a CU's LDS");
    if (bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_lnpost_tree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    note_kernel("k_lnpost_tree");
    hipLaunchKernelGGL(k_lnpost_tree, dim3((unsigned)((n + lanes - 1) / lanes)), dim3(lanes),      // one sample per lane
                       bytes, as_stream(stream), A, (int)axes);
    HIP_TRY(hipGetLastError());
    return ISO_OK;
}

int iso_tree_lnpost_host(iso_tree_model* m, const double* pars, int64_t n, double* lnpost_out, double* lnprior_out,
                         double* lnlike_out)
{
    if (!m || (!pars && n > 0)) return fail(ISO_ERR_INVALID, "iso_tree_lnpost_host: NULL argument");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_tree_lnpost_host: n < 0");
    if (!lnpost_out && !lnprior_out && !lnlike_out) return fail(ISO_ERR_INVALID, "iso_tree_lnpost_host: no output requested");
    if (n == 0) return ISO_OK;
    DeviceGuard guard(m->device);
    iso_ctx* ctx = m->ic->ctx;
    std::lock_guard<std::mutex> lock(ctx->stage_mu);
    // a sampler's per-point callback: the model's resident mailbox wave - no launch (fast/tree_mailbox.h)
    if (n == 1 && mailbox_enabled() && tree_mailbox_ready(m)) {
        const int rc1 = tree_mailbox_call(m, pars, lnpost_out, lnprior_out, lnlike_out);
        if (rc1 <= 0) return rc1;
    }
    double *h = nullptr, *d = nullptr;
    int rc = ctx_stage(ctx, &h, &d);
    if (rc != ISO_OK) return rc;
    const int np_ = m->n_params;
    const int64_t cap = ISO_CTX_STAGE_DOUBLES / (np_ + 3);
    for (int64_t done = 0; done < n; done += cap) {
        const int64_t c = std::min<int64_t>(cap, n - done);
        std::memcpy(h, pars + done * np_, sizeof(double) * c * np_);
        double *dpost = d + c * np_, *dprior = dpost + c, *dlike = dprior + c;
        rc = iso_tree_lnpost(m, d, np_, 1, c, lnpost_out ? dpost : nullptr, lnprior_out ? dprior : nullptr,
                             lnlike_out ? dlike : nullptr, nullptr);
        if (rc != ISO_OK) return rc;
        rc = ctx_wait(ctx, h, d);
        if (rc != ISO_OK) return rc;
        const double* hp = h + c * np_;
        if (lnpost_out) std::memcpy(lnpost_out + done, hp, sizeof(double) * c);
        if (lnprior_out) std::memcpy(lnprior_out + done, hp + c, sizeof(double) * c);
        if (lnlike_out) std::memcpy(lnlike_out + done, hp + 2 * c, sizeof(double) * c);
    }
    return ISO_OK;
}

namespace {
int sampler_common(iso_sampler* sp, int device, int kind, int n_stars, int n_bands, int64_t n_ens, const FastArgs& F,
                   int multi, int nwalkers, double a, uint64_t seed)
{
    sp->device = device;
    sp->kind = kind;
    sp->n_stars = n_stars;
    sp->n_bands = n_bands;
    sp->n_params = n_stars + 4;
    sp->n_ensembles = n_ens;
    sp->W = nwalkers;
    sp->a = a;
    sp->seed = seed;
    sp->step = 0;
    sp->multi = multi;
    sp->std_priors = 0;
    sp->chain_layout = ISO_CHAIN_ROW_MAJOR;
    sp->fast = F;
    sp->form = 0;
    sp->d_tree = nullptr;
    sp->n_leaves = 0;
    sp->fast2 = F;
    sp->age = IsoTrackAge{0.0, 0.0, 0.0};
    return ISO_OK;
}

bool walkers_ok(int nwalkers, double a) { return nwalkers >= 2 && !(nwalkers & 1) && a > 1.0; }

// the moves' side of a run of the any-model persistent sampler
AnyStretchArgs any_args(const iso_sampler* sp, double* pos, double* lnp, int32_t* accepted, int nsteps, double* chain,
                        double* chain_lnp)
{
    const int64_t rows = sp->n_ensembles * sp->W;
    AnyStretchArgs S;
    S.pos = pos;
    S.lnp = lnp;
    S.accepted = accepted;
    S.W = sp->W;
    S.NP = sp->n_params;
    S.lanes = BLOCK;
    S.own_off = 0;
    S.n_ens = sp->n_ensembles;
    S.a = sp->a;
    S.seed = sp->seed;
    S.step = sp->step;
    S.nsteps = nsteps;
    S.chain_rs = sp->chain_layout == ISO_CHAIN_PARAM_MAJOR ? 1 : sp->n_params;
    S.chain_ps = sp->chain_layout == ISO_CHAIN_PARAM_MAJOR ? rows : 1;
    S.chain_pos = chain;
    S.chain_lnp = chain_lnp;
    return S;
}

// launch (query == null) or ask whether a kernel exists whose LDS fits (query != null)
bool launch_any_form(const iso_sampler* sp, const AnyStretchArgs& S, int* query, hipStream_t s)
{
    switch (sp->form) {
    case ISO_SAMPLER_TREE: return launch_stretch_tree(sp->n_bands, sp->n_leaves, sp->fast, sp->d_tree, S, query, s);
    case ISO_SAMPLER_ISOTRACK: return launch_stretch_isotrack(sp->n_bands, sp->fast, sp->fast2, sp->age, S, query, s);
    case ISO_SAMPLER_WIDE: return launch_stretch_wide(sp->kind, sp->n_stars, sp->fast, S, query, s);
    }
    return false;
}

// a sampler of one of the any-model forms is only handed out when a kernel for its shape exists and an ensemble fits a CU's LDS
int any_form_ready(iso_sampler* sp, const char* who)
{
    int lanes = 0;
    const AnyStretchArgs S = any_args(sp, nullptr, nullptr, nullptr, 0, nullptr, nullptr);
    if (launch_any_form(sp, S, &lanes, nullptr)) return ISO_OK;
    delete sp;
    return fail(ISO_ERR_INVALID, std::string(who) + ": no device-resident sampler for this shape (band count without a kernel, or "
                                                    "an ensemble whose positions do not fit a CU's 160 KB of LDS)");
}
}  // namespace

int iso_sampler_create_model(iso_model* m, int nwalkers, double a, uint64_t seed, iso_sampler** out)
{
    if (!m || !out) return fail(ISO_ERR_INVALID, "iso_sampler_create_model: NULL argument");
    if (nwalkers < 2 || (nwalkers & 1) || !(a > 1.0)) return fail(ISO_ERR_INVALID, "iso_sampler_create_model: need an even walker count and a > 1");
    if (!m->fast_ok || !m->fast.hotq || (!m->fast.bcq && m->desc.n_bands > 0))
        return fail(ISO_ERR_INVALID, "iso_sampler_create_model: the model is not on the corner-packed fast path "
                                     "(ISOCHRONES_AMD_PATH=auto, tables that could be packed)");
    iso_sampler* sp = new (std::nothrow) iso_sampler();
    if (!sp) return fail(ISO_ERR_NOMEM, "iso_sampler_create_model: out of host memory");
    sampler_common(sp, m->device, m->ic->kind, m->desc.n_stars, m->desc.n_bands, 1, m->fast, 0, nwalkers, a, seed);
    if (m->desc.n_bands > FAST_NB_MAX) {
        // 13-32 bands: the band-tiled evaluation inside the any-model persistent sampler (k_stretch_wide)
        if (m->fast.astq) {
            delete sp;
            return fail(ISO_ERR_INVALID, "iso_sampler_create_model: no device-resident sampler for asteroseismic terms with more than 12 bands");
        }
        sp->form = ISO_SAMPLER_WIDE;
        const int rc = any_form_ready(sp, "iso_sampler_create_model");
        if (rc != ISO_OK) return rc;
        *out = sp;
        return ISO_OK;
    }
    sp->std_priors = default_prior_families(m->desc);
    if (const char* e = std::getenv("ISOCHRONES_AMD_STD_PRIORS")) sp->std_priors = sp->std_priors && std::atoi(e) != 0;   // A/B switch
    if (m->fast.astq) {
        // asteroseismic terms: the persistent kernel only (its step-wise twin was pruned - a single model's ensemble fits a
        // workgroup's LDS up to ~1 000 walkers), priors read at run time
        sp->std_priors = 0;
        if (stretch_persist_lds(sp->n_bands, sp->fast.axes_len, sp->W, sp->n_params, nullptr) > PERSIST_LDS_MAX) {
            delete sp;
            return fail(ISO_ERR_INVALID, "iso_sampler_create_model: an ensemble of this size does not fit the persistent kernel's LDS, and "
                                         "models with asteroseismic terms have no step-wise sampler kernel");
        }
    }
    *out = sp;
    return ISO_OK;
}

int iso_sampler_create_model_ensembles(iso_model* m, int64_t n_ensembles, int nwalkers, double a, uint64_t seed,
                                       iso_sampler** out)
{
    if (n_ensembles < 1 || n_ensembles > (int64_t(1) << 24))
        return fail(ISO_ERR_INVALID, "iso_sampler_create_model_ensembles: n_ensembles out of range");
    const int rc = iso_sampler_create_model(m, nwalkers, a, seed, out);
    if (rc == ISO_OK) (*out)->n_ensembles = n_ensembles;      // every row evaluates the one model (multi = 0)
    return rc;
}

int iso_sampler_create_catalog(iso_catalog* c, int nwalkers, double a, uint64_t seed, iso_sampler** out)
{
    if (!c || !out) return fail(ISO_ERR_INVALID, "iso_sampler_create_catalog: NULL argument");
    if (nwalkers < 2 || (nwalkers & 1) || !(a > 1.0)) return fail(ISO_ERR_INVALID, "iso_sampler_create_catalog: need an even walker count and a > 1");
    iso_sampler* sp = new (std::nothrow) iso_sampler();
    if (!sp) return fail(ISO_ERR_NOMEM, "iso_sampler_create_catalog: out of host memory");
    sampler_common(sp, c->device, c->ic->kind, c->n_stars, c->n_bands, c->n_models, c->fast, 1, nwalkers, a, seed);
    // stars that share their priors, and those the reference's defaults: the resident catalog kernel reads them from the
    // first block through scalar loads with the families as compile-time constants (fast/launch.h)
    sp->std_priors = c->std_priors && c->fast.shared_priors;
    if (const char* e = std::getenv("ISOCHRONES_AMD_STD_PRIORS")) sp->std_priors = sp->std_priors && std::atoi(e) != 0;   // A/B switch
    *out = sp;
    return ISO_OK;
}

int iso_sampler_create_tree(iso_tree_model* m, int64_t n_ensembles, int nwalkers, double a, uint64_t seed, iso_sampler** out)
{
    if (!m || !out) return fail(ISO_ERR_INVALID, "iso_sampler_create_tree: NULL argument");
    if (!walkers_ok(nwalkers, a)) return fail(ISO_ERR_INVALID, "iso_sampler_create_tree: need an even walker count and a > 1");
    if (n_ensembles < 1 || n_ensembles > (int64_t(1) << 24))
        return fail(ISO_ERR_INVALID, "iso_sampler_create_tree: n_ensembles out of range");
    if (!m->fast_ok)
        return fail(ISO_ERR_INVALID, "iso_sampler_create_tree: the tree model is not on the corner-packed fast path (1-16 bands, "
                                     "ISOCHRONES_AMD_PATH=auto)");
    iso_sampler* sp = new (std::nothrow) iso_sampler();
    if (!sp) return fail(ISO_ERR_NOMEM, "iso_sampler_create_tree: out of host memory");
    sampler_common(sp, m->device, ISO_KIND_ISO, 0, m->n_bands, n_ensembles, m->fast, 0, nwalkers, a, seed);
    sp->n_params = m->n_params;
    sp->form = ISO_SAMPLER_TREE;
    sp->d_tree = m->d_tree;
    sp->n_leaves = m->n_leaves;
    const int rc = any_form_ready(sp, "iso_sampler_create_tree");
    if (rc != ISO_OK) return rc;
    *out = sp;
    return ISO_OK;
}

int iso_sampler_create_isotrack(iso_model* iso_m, iso_model* track_m, double age_lo, double age_hi, double age_lnorm,
                                int64_t n_ensembles, int nwalkers, double a, uint64_t seed, iso_sampler** out)
{
    if (!iso_m || !track_m || !out) return fail(ISO_ERR_INVALID, "iso_sampler_create_isotrack: NULL argument");
    if (!walkers_ok(nwalkers, a)) return fail(ISO_ERR_INVALID, "iso_sampler_create_isotrack: need an even walker count and a > 1");
    if (n_ensembles < 1 || n_ensembles > (int64_t(1) << 24))
        return fail(ISO_ERR_INVALID, "iso_sampler_create_isotrack: n_ensembles out of range");
    if (iso_m->ic->kind != ISO_KIND_ISO || track_m->ic->kind != ISO_KIND_TRACK || iso_m->desc.n_stars != 1 || track_m->desc.n_stars != 1)
        return fail(ISO_ERR_INVALID, "iso_sampler_create_isotrack: need a single-star model on the isochrone grid and one on the track grid");
    if (iso_m->device != track_m->device) return fail(ISO_ERR_INVALID, "iso_sampler_create_isotrack: the two models live on different devices");
    if (iso_m->desc.n_bands != track_m->desc.n_bands || iso_m->desc.n_bands > FAST_NB_MAX)
        return fail(ISO_ERR_INVALID, "iso_sampler_create_isotrack: both models must carry the same (at most 12) bands");
    for (iso_model* m : {iso_m, track_m})
        if (!m->fast_ok || !m->fast.hotq || (!m->fast.bcq && m->desc.n_bands > 0) || m->fast.astq)
            return fail(ISO_ERR_INVALID, "iso_sampler_create_isotrack: both models must be on the corner-packed fast path, without "
                                         "asteroseismic terms");
    iso_sampler* sp = new (std::nothrow) iso_sampler();
    if (!sp) return fail(ISO_ERR_NOMEM, "iso_sampler_create_isotrack: out of host memory");
    sampler_common(sp, iso_m->device, ISO_KIND_ISO, 1, iso_m->desc.n_bands, n_ensembles, iso_m->fast, 0, nwalkers, a, seed);
    sp->n_params = 6;
    sp->form = ISO_SAMPLER_ISOTRACK;
    sp->fast2 = track_m->fast;
    sp->age = IsoTrackAge{age_lo, age_hi, age_lnorm};
    const int rc = any_form_ready(sp, "iso_sampler_create_isotrack");
    if (rc != ISO_OK) return rc;
    *out = sp;
    return ISO_OK;
}

void iso_sampler_destroy(iso_sampler* s) { delete s; }

int iso_sampler_set_chain_layout(iso_sampler* s, int layout)
{
    if (!s) return fail(ISO_ERR_INVALID, "iso_sampler_set_chain_layout: NULL argument");
    if (layout != ISO_CHAIN_ROW_MAJOR && layout != ISO_CHAIN_PARAM_MAJOR)
        return fail(ISO_ERR_INVALID, "iso_sampler_set_chain_layout: unknown layout");
    s->chain_layout = layout;
    return ISO_OK;
}


namespace {
// What one iso_sampler_run launches for the BasicStarModel / catalog kernels - decided in ONE place, readable from outside
// (iso_debug_sampler_plan) and asserted by tests.  The rules, in the order they are applied (S = the launch's StretchArgs;
// "resident" = every workgroup of the launch on the chip at once, by the runtime's occupancy figure for the very
// instantiation that would be launched):
//
//   | question                              | rule                                                                          | evidence                         |
//   |---------------------------------------|-------------------------------------------------------------------------------|----------------------------------|
//   | persistent or one launch per half-step| persistent whenever an ensemble's LDS fits the CU's 160 KB (mode `auto`);      | r03/sampler_mode_sweep.txt,      |
//   |                                       | `stepwise` / `persistent` / `persistent-dense` force a form (tests, A/B)      | r06 (the 64-KB limit: 55 -> 31 us)|
//   | uncapped or register-capped (dense)   | the uncapped form if it keeps the launch resident (2 workgroups per CU), else | r03, r04                         |
//   |                                       | the dense one (3-4 per CU) in rounds; dense needs priors the stars share      |                                  |
//   | threads per workgroup (dense)         | 192 for ensembles of 129-192 moves per half-step (5 workgroups per CU), else  | r05/ab_three_wave_workgroups     |
//   |                                       | 256                                                                           |                                  |
//   | priors compiled in (dense, defaults)  | yes, unless that splits into rounds a launch the run-time form keeps resident | r05/ab_dense_stdp_final          |
//   | ensembles per workgroup               | halved while >= 64 moves per half-step remain and the workgroups still fit    | r04, r06/catalog_groups          |
//   |                                       | the CUs; kept only if the launch stays resident with the smaller groups       |                                  |
//   | one star per lane / row (single model)| binary: <= 128 moves (<= 64 beyond 4 bands); triple: <= 64 moves; default     | r04/pair_kernel_ab,              |
//   |                                       | prior families only (fast/launch.h picks the kernel from S.pair / triple_moves)| r06/triple_sweep                 |
struct SamplerPlan {
    bool persistent;
    int per_cu;          // workgroups per CU of the instantiation that will be launched (occupancy query)
    int group;           // ensembles a workgroup's LDS is laid out for
};

int plan_sampler_run(iso_sampler* sp, int nsteps, hipStream_t s, StretchArgs& S, SamplerPlan& P)
{
    // ISOCHRONES_AMD_SAMPLER = auto | persistent | stepwise.  Both forms produce bit-identical chains.  The persistent
    // kernel (workgroups own their ensembles for all iterations of the call, positions in LDS) is the faster one
    // at every catalog size measured (tools/sampler_mode_sweep.py, profiles/r03: 25-35 % over 2 x 10^3 ... 4 x 10^5 stars,
    // 1-6 bands, 1-2 stars per system): beyond the chip's capacity its workgroups run in rounds, and the few
    // thousand ensembles resident at a time re-read table lines that are still in L2 / Infinity Cache, where a
    // half-step launch over the whole catalog streams everything from HBM.  `auto` therefore runs it whenever an
    // ensemble fits a workgroup's LDS.
    const char* env = getenv("ISOCHRONES_AMD_SAMPLER");
    std::string mode = env ? env : "auto";
    const bool force_dense = mode == "persistent-dense";          // test hook: the register-capped instantiation
    if (force_dense) mode = "persistent";
    if (mode != "auto" && mode != "persistent" && mode != "stepwise")
        return fail(ISO_ERR_INVALID, "ISOCHRONES_AMD_SAMPLER must be auto, persistent or stepwise");
    if (sp->fast.astq && mode == "stepwise") mode = "persistent";      // asteroseismic models have the persistent form only
    int group = 1;
    const size_t lds_bytes = stretch_persist_lds(sp->n_bands, sp->fast.axes_len, sp->W, sp->n_params, &group);
    // (up to the CU's 160 KB, asked for explicitly beyond 64 - fast/launch.h.  Until round 6 the limit here was 64 KB, and a
    // 256-walker triple with nine bands - 67 KB with the LDS laid out for the two ensembles a workgroup of 128-move half-steps
    // could hold - ran one launch per half-step: 55 us per step, the "triple latency" of round 5's review)
    const bool fits = lds_bytes <= PERSIST_LDS_MAX;
    if (mode == "persistent" && !fits)
        return fail(ISO_ERR_INVALID, "iso_sampler_run: ensemble too large for the persistent kernel's LDS");
    // resident = workgroups the chip holds at once (occupancy of this kernel instantiation as the runtime reports it)
    int cus = 0, per_cu = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, sp->device));
    const int64_t blocks = (sp->n_ensembles + group - 1) / group;
    // two instantiations: uncapped registers (2 workgroups per CU, fastest per iteration) and "dense" (4 per CU for single
    // stars with <= 6 bands, else 3): the first that keeps every workgroup resident, else the dense one in rounds
    int dense = 0;
    if (fits && mode != "stepwise") {
        S.nsteps = 1;
        // (the register-capped instantiation takes the stars' shared priors from the first block: not for a catalog whose
        // stars carry priors of their own)
        const bool dense_ok = !sp->multi || sp->fast.shared_priors;
        for (int dn = (force_dense && dense_ok) ? 1 : 0; dn < (dense_ok ? 2 : 1); ++dn) {
            S.dense = dn;
            S.occupancy_query = &per_cu;
            per_cu = 0;
            if (!launch_stretch(sp->kind, sp->n_stars, sp->n_bands, sp->fast, S, s)) per_cu = 0;
            S.occupancy_query = nullptr;
            dense = dn;
            if (blocks <= (int64_t)cus * per_cu) break;
        }
    }
    S.dense = dense;
    // The register-capped form with ensembles of 129 ... 192 moves per half-step (258 ... 384 walkers; the reference's default is
    // 300): workgroups of THREE waves - the moves are packed into full waves, a fourth would only sit at the barrier - so
    // that five workgroups share a CU instead of four.  ISOCHRONES_AMD_DENSE_THREADS=256 keeps four waves (A/B, tests).
    S.threads = 0;
    if (dense && sp->W / 2 > 128 && sp->W / 2 <= 192) {
        int want = 192;
        if (const char* e = std::getenv("ISOCHRONES_AMD_DENSE_THREADS")) want = std::atoi(e);
        if (want == 192) {
            S.threads = 192;
            int per_cu_t = 0;
            S.occupancy_query = &per_cu_t;
            if (!launch_stretch(sp->kind, sp->n_stars, sp->n_bands, sp->fast, S, s)) per_cu_t = 0;
            S.occupancy_query = nullptr;
            if (per_cu_t > 0) per_cu = per_cu_t;
            else S.threads = 0;
        }
    }
    // The register-capped form exists twice: prior families read at run time (four waves per SIMD for single stars with up to six
    // bands: more workgroups per CU), or - the launch's stars share the reference's default priors - compiled in, at the registers
    // of three waves per SIMD (6-9 % fewer cycles per move, a fifth fewer workgroups per CU).  The second one unless it would
    // split into rounds a launch that the first keeps on the chip in ONE (1 250 stars of the reference shape: 12.8 against
    // 14 ms; 10^4 stars: 82 -> 77 ms the other way; profiles/r05/ab_dense_stdp3.jsonl).  ISOCHRONES_AMD_DENSE_STDP=0 / 1 pins it.
    S.dense_stdp = 0;
    if (dense && sp->std_priors) {
        int per_cu_std = 0;
        S.dense_stdp = 1;
        S.occupancy_query = &per_cu_std;
        if (!launch_stretch(sp->kind, sp->n_stars, sp->n_bands, sp->fast, S, s)) per_cu_std = 0;
        S.occupancy_query = nullptr;
        const bool std_one_round = per_cu_std > 0 && blocks <= (int64_t)cus * per_cu_std;
        const bool rt_one_round = per_cu > 0 && blocks <= (int64_t)cus * per_cu;
        bool take = per_cu_std > 0 && (std_one_round || !rt_one_round);
        if (const char* e = std::getenv("ISOCHRONES_AMD_DENSE_STDP")) take = per_cu_std > 0 && std::atoi(e) != 0;
        if (take) per_cu = per_cu_std;
        else S.dense_stdp = 0;
    }
    // A catalog that leaves CUs idle at `group` ensembles per workgroup is spread over more of them: fewer ensembles per
    // workgroup (a power of two, at least 64 moves per half-step so that every wave keeps a full gather round), as many
    // workgroups as there are CUs at most.  ISOCHRONES_AMD_PERSIST_GROUP=n pins the number (sweeps, A/B).
    if (fits && per_cu > 0 && mode != "stepwise" && sp->n_ensembles > 1) {
        const int h = sp->W / 2;
        int gmin = 1;
        while (gmin * h < 64 && gmin < group) gmin <<= 1;
        int g = group;
        while ((g >> 1) >= gmin && (sp->n_ensembles + (g >> 1) - 1) / (g >> 1) <= (int64_t)cus) g >>= 1;     // (never below gmin: a
                                                                                         // group size need not be a power of two)
        if (const char* e = std::getenv("ISOCHRONES_AMD_PERSIST_GROUP")) {
            const int want = std::atoi(e);
            if (want >= 1) g = std::min(want, group);
        }
        if (g < group) {
            S.group = g;
            // fewer ensembles per workgroup = more workgroups, and possibly ANOTHER instantiation (a single binary's
            // one-star-per-lane form is chosen by the moves a workgroup holds): ask again with the group in place, so that the
            // occupancy is the launched kernel's.  If the launch no longer fits the chip in one round, keep the full groups.
            int per_cu_g = 0;
            S.occupancy_query = &per_cu_g;
            if (!launch_stretch(sp->kind, sp->n_stars, sp->n_bands, sp->fast, S, s)) per_cu_g = 0;
            S.occupancy_query = nullptr;
            const int64_t blocks_g = (sp->n_ensembles + g - 1) / g;
            if (per_cu_g <= 0 || blocks_g > (int64_t)cus * per_cu_g) S.group = 0;
        }
    }
    // (round 2 kept the step-wise form for catalogs of 1-1.4 rounds, where a nearly empty second round cost more than it;
    // with four workgroups per CU the persistent form is ahead there too - profiles/r03/sampler_mode_sweep.txt)
    P.persistent = nsteps > 0 && fits && per_cu > 0 && mode != "stepwise";
    P.per_cu = per_cu;
    P.group = group;
    return ISO_OK;
}
}  // namespace

int iso_sampler_run(iso_sampler* sp, double* pos, double* lnp, int nsteps, double* chain, double* chain_lnp,
                    int32_t* accepted, void* stream)
{
    if (!sp || !pos || !lnp) return fail(ISO_ERR_INVALID, "iso_sampler_run: NULL argument");
    if (nsteps < 0) return fail(ISO_ERR_INVALID, "iso_sampler_run: nsteps < 0");
    DeviceGuard guard(sp->device);
    hipStream_t s = as_stream(stream);
    const int64_t rows = sp->n_ensembles * sp->W;
    if (sp->form != 0) {
        // trees, IsoTrackModel, 13-32 bands: one persistent launch, one workgroup per ensemble (fast/sampler_any.h)
        if (nsteps == 0) return ISO_OK;
        const AnyStretchArgs S = any_args(sp, pos, lnp, accepted, nsteps, chain, chain_lnp);
        sp->step += (uint32_t)nsteps;
        if (!launch_any_form(sp, S, nullptr, s)) {
            const hipError_t e = hipGetLastError();
            return fail(e == hipSuccess ? ISO_ERR_INVALID : ISO_ERR_HIP,
                        std::string("iso_sampler_run: any-model sampler launch failed") + (e == hipSuccess ? "" : std::string(": ") + hipGetErrorString(e)));
        }
        HIP_TRY(hipGetLastError());
        return ISO_OK;
    }
    StretchArgs S;
    S.occupancy_query = nullptr;
    S.dense = 0;
    S.group = 0;
    S.threads = 0;
    S.dense_stdp = 0;
    S.triple_moves = 64;  // up to how many moves per half-step a single triple runs one star per row (profiles/r06/triple_sweep.jsonl)
    S.pair = 1;           // ISOCHRONES_AMD_STAR_LANES=0: a single binary's fit through the one-lane-walks-both-stars kernel (A/B, tests)
    if (const char* e = std::getenv("ISOCHRONES_AMD_STAR_LANES")) S.pair = std::atoi(e) != 0;
    S.pos = pos;
    S.lnp = lnp;
    S.accepted = accepted;
    S.W = sp->W;
    S.multi = sp->multi;
    S.std_priors = sp->std_priors;
    S.n_active = sp->n_ensembles * (sp->W / 2);
    if (S.n_active >= (int64_t(1) << 31)) return fail(ISO_ERR_INVALID, "iso_sampler_run: more than 2^31 moves per half-step");
    S.a = sp->a;
    S.seed = sp->seed;
    S.chain_rs = sp->chain_layout == ISO_CHAIN_PARAM_MAJOR ? 1 : sp->n_params;
    S.chain_ps = sp->chain_layout == ISO_CHAIN_PARAM_MAJOR ? rows : 1;
    SamplerPlan P;
    {
        const int rc = plan_sampler_run(sp, nsteps, s, S, P);
        if (rc != ISO_OK) return rc;
    }
    const int per_cu = P.per_cu, group = P.group;
    const bool persistent = P.persistent;
    {
        int32_t* pl = iso::t_sampler_plan;
        pl[0] = persistent ? 1 : 0;
        pl[1] = S.dense;
        pl[2] = S.threads > 0 ? S.threads : BLOCK;
        pl[3] = S.dense_stdp;
        pl[4] = S.group > 0 ? S.group : group;
        pl[5] = per_cu;
        pl[6] = (int32_t)std::min<int64_t>((sp->n_ensembles + pl[4] - 1) / pl[4], 0x7fffffff);
        pl[7] = 0;
    }
    if (persistent) {
        S.step = sp->step;
        S.nsteps = nsteps;
        S.half = 0;
        S.chain_pos = chain;
        S.chain_lnp = chain_lnp;
        sp->step += (uint32_t)nsteps;
        if (!launch_stretch(sp->kind, sp->n_stars, sp->n_bands, sp->fast, S, s))
            return fail(ISO_ERR_INVALID, "iso_sampler_run: no kernel specialisation");
        HIP_TRY(hipGetLastError());
        return ISO_OK;
    }
    S.nsteps = 0;
    for (int it = 0; it < nsteps; ++it) {
        S.step = sp->step++;
        S.chain_pos = chain ? chain + (int64_t)it * rows * sp->n_params : nullptr;
        S.chain_lnp = chain_lnp ? chain_lnp + (int64_t)it * rows : nullptr;
        for (int half = 0; half < 2; ++half) {
            S.half = half;
            if (!launch_stretch(sp->kind, sp->n_stars, sp->n_bands, sp->fast, S, s))
                return fail(ISO_ERR_INVALID, "iso_sampler_run: no kernel specialisation");
        }
    }
    HIP_TRY(hipGetLastError());
    return ISO_OK;
}

int iso_chain_quantiles(iso_ctx* ctx, const double* chain, int64_t nsteps, int64_t n_ens, int W, int n_params,
                        const double* q, int nq, double* out, void* stream)
{
    return iso_chain_quantiles_layout(ctx, chain, ISO_CHAIN_ROW_MAJOR, nsteps, n_ens, W, n_params, q, nq, out, stream);
}

int iso_chain_quantiles_layout(iso_ctx* ctx, const double* chain, int layout, int64_t nsteps, int64_t n_ens, int W,
                               int n_params, const double* q, int nq, double* out, void* stream)
{
    if (!ctx || !chain || !q || !out) return fail(ISO_ERR_INVALID, "iso_chain_quantiles: NULL argument");
    if (layout != ISO_CHAIN_ROW_MAJOR && layout != ISO_CHAIN_PARAM_MAJOR)
        return fail(ISO_ERR_INVALID, "iso_chain_quantiles: unknown chain layout");
    if (nsteps < 1 || n_ens < 1 || W < 1 || n_params < 1 || nq < 1 || nq > 8)
        return fail(ISO_ERR_INVALID, "iso_chain_quantiles: counts out of range");
    const int64_t m = nsteps * W;
    if (m >= ((int64_t)1 << 31)) return fail(ISO_ERR_INVALID, "iso_chain_quantiles: 2^31 or more samples per ensemble");
    if (n_ens * n_params > 0x7fffffff) return fail(ISO_ERR_INVALID, "iso_chain_quantiles: too many ensembles");
    QuantArgs A;
    A.chain = chain;
    A.ss = n_ens * W * n_params;
    A.rs = layout == ISO_CHAIN_PARAM_MAJOR ? 1 : n_params;
    A.ps = layout == ISO_CHAIN_PARAM_MAJOR ? n_ens * W : 1;
    A.nsteps = nsteps;
    A.n_ens = n_ens;
    A.W = W;
    A.D = n_params;
    A.nq = nq;
    A.P = 2;
    while (A.P < m && A.P < 8192) A.P <<= 1;
    for (int k = 0; k < 8; ++k) A.q[k] = 0.0;
    for (int k = 0; k < nq; ++k) {
        if (!(q[k] >= 0.0 && q[k] <= 1.0)) return fail(ISO_ERR_INVALID, "iso_chain_quantiles: quantile level outside [0, 1]");
        A.q[k] = q[k];
    }
    A.out = out;
    DeviceGuard guard(ctx->device);
    // selection kernel (3 passes + tiny lists); ISOCHRONES_AMD_QUANTILES=sort forces the full LDS sort (tests / A-B)
    const char* qm = getenv("ISOCHRONES_AMD_QUANTILES");
    const size_t sel_bytes = (size_t)QSEL_RANKS * QSEL_CAP * sizeof(double) + QSEL_BINS * sizeof(int) + QSEL_BINS +
                             4 * QSEL_RANKS * sizeof(int) + 8 * sizeof(double) + 2 * sizeof(int) + 8 +
                             QSEL_RANKS * sizeof(double);
    const size_t sort_bytes = (size_t)A.P * sizeof(double);
    const dim3 g((unsigned)(n_ens * n_params)), b(BLOCK);
    A.only_flagged = 0;
    if (m > 8192) {
        // longer than the LDS forms hold (the reference's default fit: 300 walkers x 100 iterations per parameter):
        // selection by refinement, every pass streamed from the chain
        note_kernel("k_chain_quantiles_big");
        hipLaunchKernelGGL(k_chain_quantiles_big, g, b, QBIG_LDS, as_stream(stream), A);
    } else if (qm && !std::strcmp(qm, "sort")) {
        note_kernel("k_chain_quantiles");
        hipLaunchKernelGGL(k_chain_quantiles, g, b, sort_bytes, as_stream(stream), A);
    } else if (m <= 64 * QW_IPL_BIG && !(qm && !std::strcmp(qm, "workgroup"))) {
        // one wave per pair, values in registers; the few pairs it flags (heavy ties, non-finite values) go to the
        // workgroup kernel, which exits at once everywhere else
        const int64_t pairs = n_ens * n_params;
        const dim3 gw((unsigned)((pairs + QW_WAVES - 1) / QW_WAVES));
        // the common chain shapes (W divides 64; 12 / 25 / 50 / 100 full registers of 64 values, + a partial one): everything
        // per-value decided at compile time (ISOCHRONES_AMD_QUANTILES=wave keeps the generic wave kernel for A/B runs)
        const int full = m / 64;
        const bool tail = (m % 64) != 0;
        const size_t qsh = (size_t)QW_WAVES * QW_LDS_PER_WAVE;
        bool exact = W <= 64 && (64 % W) == 0 && !(qm && !std::strcmp(qm, "wave"));
        // (its loads address the chain as scalar base + 32-bit lane offset: steps 0 .. 64 / W - 1, walkers 0 .. W - 1)
        if (exact && ((64 / W) * std::llabs((long long)A.ss) + W * std::llabs((long long)A.rs)) * 8 >= ((int64_t)1 << 32)) exact = false;
        if (exact && (A.ss < 0 || A.rs < 0)) exact = false;
        if (exact) {
            hipStream_t st = as_stream(stream);
#define ISO_QEXACT(F)                                                                                      \
            case F:                                                                                        \
                note_kernel("k_chain_quantiles_exact<%d, %s>", F, tf(tail));                               \
                if (tail) hipLaunchKernelGGL((k_chain_quantiles_exact<F, true>), gw, b, qsh, st, A);       \
                else hipLaunchKernelGGL((k_chain_quantiles_exact<F, false>), gw, b, qsh, st, A);           \
                break;
            switch (full) {
                ISO_QEXACT(12) ISO_QEXACT(25) ISO_QEXACT(50) ISO_QEXACT(100)
            default: exact = false;
            }
#undef ISO_QEXACT
        }
        if (exact) {
        } else if (m <= 64 * QW_IPL) {
            note_kernel("k_chain_quantiles_wave<%d>", QW_IPL);
            hipLaunchKernelGGL(k_chain_quantiles_wave<QW_IPL>, gw, b, (size_t)QW_WAVES * QW_LDS_PER_WAVE, as_stream(stream), A);
        } else {
            note_kernel("k_chain_quantiles_wave<%d>", QW_IPL_BIG);
            hipLaunchKernelGGL(k_chain_quantiles_wave<QW_IPL_BIG>, gw, b, (size_t)QW_WAVES * QW_LDS_PER_WAVE, as_stream(stream), A);
        }
        HIP_TRY(hipGetLastError());
        A.only_flagged = 1;
        note_kernel("k_chain_quantiles_select");
        hipLaunchKernelGGL(k_chain_quantiles_select, g, b, std::max(sel_bytes, sort_bytes), as_stream(stream), A);
    } else {
        note_kernel("k_chain_quantiles_select");
        hipLaunchKernelGGL(k_chain_quantiles_select, g, b, std::max(sel_bytes, sort_bytes), as_stream(stream), A);
    }
    HIP_TRY(hipGetLastError());
    return ISO_OK;
}

int iso_time_lnpost(iso_model* m, const double* pars, int64_t stride_n, int64_t stride_p, int64_t n,
                    double* lnpost_out, int reps, void* stream, double* ms_per_launch)
{
    if (!m || !pars || !lnpost_out || !ms_per_launch || reps < 1 || n < 1)
        return fail(ISO_ERR_INVALID, "iso_time_lnpost: bad argument");
    DeviceGuard guard(m->ic->ctx->device);
    hipStream_t s = as_stream(stream);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    int rc = ISO_OK;
    HIP_TRY(hipEventRecord(e0, s));
    for (int r = 0; r < reps && rc == ISO_OK; ++r)
        rc = enqueue_lnpost(m, pars, stride_n, stride_p, n, lnpost_out, nullptr, nullptr, s);
    hipError_t e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != ISO_OK) return rc;
    if (e != hipSuccess) return fail(ISO_ERR_HIP, std::string("iso_time_lnpost: ") + hipGetErrorString(e));
    *ms_per_launch = (double)ms / reps;
    return ISO_OK;
}

int iso_time_lnpost_rotating(iso_model* m, const double* const* pars, double* const* lnpost_out, int n_batches,
                             int64_t stride_n, int64_t stride_p, int64_t n, int reps, void* stream,
                             double* ms_per_launch)
{
    if (!m || !pars || !lnpost_out || !ms_per_launch || reps < 1 || n < 1 || n_batches < 1)
        return fail(ISO_ERR_INVALID, "iso_time_lnpost_rotating: bad argument");
    for (int b = 0; b < n_batches; ++b)
        if (!pars[b] || !lnpost_out[b]) return fail(ISO_ERR_INVALID, "iso_time_lnpost_rotating: NULL batch pointer");
    DeviceGuard guard(m->ic->ctx->device);
    hipStream_t s = as_stream(stream);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    int rc = ISO_OK;
    HIP_TRY(hipEventRecord(e0, s));
    for (int r = 0; r < reps && rc == ISO_OK; ++r)
        rc = enqueue_lnpost(m, pars[r % n_batches], stride_n, stride_p, n, lnpost_out[r % n_batches], nullptr, nullptr, s);
    hipError_t e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != ISO_OK) return rc;
    if (e != hipSuccess) return fail(ISO_ERR_HIP, std::string("iso_time_lnpost_rotating: ") + hipGetErrorString(e));
    *ms_per_launch = (double)ms / reps;
    return ISO_OK;
}

}  // extern "C"
