// unit-cube transform and the one-off table packers
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

// -------------------------------------------------------------------------------------------
// small kernels
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK, 2) void k_unit_cube(const DevModel* m, double* cube, int64_t stride_n,
                                                    int64_t stride_p, int64_t n)
{
    const int np = m->n_stars + 4;
    const int64_t total = n * np;
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * BLOCK) {
        const int64_t i = e / np;
        const int p = (int)(e - i * np);
        double* c = cube + i * stride_n + p * stride_p;
        const double lo = m->bound_lo[p], hi = m->bound_hi[p];
        {
#pragma clang fp contract(off)   // unfused: bit-identical to the reference's (hi - lo) * u + lo
            const double prod = (hi - lo) * *c;
            *c = prod + lo;
        }
    }
}

struct PackHotArgs {
    const double* grid;
    int ncol;
    int64_t ncells;
    int32_t src[HOT_COLS];   // -1 -> NaN fill
    double* hot;
};

__global__ __launch_bounds__(BLOCK, 2) void k_pack_hot(const PackHotArgs A)
{
    const int64_t total = A.ncells * HOT_COLS;
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * BLOCK) {
        const int64_t cell = e / HOT_COLS;
        const int q = (int)(e - cell * HOT_COLS);
        const int s = A.src[q];
        A.hot[e] = (s >= 0) ? A.grid[cell * A.ncol + s] : d_nan();
    }
}

struct PackBcArgs {
    const double* grid;
    int ncol, nb;
    int64_t ncells;
    int32_t src[ISO_MAX_BANDS];
    double* out;
};

__global__ __launch_bounds__(BLOCK, 2) void k_pack_bc(const PackBcArgs A)
{
    const int64_t total = A.ncells * A.nb;
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * BLOCK) {
        const int64_t cell = e / A.nb;
        const int b = (int)(e - cell * A.nb);
        A.out[e] = A.grid[cell * A.ncol + A.src[b]];
    }
}

struct PackCornersArgs {
    const double* src;      // compact table, `ncol` doubles per cell
    int ncol, keep, col0;   // keep columns col0 .. col0+keep-1 of every corner (BC: col0 = 0, keep = ncol = n_bands)
    int ndim;               // 3 (model table) or 4 (BC table)
    int64_t n[4];           // axis lengths
    int64_t ncells;
    double* out;            // [cell][2^ndim * keep], laid out for the 4-lanes-per-sample gather
};

// Corner-packed layout: every cell carries its own 2^D corners, ordered so that 4 cooperating lanes
// read 64 contiguous bytes per load instruction (see iso_fast_kernel.h).
//   ndim 3: double index e = 2*(4k + j) + comp  ->  corner c = 4*(k/P) + j, column col0 + 2*(k%P) + comp,
//           P = keep/2 column pairs (3 for the model table, 1 for the asteroseismic pair)
//   ndim 4: double index e = 2*((k*NB + band)*4 + j) + comp  ->  axis-0 offset k, (axis-1, axis-2)
//           offsets = bits of j, axis-3 offset comp
__global__ __launch_bounds__(BLOCK, 2) void k_pack_corners(const PackCornersArgs A)
{
    const int per = (1 << A.ndim) * A.keep;
    const int64_t total = A.ncells * per;
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * BLOCK) {
        const int64_t cell = e / per;
        const int r = (int)(e - cell * per);
        const int comp = r & 1, piece = r >> 1, j = piece & 3, kk = piece >> 2;
        int off[4], col;
        if (A.ndim == 3) {
            const int pairs = A.keep >> 1;
            off[0] = kk / pairs; off[1] = (j >> 1) & 1; off[2] = j & 1; off[3] = 0;
            col = A.col0 + 2 * (kk % pairs) + comp;
        } else {
            off[0] = kk / A.keep; off[1] = (j >> 1) & 1; off[2] = j & 1; off[3] = comp;
            col = kk % A.keep;
        }
        int64_t rem = cell, src_cell = 0, mul = 1;
        for (int d = A.ndim - 1; d >= 0; --d) {
            int64_t id = rem % A.n[d];
            rem /= A.n[d];
            id = min(id + off[d], A.n[d] - 1);      // edge cells are never addressed (i <= n-2)
            src_cell += id * mul;
            mul *= A.n[d];
        }
        A.out[e] = A.src[src_cell * A.ncol + col];
    }
}

// -------------------------------------------------------------------------------------------
// catalog: per-star constant blocks built on the device from a template block + per-star columns
// -------------------------------------------------------------------------------------------
struct FillCatalogArgs {
    DevModel* models;          // [n]
    const DevModel* tmpl;
    int64_t n;
    int nb, i_dist;            // bands; index of the distance parameter (n_stars + 2)
    const double *mag_val, *mag_unc;      // [n][nb]
    const double *spec_val, *spec_unc;    // [n][3]
    const int32_t* has_plx;               // [n]
    const double *plx_val, *plx_unc;      // [n]
    const double* dist_hi;                // [n] or null
};

__global__ __launch_bounds__(BLOCK, 2) void k_catalog_copy_template(const FillCatalogArgs A)
{
    constexpr int64_t WORDS = sizeof(DevModel) / sizeof(double);
    static_assert(sizeof(DevModel) % sizeof(double) == 0, "DevModel must be a whole number of doubles");
    const double* __restrict__ src = reinterpret_cast<const double*>(A.tmpl);
    double* __restrict__ dst = reinterpret_cast<double*>(A.models);
    const int64_t total = A.n * WORDS;
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * BLOCK)
        dst[e] = src[e % WORDS];
}

__device__ __forceinline__ void dev_gauss_consts(double unc, double& g0, double& unc2, double& hinv)
{
    g0 = log(1.0 / sqrt(2 * M_PI)) + log(unc);
    unc2 = unc * unc;
    hinv = 0.5 / unc2;
}

__global__ __launch_bounds__(BLOCK, 2) void k_catalog_fill(const FillCatalogArgs A)
{
    for (int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x; s < A.n; s += (int64_t)gridDim.x * BLOCK) {
        DevModel& M = A.models[s];
        for (int b = 0; b < A.nb; ++b) {
            M.mag_val[b] = A.mag_val[s * A.nb + b];
            dev_gauss_consts(A.mag_unc[s * A.nb + b], M.mag_g0[b], M.mag_unc2[b], M.mag_hinv[b]);
        }
        for (int q = 0; q < 3; ++q) {
            M.spec_val[q] = A.spec_val[s * 3 + q];
            dev_gauss_consts(A.spec_unc[s * 3 + q], M.spec_g0[q], M.spec_unc2[q], M.spec_hinv[q]);
        }
        M.has_parallax = A.has_plx[s];
        M.plx_val = A.plx_val[s];
        dev_gauss_consts(A.plx_unc[s], M.plx_g0, M.plx_unc2, M.plx_hinv);
        if (A.dist_hi) {
            const double hi = A.dist_hi[s];
            DevPrior& P = M.prior_distance;
            P.hi = hi;
            if (P.kind == ISO_PRIOR_FLAT) {
                P.k0 = 1.0 / (hi - P.lo);
                P.k1 = log(P.k0);
            } else if (P.kind == ISO_PRIOR_FLATLOG) {
                P.k0 = pow(10.0, hi) - pow(10.0, P.lo);
                P.k1 = log(log(10.0) / P.k0);
            } else if (P.kind == ISO_PRIOR_POWERLAW) {
                P.k0 = (1 + P.a) / (pow(hi, 1 + P.a) - pow(P.lo, 1 + P.a));
                P.k1 = log(P.k0);
            }
            M.bound_hi[A.i_dist] = hi;
        }
    }
}

// A catalog fit keeps its batch rectangular: a star whose start-point search failed (fewer than W finite candidates: its
// rows are NaN) borrows the walkers of the batch's first good star with lnpost 0 - its own (hopeless) posterior then never
// accepts a move, and its result row is blanked at the end (catalog.py: fit_stars_gpu; the reference isolates a failing
// star with try / except, isochrones/starfit.py:155-159).  One workgroup per star; on the stream, no host round trip.
__global__ __launch_bounds__(BLOCK, 2) void k_catalog_patch_failed(double* __restrict__ pos, double* __restrict__ lnp,
                                                                  const int32_t* __restrict__ failed, int64_t n_stars, int W, int D)
{
    const int64_t s = blockIdx.x;
    if (!failed[s]) return;
    __shared__ int64_t src;
    if (threadIdx.x == 0) {
        int64_t k = 0;
        while (k < n_stars && failed[k]) ++k;
        src = k;
    }
    __syncthreads();
    const int64_t from = src;
    if (from >= n_stars) {                       // no star of the batch has start points: the rows stay NaN
        for (int j = threadIdx.x; j < W; j += BLOCK) lnp[s * W + j] = 0.0;
        return;
    }
    for (int j = threadIdx.x; j < W * D; j += BLOCK) pos[s * W * D + j] = pos[from * W * D + j];
    for (int j = threadIdx.x; j < W; j += BLOCK) lnp[s * W + j] = 0.0;
}
