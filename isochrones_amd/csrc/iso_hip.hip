// isochrones_amd — hand-written HIP (gfx950 / CDNA4) implementation of the isochrones hot path
// behind the C ABI of include/isochrones_amd.h.
//
// Kernels (all float64, gather/latency bound — no MFMA on purpose, this is interpolation):
//   k_interp<ND>        K3  N-D bracket search + multilinear gather of an arbitrary column subset
//                           (reference semantics: isochrones/interp.py:10-35, 63-338)
//   k_interp_mag        K4  3-D model gather -> 4-D BC gather -> magnitudes
//                           (reference semantics: isochrones/mags.py:8-124)
//   k_lnpost<...>       K1+K2 fused: priors + 1-3 component stars + likelihood reduce
//                           (isochrones/likelihood.py:16-147, starmodel.py:538-542,1563-1635,
//                            priors.py lnpdf's)
//   k_unit_cube             mnest_prior (starmodel.py:1637-1640)
//   k_pack_hot / k_pack_bc  one-off table repacks (hot columns AoS; model's bands only)
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

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

// ======================================================================================
// host-side bookkeeping
// ======================================================================================
#include "iso_internal.h"

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

__device__ __forceinline__ double d_nan() { return __longlong_as_double(0x7ff8000000000000LL); }
__device__ __forceinline__ double d_inf() { return __longlong_as_double(0x7ff0000000000000LL); }

// Cooperative copy of the non-uniform axes into LDS.  Must be followed by __syncthreads().
template <int NAX>
__device__ __forceinline__ void stage_axes(const AxisD* ax, double* lds)
{
#pragma unroll
    for (int d = 0; d < NAX; ++d) {
        if (ax[d].lds_off >= 0) {
            const double* __restrict__ src = ax[d].g;
            double* dst = lds + ax[d].lds_off;
            for (int j = threadIdx.x; j < ax[d].n; j += blockDim.x) dst[j] = src[j];
        }
    }
}

// Branch-free bisection on a sorted axis: base = largest index with a[base] <= x, clamped to n-2.
template <typename PTR>
__device__ __forceinline__ void bisect(PTR ax, int n, double x, int& i, double& t)
{
    int base = 0, len = n;
    while (len > 1) {
        const int half = len >> 1;
        base = (ax[base + half] <= x) ? base + half : base;
        len -= half;
    }
    base = min(base, n - 2);
    const double lo = ax[base], hi = ax[base + 1];
    i = base;
    t = (x - lo) / (hi - lo);
}

// Bracket of x on one axis.  Precondition: a_0 <= x <= a_{n-1} (caller has done the bounds
// test, reference isochrones/interp.py:106-114).
__device__ __forceinline__ void bracket(const AxisD& A, const double* lds, double x, int& i, double& t)
{
    const int n = A.n;
    if (A.uniform) {
        // O(1) index with an exact fix-up against the node values (node(i) reproduces the stored
        // axis value bit-for-bit, verified on the host at table creation).
        const double a0 = A.a0, st = A.step;
        int k = (int)((x - a0) / st);
        k = max(0, min(k, n - 2));
        double lo = fma((double)k, st, a0);
        if (lo > x) {
            --k;
        } else if (k < n - 2 && fma((double)(k + 1), st, a0) <= x) {
            ++k;
        }
        k = max(0, min(k, n - 2));
        lo = fma((double)k, st, a0);
        const double hi = fma((double)(k + 1), st, a0);
        i = k;
        t = (x - lo) / (hi - lo);
        return;
    }
    if (A.lds_off >= 0) bisect(lds + A.lds_off, n, x, i, t);   // LDS address space (ds_read)
    else bisect(A.g, n, x, i, t);                              // global
}

__device__ __forceinline__ bool out_of_axis(const AxisD& A, const double* lds, double x)
{
    double first, last;
    if (A.uniform) {
        first = A.a0;
        last = fma((double)(A.n - 1), A.step, A.a0);
    } else {
        if (A.lds_off >= 0) {
            first = lds[A.lds_off];
            last = lds[A.lds_off + A.n - 1];
        } else {
            first = A.g[0];
            last = A.g[A.n - 1];
        }
    }
    // written so that NaN is *not* out of bounds here (the NaN test comes first in the reference)
    return (x < first) || (x > last);
}

// -------------------------------------------------------------------------------------------
// K3: generic N-D interpolation of k selected columns
// -------------------------------------------------------------------------------------------
struct InterpArgs {
    AxisD ax[ISO_MAX_DIM];
    int64_t stride[ISO_MAX_DIM];   // cell strides
    const double* grid;
    int ncol;
    const double* x[ISO_MAX_DIM];
    int64_t n;
    int k;
    int32_t icols[ISO_MAX_COLS];
    double* out;
};

// Column-parallel mapping: G = ceil(k/2) adjacent lanes share one sample, lane `sub` owns the
// selected columns 2*sub and 2*sub+1.  For every corner the G lanes read neighbouring columns of
// the same table row (one or two cache lines) and finally write k contiguous doubles — coalesced
// loads and stores with no cross-lane reduction; the bracket search is repeated by the G lanes
// (cheap: ~150 VALU against >= 1 KB of gathered table per sample).  k = 1, 2 degenerate to one lane
// per sample.
template <int ND>
__global__ __launch_bounds__(BLOCK) void k_interp(const InterpArgs A)
{
    extern __shared__ double lds[];
    stage_axes<ND>(A.ax, lds);
    __syncthreads();
    const int G = (A.k + 1) >> 1;            // lanes per sample
    const int S = 64 / G;                    // samples per wave
    const int lane = threadIdx.x & 63;
    const int slot = lane / G, sub = lane - slot * G;
    if (slot >= S) return;                   // leftover lanes (no cross-lane operations below)
    const int c0 = A.icols[2 * sub];
    const bool two = (2 * sub + 1) < A.k;
    const int c1 = two ? A.icols[2 * sub + 1] : c0;
    const int64_t wave0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    for (int64_t i = wave0 * S + slot; i < A.n; i += nwaves * S) {
        double x[ND];
        bool bad = false;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            x[d] = A.x[d][i];
            bad |= (x[d] != x[d]);
        }
        if (!bad) {
#pragma unroll
            for (int d = 0; d < ND; ++d) bad |= out_of_axis(A.ax[d], lds, x[d]);
        }
        double* o = A.out + i * A.k + 2 * sub;
        if (bad) {
            o[0] = d_nan();
            if (two) o[1] = d_nan();
            continue;
        }
        double t[ND];
        int64_t base = 0;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            int idx;
            bracket(A.ax[d], lds, x[d], idx, t[d]);
            base += (int64_t)idx * A.stride[d];
        }
        double v0 = 0.0, v1 = 0.0;
#pragma unroll
        for (int j = 0; j < (1 << ND); ++j) {
            double ww = 1.0;
            int64_t oo = base;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int bit = (j >> (ND - 1 - d)) & 1;
                ww *= bit ? t[d] : (1 - t[d]);
                oo += bit ? A.stride[d] : 0;
            }
            const double* __restrict__ cell = A.grid + oo * A.ncol;
            v0 += cell[c0] * ww;
            v1 += cell[c1] * ww;
        }
        o[0] = v0;
        if (two) o[1] = v1;
    }
}

// -------------------------------------------------------------------------------------------
// K3 on the "wide pack" of a 3-D table: layout [cell][column][corner 0..7] (corner bit2/bit1/bit0 = +1
// on axis 0/1/2), i.e. the 8 corner values one column needs are one aligned 64-B piece.  A selected
// column costs one such piece instead of 8 scattered rows, for any column subset; the price is 8x the
// table in HBM (5.8 GB for the MIST track table - this part has 288 GB).  Built on the first large batch.
//
// One lane owns one sample for the bracket search and publishes (cell, t0, t1, t2) in a wave-private
// LDS slot.  The wave's 64 x k (sample, column) units are then served by quads, 16 units per wave
// instruction in row-major order of the output: each lane of a quad loads 16 B (two corners that
// differ on axis 2), weights them, two DPP quad-permute adds finish the 8-corner sum, and the 16
// results of a pass leave as one contiguous 128-B store.
// -------------------------------------------------------------------------------------------
struct WideArgs {
    AxisD ax[3];
    int64_t stride[3];
    const double* wide;      // [ncells][ncol][8]
    int ncol;
    const double* x[3];
    int64_t n;
    int k;
    uint64_t kinv;           // floor(2^32 / k) + 1: u / k == (u * kinv) >> 32 for u < 2^16 (k = 1: 2^32 + 1)
    int lds_axes;            // doubles of staged axes
    int32_t icols[ISO_MAX_COLS];
    double* out;
};

struct PackWideArgs {
    const double* grid;      // [n0][n1][n2][ncol]
    double* out;
    int64_t n0, n1, n2;
    int ncol;
};

__global__ __launch_bounds__(BLOCK) void k_pack_wide(const PackWideArgs P)
{
    const int64_t total = P.n0 * P.n1 * P.n2 * P.ncol * 8;
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * BLOCK) {
        const int j = (int)(e & 7);
        const int64_t r = e >> 3;
        const int col = (int)(r % P.ncol);
        const int64_t cell = r / P.ncol;
        const int64_t i2 = cell % P.n2, i1 = (cell / P.n2) % P.n1, i0 = cell / (P.n2 * P.n1);
        // the last cell of an axis is never a bracket's lower corner; its "+1" entries repeat the edge
        const int64_t a0 = min(i0 + ((j >> 2) & 1), P.n0 - 1), a1 = min(i1 + ((j >> 1) & 1), P.n1 - 1),
                      a2 = min(i2 + (j & 1), P.n2 - 1);
        P.out[e] = P.grid[((a0 * P.n1 + a1) * P.n2 + a2) * P.ncol + col];
    }
}

__device__ __forceinline__ double wide_dpp(double x, int which)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    if (which == 0) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false);
    }
    return __hiloint2double(hi, lo);
}

constexpr int WIDE_SLOT = 5;      // doubles per request slot (4 used; odd stride: conflict-free)
constexpr int WIDE_UNROLL = 8;    // passes whose loads are in flight together

__global__ __launch_bounds__(BLOCK) void k_interp3_wide(const WideArgs A)
{
    extern __shared__ double lds[];
    stage_axes<3>(A.ax, lds);
    int32_t* lcols = reinterpret_cast<int32_t*>(lds + A.lds_axes);
    for (int j = threadIdx.x; j < A.k; j += BLOCK) lcols[j] = A.icols[j];
    __syncthreads();
    double* slots = lds + A.lds_axes + (ISO_MAX_COLS / 2) + (threadIdx.x >> 6) * 64 * WIDE_SLOT;
    const int lane = threadIdx.x & 63;
    const int64_t first = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) & ~(int64_t)63;   // the wave's first sample
    const int64_t i = first + lane;
    {
        bool bad = i >= A.n;
        double x[3] = {0.0, 0.0, 0.0};
        if (!bad) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                x[d] = A.x[d][i];
                bad |= (x[d] != x[d]);
            }
        }
        if (!bad) {
#pragma unroll
            for (int d = 0; d < 3; ++d) bad |= out_of_axis(A.ax[d], lds, x[d]);
        }
        double t[3] = {0.0, 0.0, 0.0};
        int64_t cell = -1;
        if (!bad) {
            cell = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                int idx;
                bracket(A.ax[d], lds, x[d], idx, t[d]);
                cell += (int64_t)idx * A.stride[d];
            }
        }
        double* mine = slots + lane * WIDE_SLOT;
        mine[0] = __longlong_as_double(cell);
        mine[1] = t[0];
        mine[2] = t[1];
        mine[3] = t[2];
    }
    __builtin_amdgcn_wave_barrier();
    const int j = lane & 3, grp = lane >> 2;
    const int k = A.k;
    const int here = (int)min((int64_t)64, A.n - first);       // samples of this wave
    const int units = here * k;
    double* __restrict__ out = A.out + first * k;
    for (int u0 = 0; u0 < units; u0 += 16 * WIDE_UNROLL) {
        double2 v[WIDE_UNROLL];
        double wx[WIDE_UNROLL], wy[WIDE_UNROLL];
        bool bad[WIDE_UNROLL];
#pragma unroll
        for (int r = 0; r < WIDE_UNROLL; ++r) {
            const int u = min(u0 + 16 * r + grp, units - 1);
            const int s = (int)(((uint64_t)(uint32_t)u * A.kinv) >> 32);      // u / k
            const int c = u - s * k;
            const double* rq = slots + s * WIDE_SLOT;
            const long long cell = __double_as_longlong(rq[0]);
            const double t0 = rq[1], t1 = rq[2], t2 = rq[3];
            bad[r] = cell < 0;
            const int64_t cc = bad[r] ? 0 : cell;
            v[r] = *reinterpret_cast<const double2*>(A.wide + ((cc * A.ncol + lcols[c]) << 3) + 2 * j);
            const double g = ((j & 2) ? t0 : (1 - t0)) * ((j & 1) ? t1 : (1 - t1));
            wx[r] = g * (1 - t2);
            wy[r] = g * t2;
        }
#pragma unroll
        for (int r = 0; r < WIDE_UNROLL; ++r) {
            double part = v[r].x * wx[r] + v[r].y * wy[r];
            part += wide_dpp(part, 0);
            part += wide_dpp(part, 1);
            const int u = u0 + 16 * r + grp;
            if (j == 0 && u < units) out[u] = bad[r] ? d_nan() : part;
        }
    }
}

// -------------------------------------------------------------------------------------------
// shared device pieces of K4 / K1+K2
// -------------------------------------------------------------------------------------------

// 3-D bracket of one star on the hot table.  Returns false (values undefined) if NaN / oob.
struct Cell3 {
    int64_t base;     // cell index of the (i0,i1,i2) corner
    double t0, t1, t2;
};

__device__ __forceinline__ bool locate3(const Grid3V& G, const double* lds, double x0, double x1, double x2,
                                        Cell3& c)
{
    if (x0 != x0 || x1 != x1 || x2 != x2) return false;
    if (out_of_axis(G.ax[0], lds, x0) || out_of_axis(G.ax[1], lds, x1) || out_of_axis(G.ax[2], lds, x2))
        return false;
    int i0, i1, i2;
    bracket(G.ax[0], lds, x0, i0, c.t0);
    bracket(G.ax[1], lds, x1, i1, c.t1);
    bracket(G.ax[2], lds, x2, i2, c.t2);
    c.base = (int64_t)i0 * G.s0 + (int64_t)i1 * G.s1 + i2;
    return true;
}

// Gather the first NC hot columns of the 8 corners (corner order and weight products as the
// reference: bit (2-k) of j offsets axis k; weight = ((1 * w0) * w1) * w2).
template <int NC>
__device__ __forceinline__ void gather3(const Grid3V& G, const Cell3& c, double* __restrict__ v)
{
#pragma unroll
    for (int q = 0; q < NC; ++q) v[q] = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int b0 = (j >> 2) & 1, b1 = (j >> 1) & 1, b2 = j & 1;
        double w = 1.0;
        w *= b0 ? c.t0 : (1 - c.t0);
        w *= b1 ? c.t1 : (1 - c.t1);
        w *= b2 ? c.t2 : (1 - c.t2);
        const int64_t cell = c.base + (b0 ? G.s0 : 0) + (b1 ? G.s1 : 0) + b2;
        const double2* __restrict__ p = reinterpret_cast<const double2*>(G.hot + cell * HOT_COLS);
#pragma unroll
        for (int q = 0; q < NC; q += 2) {
            const double2 u = p[q >> 1];
            v[q] += u.x * w;
            if (q + 1 < NC) v[q + 1] += u.y * w;
        }
    }
}

struct Cell4 {
    int64_t base;
    double t0, t1, t2, t3;
};

__device__ __forceinline__ bool locate4(const Grid4V& G, const double* lds, double x0, double x1, double x2,
                                        double x3, Cell4& c)
{
    if (x0 != x0 || x1 != x1 || x2 != x2 || x3 != x3) return false;
    if (out_of_axis(G.ax[0], lds, x0) || out_of_axis(G.ax[1], lds, x1) || out_of_axis(G.ax[2], lds, x2) ||
        out_of_axis(G.ax[3], lds, x3))
        return false;
    int i0, i1, i2, i3;
    bracket(G.ax[0], lds, x0, i0, c.t0);
    bracket(G.ax[1], lds, x1, i1, c.t1);
    bracket(G.ax[2], lds, x2, i2, c.t2);
    bracket(G.ax[3], lds, x3, i3, c.t3);
    c.base = (int64_t)i0 * G.s0 + (int64_t)i1 * G.s1 + (int64_t)i2 * G.s2 + i3;
    return true;
}

__device__ __forceinline__ double weight4(const Cell4& c, int j)
{
    double w = 1.0;
    w *= ((j >> 3) & 1) ? c.t0 : (1 - c.t0);
    w *= ((j >> 2) & 1) ? c.t1 : (1 - c.t1);
    w *= ((j >> 1) & 1) ? c.t2 : (1 - c.t2);
    w *= (j & 1) ? c.t3 : (1 - c.t3);
    return w;
}

__device__ __forceinline__ int64_t corner4(const Grid4V& G, const Cell4& c, int j)
{
    return c.base + (((j >> 3) & 1) ? G.s0 : 0) + (((j >> 2) & 1) ? G.s1 : 0) + (((j >> 1) & 1) ? G.s2 : 0) +
           (j & 1);
}

// one column of the BC table at a located cell
__device__ __forceinline__ double gather4_col(const Grid4V& G, const Cell4& c, int col)
{
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += G.tab[corner4(G, c, j) * G.ncol + col] * weight4(c, j);
    return v;
}

// NB contiguous columns (packed BC table, ncol == NB) at a located cell
template <int NB>
__device__ __forceinline__ void gather4_packed(const Grid4V& G, const Cell4& c, double* __restrict__ v)
{
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double w = weight4(c, j);
        const double* __restrict__ p = G.tab + corner4(G, c, j) * NB;
#pragma unroll
        for (int b = 0; b < NB; ++b) v[b] += p[b] * w;
    }
}

// parameter permutation: (mass, eep, feh) -> table axes (feh, mass, eep);  (eep, age, feh) -> (age, feh, eep)
template <int KIND>
__device__ __forceinline__ void to_axes(double p0, double p1, double p2, double& x0, double& x1, double& x2)
{
    if (KIND == ISO_KIND_TRACK) {
        x0 = p2; x1 = p0; x2 = p1;
    } else {
        x0 = p1; x1 = p2; x2 = p0;
    }
}

// -------------------------------------------------------------------------------------------
// K4: interp_mag
// -------------------------------------------------------------------------------------------
struct MagArgs {
    Grid3V g3;
    Grid4V g4;
    int kind;
    const double* pars;
    int64_t stride_n, stride_p, n;
    int nb;
    int32_t bc_cols[ISO_MAX_BANDS];
    double *Teff, *logg, *feh, *mags;
};

// Column-parallel like k_interp: G = max(1, ceil(nb/2)) adjacent lanes share one sample; every lane
// repeats the (cheap) model-table gather of the four stellar columns — the G lanes read identical
// addresses, i.e. one request — and the BC brackets, then lane `sub` owns the bands 2*sub, 2*sub+1:
// per corner the G lanes read neighbouring columns of one BC row and finally write nb contiguous
// magnitudes.
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_interp_mag(const MagArgs A)
{
    extern __shared__ double lds[];
    stage_axes<3>(A.g3.ax, lds);
    stage_axes<4>(A.g4.ax, lds);
    __syncthreads();
    const int G = (A.nb + 1) >> 1 > 0 ? (A.nb + 1) >> 1 : 1;
    const int S = 64 / G;
    const int lane = threadIdx.x & 63;
    const int slot = lane / G, sub = lane - slot * G;
    if (slot >= S) return;
    const bool has0 = (2 * sub) < A.nb, has1 = (2 * sub + 1) < A.nb;
    const int c0 = has0 ? A.bc_cols[2 * sub] : 0, c1 = has1 ? A.bc_cols[2 * sub + 1] : c0;
    const int64_t wave0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    for (int64_t i = wave0 * S + slot; i < A.n; i += nwaves * S) {
        const double* __restrict__ p = A.pars + i * A.stride_n;
        const double p0 = p[0], p1 = p[A.stride_p], p2 = p[2 * A.stride_p];
        const double dist = p[3 * A.stride_p], AV = p[4 * A.stride_p];
        double x0, x1, x2;
        to_axes<KIND>(p0, p1, p2, x0, x1, x2);
        double star[4] = {d_nan(), d_nan(), d_nan(), d_nan()};
        Cell3 c3;
        if (locate3(A.g3, lds, x0, x1, x2, c3)) gather3<4>(A.g3, c3, star);
        if (sub == 0) {
            if (A.Teff) A.Teff[i] = star[0];
            if (A.logg) A.logg[i] = star[1];
            if (A.feh) A.feh[i] = star[2];
        }
        if (A.mags && has0) {
            Cell4 c4;
            const bool ok = locate4(A.g4, lds, star[0], star[1], star[2], AV, c4);
            const double dm = 5 * log10(dist / 10.0);
            double b0 = d_nan(), b1 = d_nan();
            if (ok) {
                b0 = b1 = 0.0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const double* __restrict__ row = A.g4.tab + corner4(A.g4, c4, j) * A.g4.ncol;
                    const double ww = weight4(c4, j);
                    b0 += row[c0] * ww;
                    b1 += row[c1] * ww;
                }
            }
            double* o = A.mags + i * A.nb + 2 * sub;
            o[0] = star[3] + dm - b0;
            if (has1) o[1] = star[3] + dm - b1;
        }
    }
}

// -------------------------------------------------------------------------------------------
// priors
// -------------------------------------------------------------------------------------------
#define LOG_INV_ROOT_2PI (-0.91893853320467267)   // log(1/sqrt(2 pi))
#define INV_ROOT_2PI 0.3989422804014327
#define LN10 2.302585092994046

__device__ __forceinline__ double lognormal_pdf(const DevPrior& P, double x, double mu, double sigma,
                                                double scale)
{
    const double y = x / scale;
    const double ly = log(y) / sigma;
    return INV_ROOT_2PI / (sigma * y) * exp(-0.5 * (ly * ly)) / scale;
}

__device__ __forceinline__ double lognormal_lnpdf(double x, double mu, double sigma, double scale,
                                                  double log_sigma)
{
    const double y = x / scale;
    const double l = log(y);
    const double ly = l / sigma;
    return LOG_INV_ROOT_2PI - (log_sigma + l) - 0.5 * (ly * ly) - mu;
}

__device__ __forceinline__ double feh_shape(const DevPrior& P, double feh)
{
    double disk;
    if (P.c != 0.0) {
        const double u = feh - 0.016, v = feh + 0.15;
        disk = 1.0 / 2.5066282746310007 *
               (0.8 / 0.15 * exp(-0.5 * (u * u) / (0.15 * 0.15)) + 0.2 / 0.22 * exp(-0.5 * (v * v) / (0.22 * 0.22)));
    } else {
        const double u = feh + 0.3;
        disk = INV_ROOT_2PI / 0.3 * exp(-0.5 * (u * u) / (0.3 * 0.3));
    }
    const double h = feh + 1.5;
    const double halo = P.k0 * exp(-0.5 * (h * h) / (0.4 * 0.4));   // k0 = 1/sqrt(2 pi 0.4^2)
    return P.a * halo + (1 - P.a) * disk;
}

// _pdf(x) of a family (no bounds handling)
__device__ double prior_raw(const DevPrior& P, double x)
{
    switch (P.kind) {
    case ISO_PRIOR_FLAT: return P.k0;                                  // 1/(hi-lo)
    case ISO_PRIOR_FLATLOG: return LN10 * exp10(x) / P.k0;             // k0 = 10^hi - 10^lo
    case ISO_PRIOR_POWERLAW: return P.k0 * pow(x, P.a);                // k0 = C
    case ISO_PRIOR_GAUSS: {
        const double z = (x - P.a) / P.b;
        return exp(-(z * z) / 2.0) * INV_ROOT_2PI / P.b / P.k0;        // k0 = exp(lognorm)
    }
    case ISO_PRIOR_LOGNORMAL: return lognormal_pdf(P, x, P.a, P.b, P.k0);   // k0 = exp(mu)
    case ISO_PRIOR_CHABRIER:
        if (x < P.d) {
            const double c = (x < 0) ? 0.0 : lognormal_pdf(P, x, P.a, P.b, P.k0);
            return c / P.e;
        } else {
            const double c = (x < P.g || x > P.h) ? 0.0 : P.k2 * pow(x, P.c);   // k2 = C of the power law
            return c / P.f;
        }
    case ISO_PRIOR_FEH: return feh_shape(P, x);
    }
    return d_nan();
}

// prior(x): the reference's __call__ form (pdf with its bounds tests)
__device__ double prior_call(const DevPrior& P, double x)
{
    if (P.kind == ISO_PRIOR_LOGNORMAL) {
        if (x < 0) return 0.0;
        return lognormal_pdf(P, x, P.a, P.b, P.k0);
    }
    if (x < P.lo || x > P.hi) return 0.0;
    const double r = prior_raw(P, x);
    return (P.kind == ISO_PRIOR_FEH) ? r / P.b : r;
}

__device__ double prior_lnpdf(const DevPrior& P, double x)
{
    switch (P.kind) {
    case ISO_PRIOR_FLAT:
    case ISO_PRIOR_FLATLOG: {
        if (x < P.lo || x > P.hi) return -d_inf();
        const double pdf = prior_raw(P, x);
        return pdf != 0 ? log(pdf) : -d_inf();
    }
    case ISO_PRIOR_POWERLAW:
        if (P.bounded && (x < P.lo || x > P.hi)) return -d_inf();
        return P.k1 + P.a * log(x);                                    // k1 = log(C)
    case ISO_PRIOR_GAUSS: {
        if (P.bounded && (x < P.lo || x > P.hi)) return -d_inf();
        const double z = (x - P.a) / P.b;
        return (-(z * z) / 2.0 + LOG_INV_ROOT_2PI) - P.k1 - P.c;       // k1 = log(sigma)
    }
    case ISO_PRIOR_LOGNORMAL: return lognormal_lnpdf(x, P.a, P.b, P.k0, P.k1);   // k1 = log(sigma)
    case ISO_PRIOR_CHABRIER:
        if (x < P.d) return lognormal_lnpdf(x, P.a, P.b, P.k0, P.k1) - P.k3;     // k3 = log(e)
        if (x < P.g || x > P.h) return -d_inf();
        return (P.k5 + P.c * log(x)) - P.k4;                     // k5 = log(C), k4 = log(f)
    case ISO_PRIOR_FEH: {
        const double pdf = prior_call(P, x);
        return pdf != 0 ? log(pdf) : -d_inf();
    }
    }
    return d_nan();
}

__device__ __forceinline__ double gauss_term(double val, double g0, double unc2, double model)
{
    const double r = val - model;
    return g0 - 0.5 * r * r / unc2;
}

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
template <int KIND, int NS, int NB, bool PARTS>
__global__ __launch_bounds__(BLOCK) void k_lnpost(const PostArgs A)
{
    extern __shared__ double lds[];
    stage_axes<3>(A.g3.ax, lds);
    stage_axes<4>(A.g4.ax, lds);
    __syncthreads();
    const DevModel& M = *A.m;
    constexpr int NP = NS + 4;
    const int64_t stride_grid = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < A.n; i += stride_grid) {
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
                gather3<6>(A.g3, c3[s], star[s]);
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
                        gather4_packed<(NB > 0 ? NB : 1)>(A.g4, c4[s], bc);
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
                        const double bc = ok4[s] ? gather4_col(A.g4, c4[s], b) : d_nan();
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
                double a2[8];
                if (ok3[0]) {
                    gather3<8>(A.g3, c3[0], a2);
                } else {
                    a2[6] = a2[7] = d_nan();
                }
                lnl += gauss_term(M.numax_val, M.numax_g0, M.numax_unc2, a2[6]);
                if (M.has_dnu) lnl += gauss_term(M.dnu_val, M.dnu_g0, M.dnu_unc2, a2[7]);
            }
        }
        if (A.lnpost) A.lnpost[i] = prior_ok ? lnp + lnl : -d_inf();
        if (PARTS) {
            if (A.lnprior) A.lnprior[i] = lnp;
            if (A.lnlike) A.lnlike[i] = lnl;
        }
    }
}

// -------------------------------------------------------------------------------------------
// "next" row f2: (age, feh, mass) -> EEP on the ragged per-track age arrays
// (reference semantics: isochrones/interp.py:488-558 interp_eep / interp_eeps)
// -------------------------------------------------------------------------------------------
struct EepArgs {
    AxisD ax[2];              // feh, mass
    const double* ages;       // [n0*n1][n_eep], NaN past `lengths`
    const int64_t* lengths;   // [n0*n1]
    int n1;
    int64_t n_eep;
    double eep0;              // EEP of array index 0 (1 for MIST)
    const double *x, *x0, *x1;
    int64_t n;
    double* out;
};

// number of elements of arr[0..N) that are < x  (== the reference's searchsorted L)
__device__ __forceinline__ int64_t count_less(const double* __restrict__ arr, double x, int64_t N)
{
    int64_t base = 0, len = N;
    if (N <= 0) return 0;
    while (len > 1) {                       // base = largest index with arr[base] < x, or 0
        const int64_t half = len >> 1;
        base = (arr[base + half] < x) ? base + half : base;
        len -= half;
    }
    return (arr[base] < x) ? base + 1 : base;
}

__global__ __launch_bounds__(BLOCK) void k_interp_eep(const EepArgs A)
{
    extern __shared__ double lds[];
    stage_axes<2>(A.ax, lds);
    __syncthreads();
    const int64_t stride_grid = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < A.n; i += stride_grid) {
        const double x = A.x[i], x0 = A.x0[i], x1 = A.x1[i];
        double r = d_nan();
        if (!(x != x || x0 != x0 || x1 != x1) && !out_of_axis(A.ax[0], lds, x0) && !out_of_axis(A.ax[1], lds, x1)) {
            int i0, i1;
            double d0, d1;
            bracket(A.ax[0], lds, x0, i0, d0);
            bracket(A.ax[1], lds, x1, i1, d1);
            const int64_t ind[4] = {(int64_t)i0 * A.n1 + i1, (int64_t)i0 * A.n1 + i1 + 1,
                                    (int64_t)(i0 + 1) * A.n1 + i1, (int64_t)(i0 + 1) * A.n1 + i1 + 1};
            int64_t ie[4], len[4];
            bool bad = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                len[k] = A.lengths[ind[k]];
                ie[k] = count_less(A.ages + ind[k] * A.n_eep, x, len[k]);
                bad |= ie[k] > A.n_eep - 1;
            }
            if (!bad) {
                double e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = A.eep0 + (double)ie[k];
                if (ie[0] >= len[0]) e[0] = e[1];      // sequential substitution, as the reference
                if (ie[1] >= len[1]) e[1] = e[0];
                if (ie[2] >= len[2]) e[2] = e[3];
                if (ie[3] >= len[3]) e[3] = e[2];
                const double e_0 = (1 - d1) * e[0] + d1 * e[1];
                const double e_1 = (1 - d1) * e[2] + d1 * e[3];
                r = (1 - d0) * e_0 + d0 * e_1;
            }
        }
        A.out[i] = r;
    }
}

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

__global__ __launch_bounds__(BLOCK) void k_lnpost_tree(const TreeArgs A)
{
    extern __shared__ double lds[];
    stage_axes<3>(A.g3.ax, lds);
    stage_axes<4>(A.g4.ax, lds);
    __syncthreads();
    const DevTree& T = *A.T;
    const int64_t stride_grid = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < A.n; i += stride_grid) {
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

// -------------------------------------------------------------------------------------------
// posterior summaries of a stored chain: one workgroup sorts the nsteps*W values of one
// (ensemble, parameter) pair in LDS (bitonic network on the next power of two, padded with +inf)
// and writes the requested quantiles (linear interpolation between order statistics)
// -------------------------------------------------------------------------------------------
struct QuantArgs {
    const double* chain;     // [nsteps][n_ens*W][D]
    int64_t nsteps, n_ens;
    int W, D, nq, P;         // P = power of two >= nsteps*W
    double q[8];
    double* out;             // [n_ens][D][nq]
};

__global__ __launch_bounds__(BLOCK) void k_chain_quantiles(const QuantArgs A)
{
    extern __shared__ double lds[];
    const int64_t e = blockIdx.x / A.D;
    const int d = (int)(blockIdx.x - e * A.D);
    const int m = (int)(A.nsteps * A.W);
    const int64_t rows = A.n_ens * A.W;
    for (int i = threadIdx.x; i < A.P; i += BLOCK) {
        double v = d_inf();
        if (i < m) {
            const int t = i / A.W, w = i - t * A.W;
            v = A.chain[((int64_t)t * rows + e * A.W + w) * A.D + d];
            if (v != v) v = d_inf();                 // NaN sorts last (cannot occur in an accepted chain)
        }
        lds[i] = v;
    }
    __syncthreads();
    for (int k = 2; k <= A.P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < A.P; i += BLOCK) {
                const int x = i ^ j;
                if (x > i) {
                    const double a = lds[i], b = lds[x];
                    const bool asc = (i & k) == 0;
                    if ((a > b) == asc) {
                        lds[i] = b;
                        lds[x] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    if ((int)threadIdx.x < A.nq) {
        double pos;
        {
#pragma clang fp contract(off)   // numpy's virtual index for method="linear": n q + (1 + q (1 - 1 - 1)) - 1
            const double qq = A.q[threadIdx.x];
            pos = ((double)m * qq + (1.0 + qq * -1.0)) - 1.0;
        }
        int i0 = (int)floor(pos);
        i0 = max(0, min(i0, m - 1));
        const int i1 = min(i0 + 1, m - 1);
        const double f = pos - (double)i0;
        const double a = lds[i0], b = lds[i1];
        double r;
        {
#pragma clang fp contract(off)   // numpy's _lerp, unfused: a + (b-a)t, from the upper end for t >= 0.5
            const double diff = b - a;
            r = (f >= 0.5) ? b - diff * (1 - f) : a + diff * f;
        }
        A.out[(e * A.D + d) * A.nq + threadIdx.x] = r;
    }
}

// -------------------------------------------------------------------------------------------
// small kernels
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_unit_cube(const DevModel* m, double* cube, int64_t stride_n,
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

__global__ __launch_bounds__(BLOCK) void k_pack_hot(const PackHotArgs A)
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

__global__ __launch_bounds__(BLOCK) void k_pack_bc(const PackBcArgs A)
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
__global__ __launch_bounds__(BLOCK) void k_pack_corners(const PackCornersArgs A)
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

constexpr int FAST_MAX_BLOB = 4096;   // doubles (32 KiB of LDS) the fast kernel may stage

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
    const size_t bytes = (size_t)P.ncells * (size_t)(1 << ndim) * keep * sizeof(double);
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) return e;
    P.out = *out;
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
};

struct iso_eep_table {
    int device;
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
void free_mag_pack(MagPack& mp);
int acquire_mag_pack(iso_ic* ic, const int32_t* bc_cols, int nb, int64_t n, iso::FastArgs& F);
}  // namespace

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
    *out = c;
    return ISO_OK;
}

void iso_ctx_destroy(iso_ctx* ctx) { delete ctx; }

int iso_table_create(iso_ctx* ctx, int ndim, const int64_t* shape, const double* grid, const double* const* axes,
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
    if (e == hipSuccess) e = hipMemcpy(t->d_grid, grid, bytes, hipMemcpyHostToDevice);
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

void iso_table_destroy(iso_table* t)
{
    if (!t) return;
    DeviceGuard guard(t->device);
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
            hipLaunchKernelGGL(k_interp3_wide, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), sh, as_stream(stream), W);
            HIP_TRY(hipGetLastError());
            return ISO_OK;
        }
    }
    const int lanes_per_sample = (k + 1) / 2, samples_per_wave = 64 / lanes_per_sample;
    const int64_t waves = (n + samples_per_wave - 1) / samples_per_wave;
    const dim3 g(grid_blocks(waves * 64)), b(BLOCK);
    const size_t shmem = (size_t)lds * sizeof(double);
    switch (t->ndim) {
    case 2: hipLaunchKernelGGL(k_interp<2>, g, b, shmem, as_stream(stream), A); break;
    case 3: hipLaunchKernelGGL(k_interp<3>, g, b, shmem, as_stream(stream), A); break;
    default: hipLaunchKernelGGL(k_interp<4>, g, b, shmem, as_stream(stream), A); break;
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
    // (cell indices travel as 32-bit integers through the cooperative gather)
    if (path_mode() == PATH_AUTO && model_grid->ax[2].uniform && model_grid->ncells < (int64_t(1) << 31) &&
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
    if (nb >= 1 && nb <= 12 && mags && ic->d_hotq && path_mode() == PATH_AUTO) {
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
    hipLaunchKernelGGL(k_pack_bc, dim3(grid_blocks(ncells * nb)), dim3(BLOCK), 0, 0, P);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e;
}

bool fast_eligible(const iso_ic* ic, const iso_model_desc* desc)
{
    // asteroseismic terms are only on the corner-packed form of the fast path
    return path_mode() != PATH_GENERIC && desc->n_bands >= 0 && desc->n_bands <= 12 &&
           (!desc->has_numax || (path_mode() == PATH_AUTO && ic->d_hotq)) && ic->model->ax[2].uniform;
}

// staged axes (+ reciprocal spacings) and, when the interpolator has a corner-packed model table,
// the corner-packed BC; fills F (without m / pars / outputs).  *ok = false if not representable.
hipError_t build_fast(const iso_ic* ic, int nb, const double* d_bc_hot, double** d_axes_blob, double** d_bcq,
                      FastArgs& F, bool* ok)
{
    *ok = false;
    std::vector<double> blob;
    FastAxis fa[6];
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
    if ((int)blob.size() > FAST_MAX_BLOB) return hipSuccess;
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
    F.e_a0 = ic->model->ax[2].a0;
    F.e_step = ic->model->ax[2].step;
    F.e_inv = 1.0 / F.e_step;
    F.e_n = ic->model->ax[2].n;
    F.axes_blob = *d_axes_blob;
    F.axes_len = (int)blob.size();
    F.hot = ic->d_hot;
    F.hotq = ic->d_hotq;
    F.astq = nullptr;          // set by iso_model_create for asteroseismic models
    F.s0 = ic->g3.s0; F.s1 = ic->g3.s1;
    F.bc = d_bc_hot;
    F.bcq = *d_bcq;
    F.bs2 = ic->bc->shape[3];
    F.bs1 = ic->bc->shape[2] * ic->bc->shape[3];
    F.bs0 = ic->bc->shape[1] * ic->bc->shape[2] * ic->bc->shape[3];
    *ok = true;
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
    m->stage_rows = 0;

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
    if (e != hipSuccess) {
        std::string msg = std::string("iso_model_create: ") + hipGetErrorString(e);
        iso_model_destroy(m);
        return fail(e == hipErrorOutOfMemory ? ISO_ERR_NOMEM : ISO_ERR_HIP, msg);
    }
    *out = m;
    return ISO_OK;
}

void iso_model_destroy(iso_model* m)
{
    if (!m) return;
    DeviceGuard guard(m->device);
    if (m->d_model) (void)hipFree(m->d_model);
    if (m->d_bc_hot) (void)hipFree(m->d_bc_hot);
    if (m->d_bcq) (void)hipFree(m->d_bcq);
    if (m->d_axes_blob) (void)hipFree(m->d_axes_blob);
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    delete m;
}

int iso_model_n_params(const iso_model* m) { return m ? m->desc.n_stars + 4 : ISO_ERR_INVALID; }

}  // extern "C"

namespace {

template <int KIND, int NS, bool PARTS>
void launch_lnpost_nb(int nb, dim3 g, dim3 b, size_t shmem, hipStream_t s, const PostArgs& A)
{
    switch (nb) {
    case 1: hipLaunchKernelGGL((k_lnpost<KIND, NS, 1, PARTS>), g, b, shmem, s, A); break;
    case 2: hipLaunchKernelGGL((k_lnpost<KIND, NS, 2, PARTS>), g, b, shmem, s, A); break;
    case 3: hipLaunchKernelGGL((k_lnpost<KIND, NS, 3, PARTS>), g, b, shmem, s, A); break;
    case 4: hipLaunchKernelGGL((k_lnpost<KIND, NS, 4, PARTS>), g, b, shmem, s, A); break;
    case 5: hipLaunchKernelGGL((k_lnpost<KIND, NS, 5, PARTS>), g, b, shmem, s, A); break;
    case 6: hipLaunchKernelGGL((k_lnpost<KIND, NS, 6, PARTS>), g, b, shmem, s, A); break;
    case 7: hipLaunchKernelGGL((k_lnpost<KIND, NS, 7, PARTS>), g, b, shmem, s, A); break;
    case 8: hipLaunchKernelGGL((k_lnpost<KIND, NS, 8, PARTS>), g, b, shmem, s, A); break;
    default: hipLaunchKernelGGL((k_lnpost<KIND, NS, 0, PARTS>), g, b, shmem, s, A); break;
    }
}

template <bool PARTS>
void launch_lnpost(const iso_model* m, dim3 g, dim3 b, size_t shmem, hipStream_t s, const PostArgs& A)
{
    const int nb = m->desc.n_bands;
    if (m->ic->kind == ISO_KIND_TRACK) {
        launch_lnpost_nb<ISO_KIND_TRACK, 1, PARTS>(nb, g, b, shmem, s, A);
    } else {
        switch (m->desc.n_stars) {
        case 1: launch_lnpost_nb<ISO_KIND_ISO, 1, PARTS>(nb, g, b, shmem, s, A); break;
        case 2: launch_lnpost_nb<ISO_KIND_ISO, 2, PARTS>(nb, g, b, shmem, s, A); break;
        default: launch_lnpost_nb<ISO_KIND_ISO, 3, PARTS>(nb, g, b, shmem, s, A); break;
        }
    }
}

int enqueue_lnpost(iso_model* m, const double* pars, int64_t stride_n, int64_t stride_p, int64_t n,
                   double* lnpost_out, double* lnprior_out, double* lnlike_out, hipStream_t s)
{
    if (m->fast_ok) {
        FastArgs F = m->fast;
        F.pars = pars;
        F.stride_n = stride_n;
        F.stride_p = stride_p;
        F.n = n;
        F.lnpost = lnpost_out;
        F.lnprior = lnprior_out;
        // lnprior alone still needs the likelihood flag off; lnlike requested -> evaluate everywhere
        F.lnlike = lnlike_out;
        const bool packed = F.hotq != nullptr && (F.bcq != nullptr || m->desc.n_bands == 0);
        if (launch_lnpost_fast(m->ic->kind, m->desc.n_stars, m->desc.n_bands, packed, false, F, s)) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(ISO_ERR_HIP, std::string("iso_lnpost (fast) launch: ") + hipGetErrorString(e));
            return ISO_OK;
        }
    }
    PostArgs A;
    A.g3 = m->ic->g3;
    A.g4 = m->g4;
    A.m = m->d_model;
    A.pars = pars;
    A.stride_n = stride_n;
    A.stride_p = stride_p;
    A.n = n;
    A.lnpost = lnpost_out;
    A.lnprior = lnprior_out;
    A.lnlike = lnlike_out;
    const dim3 g(grid_blocks(n)), b(BLOCK);
    const size_t shmem = (size_t)m->ic->lds_doubles * sizeof(double);
    if (lnprior_out || lnlike_out) launch_lnpost<true>(m, g, b, shmem, s, A);
    else launch_lnpost<false>(m, g, b, shmem, s, A);
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

int iso_lnpost_host(iso_model* m, const double* pars, int64_t n, double* lnpost_out, double* lnprior_out,
                    double* lnlike_out)
{
    if (!m || (!pars && n > 0)) return fail(ISO_ERR_INVALID, "iso_lnpost_host: NULL argument");
    if (n < 0) return fail(ISO_ERR_INVALID, "iso_lnpost_host: n < 0");
    if (!lnpost_out && !lnprior_out && !lnlike_out) return fail(ISO_ERR_INVALID, "iso_lnpost_host: no output requested");
    if (n == 0) return ISO_OK;
    DeviceGuard guard(m->device);
    const int np_ = m->desc.n_stars + 4;
    constexpr int64_t CAP = 8192;
    if (!m->h_stage) {
        // pinned + mapped: the kernel reads the parameters and writes the results straight through
        // PCIe — one launch + one synchronise per call, no separate copies
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_stage), sizeof(double) * CAP * (np_ + 3), hipHostMallocMapped));
        m->stage_rows = CAP;
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
    for (int64_t done = 0; done < n; done += CAP) {
        const int64_t c = std::min<int64_t>(CAP, n - done);
        std::memcpy(h_pars, pars + done * np_, sizeof(double) * c * np_);
        const int rc = enqueue_lnpost(m, d_pars, np_, 1, c, lnpost_out ? d_post : nullptr, lnprior_out ? d_prior : nullptr,
                                      lnlike_out ? d_like : nullptr, nullptr);
        if (rc != ISO_OK) return rc;
        HIP_TRY(hipStreamSynchronize(nullptr));
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
    hipLaunchKernelGGL(k_interp_eep, dim3(grid_blocks(n)), dim3(BLOCK), (size_t)lds * sizeof(double), as_stream(stream), A);
    HIP_TRY(hipGetLastError());
    return ISO_OK;
}

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
    if (!fast_eligible(ic, &d0) || !ic->d_hotq || d0.n_bands < 1)
        return fail(ISO_ERR_INVALID, "iso_catalog_create: needs 1-12 bands, a uniform EEP axis and the corner-packed "
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
    c->d_bc_hot = c->d_bcq = c->d_axes_blob = nullptr;
    std::vector<DevModel> H((size_t)n_models);
    for (int64_t k = 0; k < n_models; ++k) fill_dev_model(&descs[k], ic->kind, H[(size_t)k]);
    hipError_t e = hipMalloc(&c->d_models, sizeof(DevModel) * (size_t)n_models);
    if (e == hipSuccess) e = hipMemcpy(c->d_models, H.data(), sizeof(DevModel) * (size_t)n_models, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pack_bands(ic, d0.bc_cols, d0.n_bands, &c->d_bc_hot);
    bool ok = false;
    if (e == hipSuccess) e = build_fast(ic, d0.n_bands, c->d_bc_hot, &c->d_axes_blob, &c->d_bcq, c->fast, &ok);
    if (e == hipSuccess && (!ok || !c->d_bcq)) {
        iso_catalog_destroy(c);
        return fail(ISO_ERR_INVALID, "iso_catalog_create: tables not representable on the fast path");
    }
    if (e != hipSuccess) {
        std::string msg = std::string("iso_catalog_create: ") + hipGetErrorString(e);
        iso_catalog_destroy(c);
        return fail(e == hipErrorOutOfMemory ? ISO_ERR_NOMEM : ISO_ERR_HIP, msg);
    }
    c->fast.m = c->d_models;
    c->packed = true;
    *out = c;
    return ISO_OK;
}

void iso_catalog_destroy(iso_catalog* c)
{
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->d_models) (void)hipFree(c->d_models);
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
    if (!launch_lnpost_fast(c->ic->kind, c->n_stars, c->n_bands, true, true, F, as_stream(stream)))
        return fail(ISO_ERR_INVALID, "iso_catalog_lnpost: no kernel specialisation");
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
        gauss_consts(d->plx_unc[s], H->plx_g0[s], u2);
        gauss_consts(d->av_unc[s], H->av_g0[s], u2);
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
        gauss_consts(d->terms[t].unc, H->term_g0[t], u2);
    }
    for (int k = 0; k < d->n_spec; ++k) {
        H->spec[k] = d->spec[k];
        double u2;
        gauss_consts(d->spec[k].b, H->spec_g0[k], u2);
    }
    for (int k = 0; k < d->n_limits; ++k) H->limits[k] = d->limits[k];
    H->prior_mass = make_dev_prior(d->prior_mass);
    H->prior_age = make_dev_prior(d->prior_age);
    H->prior_feh = make_dev_prior(d->prior_feh);
    H->prior_distance = make_dev_prior(d->prior_distance);
    H->prior_AV = make_dev_prior(d->prior_AV);
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
    if (e == hipSuccess && path_mode() == PATH_AUTO && ic->d_hotq && d->n_bands >= 1 && d->n_bands <= 12 &&
        ic->model->ax[2].uniform) {
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

void iso_tree_model_destroy(iso_tree_model* m)
{
    if (!m) return;
    DeviceGuard guard(m->device);
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
    hipLaunchKernelGGL(k_lnpost_tree, dim3(grid_blocks(n)), dim3(BLOCK), (size_t)m->ic->lds_doubles * sizeof(double),
                       as_stream(stream), A);
    HIP_TRY(hipGetLastError());
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
    sp->fast = F;
    return ISO_OK;
}
}  // namespace

int iso_sampler_create_model(iso_model* m, int nwalkers, double a, uint64_t seed, iso_sampler** out)
{
    if (!m || !out) return fail(ISO_ERR_INVALID, "iso_sampler_create_model: NULL argument");
    if (nwalkers < 2 || (nwalkers & 1) || !(a > 1.0)) return fail(ISO_ERR_INVALID, "iso_sampler_create_model: need an even walker count and a > 1");
    if (!m->fast_ok || !m->fast.hotq || (!m->fast.bcq && m->desc.n_bands > 0) || m->fast.astq)
        return fail(ISO_ERR_INVALID, "iso_sampler_create_model: the model is not on the corner-packed fast path "
                                     "(needs <= 12 bands, uniform EEP axis, no asteroseismic terms, ISOCHRONES_AMD_PATH=auto)");
    iso_sampler* sp = new (std::nothrow) iso_sampler();
    if (!sp) return fail(ISO_ERR_NOMEM, "iso_sampler_create_model: out of host memory");
    sampler_common(sp, m->device, m->ic->kind, m->desc.n_stars, m->desc.n_bands, 1, m->fast, 0, nwalkers, a, seed);
    *out = sp;
    return ISO_OK;
}

int iso_sampler_create_catalog(iso_catalog* c, int nwalkers, double a, uint64_t seed, iso_sampler** out)
{
    if (!c || !out) return fail(ISO_ERR_INVALID, "iso_sampler_create_catalog: NULL argument");
    if (nwalkers < 2 || (nwalkers & 1) || !(a > 1.0)) return fail(ISO_ERR_INVALID, "iso_sampler_create_catalog: need an even walker count and a > 1");
    iso_sampler* sp = new (std::nothrow) iso_sampler();
    if (!sp) return fail(ISO_ERR_NOMEM, "iso_sampler_create_catalog: out of host memory");
    sampler_common(sp, c->device, c->ic->kind, c->n_stars, c->n_bands, c->n_models, c->fast, 1, nwalkers, a, seed);
    *out = sp;
    return ISO_OK;
}

void iso_sampler_destroy(iso_sampler* s) { delete s; }


int iso_sampler_run(iso_sampler* sp, double* pos, double* lnp, int nsteps, double* chain, double* chain_lnp,
                    int32_t* accepted, void* stream)
{
    if (!sp || !pos || !lnp) return fail(ISO_ERR_INVALID, "iso_sampler_run: NULL argument");
    if (nsteps < 0) return fail(ISO_ERR_INVALID, "iso_sampler_run: nsteps < 0");
    DeviceGuard guard(sp->device);
    hipStream_t s = as_stream(stream);
    const int64_t rows = sp->n_ensembles * sp->W;
    StretchArgs S;
    S.occupancy_query = nullptr;
    S.pos = pos;
    S.lnp = lnp;
    S.accepted = accepted;
    S.W = sp->W;
    S.multi = sp->multi;
    S.n_active = sp->n_ensembles * (sp->W / 2);
    S.a = sp->a;
    S.seed = sp->seed;
    // ISOCHRONES_AMD_SAMPLER = auto | persistent | stepwise.  The persistent kernel (one workgroup per
    // ensemble, all iterations in one launch) wins while the catalog is too small for a half-step launch
    // to fill the chip; both forms produce bit-identical chains.
    const char* env = getenv("ISOCHRONES_AMD_SAMPLER");
    const std::string mode = env ? env : "auto";
    if (mode != "auto" && mode != "persistent" && mode != "stepwise")
        return fail(ISO_ERR_INVALID, "ISOCHRONES_AMD_SAMPLER must be auto, persistent or stepwise");
    int group = 1;
    const size_t lds_bytes = stretch_persist_lds(sp->n_bands, sp->fast.axes_len, sp->W, sp->n_params, &group);
    const bool fits = lds_bytes <= 64 * 1024;
    if (mode == "persistent" && !fits)
        return fail(ISO_ERR_INVALID, "iso_sampler_run: ensemble too large for the persistent kernel's LDS");
    // auto: persistent while every workgroup of the launch is resident at once (occupancy of this kernel
    // instantiation as the runtime reports it); beyond that its workgroups would run in rounds and the
    // step-wise form has the better throughput
    int cus = 0, per_cu = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, sp->device));
    const int64_t blocks = (sp->n_ensembles + group - 1) / group;
    if (fits && mode == "auto") {
        S.nsteps = 1;
        S.occupancy_query = &per_cu;
        if (!launch_stretch(sp->kind, sp->n_stars, sp->n_bands, sp->fast, S, s)) per_cu = 0;
        S.occupancy_query = nullptr;
    }
    const int64_t resident = (int64_t)cus * per_cu;
    const bool persistent = nsteps > 0 && fits && (mode == "persistent" || (mode == "auto" && blocks <= resident));
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
    if (!ctx || !chain || !q || !out) return fail(ISO_ERR_INVALID, "iso_chain_quantiles: NULL argument");
    if (nsteps < 1 || n_ens < 1 || W < 1 || n_params < 1 || nq < 1 || nq > 8)
        return fail(ISO_ERR_INVALID, "iso_chain_quantiles: counts out of range");
    const int64_t m = nsteps * W;
    if (m > 8192) return fail(ISO_ERR_INVALID, "iso_chain_quantiles: more than 8192 samples per ensemble");
    if (n_ens * n_params > 0x7fffffff) return fail(ISO_ERR_INVALID, "iso_chain_quantiles: too many ensembles");
    QuantArgs A;
    A.chain = chain;
    A.nsteps = nsteps;
    A.n_ens = n_ens;
    A.W = W;
    A.D = n_params;
    A.nq = nq;
    A.P = 2;
    while (A.P < m) A.P <<= 1;
    for (int k = 0; k < 8; ++k) A.q[k] = 0.0;
    for (int k = 0; k < nq; ++k) {
        if (!(q[k] >= 0.0 && q[k] <= 1.0)) return fail(ISO_ERR_INVALID, "iso_chain_quantiles: quantile level outside [0, 1]");
        A.q[k] = q[k];
    }
    A.out = out;
    DeviceGuard guard(ctx->device);
    hipLaunchKernelGGL(k_chain_quantiles, dim3((unsigned)(n_ens * n_params)), dim3(BLOCK), (size_t)A.P * sizeof(double),
                       as_stream(stream), A);
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

}  // extern "C"
