// K3: N-D interpolation of selected columns (column-parallel kernel and the wide-pack kernel for 3-D tables)
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

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
// One sample's share of one lane: columns c0 and c1 of the table at x (the 2^ND corners in the reference's order, bit
// (ND-1-d) of j offsets axis d; interp.py:264-291, 309-336).  false = a NaN coordinate or a coordinate outside its axis
// (the caller writes NaN).  Shared by the batch kernel below and the resident service wave (k_service.h): the same
// instructions on the same inputs either way.
template <int ND>
__device__ __forceinline__ bool interp_point(const InterpArgs& A, const double* lds, const double* x, int c0, int c1, double& v0,
                                             double& v1)
{
    bool bad = false;
#pragma unroll
    for (int d = 0; d < ND; ++d) bad |= (x[d] != x[d]);
    if (!bad) {
#pragma unroll
        for (int d = 0; d < ND; ++d) bad |= out_of_axis(A.ax[d], lds, x[d]);
    }
    if (bad) return false;
    double t[ND];
    int64_t base = 0;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        int idx;
        bracket(A.ax[d], lds, x[d], idx, t[d]);
        base += (int64_t)idx * A.stride[d];
    }
    v0 = 0.0;
    v1 = 0.0;
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
    return true;
}

template <int ND>
__global__ __launch_bounds__(BLOCK, 2) void k_interp(const InterpArgs A)
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
#pragma unroll
        for (int d = 0; d < ND; ++d) x[d] = A.x[d][i];
        double v0, v1;
        const bool ok = interp_point<ND>(A, lds, x, c0, c1, v0, v1);
        double* o = A.out + i * A.k + 2 * sub;
        o[0] = ok ? v0 : d_nan();
        if (two) o[1] = ok ? v1 : d_nan();
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
    int groups;              // groups of 64 samples per wave
    int32_t icols[ISO_MAX_COLS];
    double* out;
};

struct PackWideArgs {
    const double* grid;      // [n0][n1][n2][ncol]
    double* out;
    int64_t n0, n1, n2;
    int ncol;
};

__global__ __launch_bounds__(BLOCK, 2) void k_pack_wide(const PackWideArgs P)
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
constexpr int WIDE_UNROLL = 8;    // passes whose loads are in flight together (U of the kernel below)

// U = passes of 16 (sample, column) units whose loads are in flight together.  8 for k >= 2 columns.  U = 4 is the
// one-column form: a wave's 64 samples are 64 units = 4 passes, the other four of the 8-pass form only repeated the last
// unit's load (and held 126 registers = 4 waves per SIMD; this form fits 6).
// A.groups = groups of 64 samples one wave serves one after the other.  With one group per wave a workgroup lives for
// 256 samples and pays for them the staging of the three axes (15 KB from L2 for the MIST track table) and a barrier:
// on a one-column call (96 algorithmic bytes per sample) that fixed part is what kept the kernel at 4.8 of the 6.3 TB/s
// the fabric delivers to a gather (profiles/r03: 34.8 us for 10^6 samples).  The coordinates of the next group are
// fetched before the current group's gathers are waited for.
template <int U>
__global__ __launch_bounds__(BLOCK, U == 4 ? 6 : 4) void k_interp3_wide(const WideArgs A)
{
    extern __shared__ double lds[];
    stage_axes<3>(A.ax, lds);
    int32_t* lcols = reinterpret_cast<int32_t*>(lds + A.lds_axes);
    for (int j = threadIdx.x; j < A.k; j += BLOCK) lcols[j] = A.icols[j];
    __syncthreads();
    double* slots = lds + A.lds_axes + (ISO_MAX_COLS / 2) + (threadIdx.x >> 6) * 64 * WIDE_SLOT;
    const int lane = threadIdx.x & 63;
    const int groups = A.groups;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t first0 = wave * groups * 64;                  // the wave's first sample
    const int j = lane & 3, grp = lane >> 2;
    const int k = A.k;
    double xn[3] = {0.0, 0.0, 0.0};
    if (first0 + lane < A.n) {
#pragma unroll
        for (int d = 0; d < 3; ++d) xn[d] = A.x[d][first0 + lane];
    }
    for (int g = 0; g < groups; ++g) {
        const int64_t first = first0 + (int64_t)g * 64;
        if (first >= A.n) break;                                // wave-uniform
        const int64_t i = first + lane;
        {
            bool bad = i >= A.n;
            double x[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                x[d] = xn[d];
                bad |= (x[d] != x[d]);
            }
            if (g + 1 < groups && i + 64 < A.n) {               // the next group's coordinates are on their way meanwhile
#pragma unroll
                for (int d = 0; d < 3; ++d) xn[d] = A.x[d][i + 64];
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
        const int here = (int)min((int64_t)64, A.n - first);       // samples of this group
        const int units = here * k;
        double* __restrict__ out = A.out + first * k;
        for (int u0 = 0; u0 < units; u0 += 16 * U) {
            double2 v[U];
            double wx[U], wy[U];
            bool bad[U];
#pragma unroll
            for (int r = 0; r < U; ++r) {
                const int u = min(u0 + 16 * r + grp, units - 1);
                const int s = (int)(((uint64_t)(uint32_t)u * A.kinv) >> 32);      // u / k
                const int c = u - s * k;
                const double* rq = slots + s * WIDE_SLOT;
                const long long cell = __double_as_longlong(rq[0]);
                const double t0 = rq[1], t1 = rq[2], t2 = rq[3];
                bad[r] = cell < 0;
                const int64_t cc = bad[r] ? 0 : cell;
                v[r] = *reinterpret_cast<const double2*>(A.wide + ((cc * A.ncol + lcols[c]) << 3) + 2 * j);
                const double gw = ((j & 2) ? t0 : (1 - t0)) * ((j & 1) ? t1 : (1 - t1));
                wx[r] = gw * (1 - t2);
                wy[r] = gw * t2;
            }
#pragma unroll
            for (int r = 0; r < U; ++r) {
                double part = v[r].x * wx[r] + v[r].y * wy[r];
                part += wide_dpp(part, 0);
                part += wide_dpp(part, 1);
                const int u = u0 + 16 * r + grp;
                if (j == 0 && u < units) out[u] = bad[r] ? d_nan() : part;
            }
        }
        __builtin_amdgcn_wave_barrier();                        // the slots are rewritten by the next group
    }
}
