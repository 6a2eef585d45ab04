// cell location + corner gathers on the compact hot table / BC table (generic kernels)
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

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

// The same sums on the corner-packed table (k_pack_corners, ndim 3, P column pairs): the 16-byte piece holding the
// column pair pr of corner j sits at piece index 4 * ((j >> 2) * P + pr) + (j & 3) of the cell.  One lane reads its
// cell's 384 contiguous bytes (three whole 128-B lines) instead of eight scattered 64-B rows.
__device__ __forceinline__ void gather3q(const Grid3V& G, const Cell3& c, double* __restrict__ v)
{
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = 0.0;
    const double2* __restrict__ p = reinterpret_cast<const double2*>(G.hotq + c.base * PACK_ENTRY);
    double2 u[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) u[k] = p[k];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double w = 1.0;
        w *= ((j >> 2) & 1) ? c.t0 : (1 - c.t0);
        w *= ((j >> 1) & 1) ? c.t1 : (1 - c.t1);
        w *= (j & 1) ? c.t2 : (1 - c.t2);
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            const double2 x = u[4 * ((j >> 2) * 3 + pr) + (j & 3)];
            v[2 * pr] += x.x * w;
            v[2 * pr + 1] += x.y * w;
        }
    }
}

// (nu_max, delta_nu) of a located cell: corner-packed pair table when there is one, else hot columns 6, 7
__device__ __forceinline__ void gather3_astero(const Grid3V& G, const Cell3& c, double* __restrict__ v)
{
    if (G.astq) {
        v[0] = v[1] = 0.0;
        const double2* __restrict__ p = reinterpret_cast<const double2*>(G.astq + c.base * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double w = 1.0;
            w *= ((j >> 2) & 1) ? c.t0 : (1 - c.t0);
            w *= ((j >> 1) & 1) ? c.t1 : (1 - c.t1);
            w *= (j & 1) ? c.t2 : (1 - c.t2);
            const double2 x = p[j];
            v[0] += x.x * w;
            v[1] += x.y * w;
        }
    } else {
        double a[8];
        gather3<8>(G, c, a);
        v[0] = a[6];
        v[1] = a[7];
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

// The BC sums on the corner-packed table (k_pack_corners, ndim 4, nb bands): the 16-byte piece with the axis-3 pair
// (o3 = 0, 1) of corner (o0, o1, o2) and band b sits at piece index (o0 * nb + b) * 4 + (o1 * 2 + o2) of the cell;
// a cell is 128 * nb contiguous bytes.
__device__ __forceinline__ double gather4q_col(const Grid4V& G, const Cell4& c, int col)
{
    const double2* __restrict__ p = reinterpret_cast<const double2*>(G.tabq + c.base * (16 * (int64_t)G.ncol));
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const double2 x = p[((j >> 3) * G.ncol + col) * 4 + ((j >> 1) & 3)];
        v += x.x * weight4(c, j);
        v += x.y * weight4(c, j + 1);
    }
    return v;
}

template <int NB>
__device__ __forceinline__ void gather4q_packed(const Grid4V& G, const Cell4& c, double* __restrict__ v)
{
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = 0.0;
    const double2* __restrict__ p = reinterpret_cast<const double2*>(G.tabq + c.base * (16 * NB));
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const double w0 = weight4(c, j), w1 = weight4(c, j + 1);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double2 x = p[((j >> 3) * NB + b) * 4 + ((j >> 1) & 3)];
            v[b] += x.x * w0;
            v[b] += x.y * w1;
        }
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
