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
