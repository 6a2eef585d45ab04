// lane-per-sample gathers (compact tables; the PACKED = false kernels)
// (part of iso_fast_kernel.h: included inside namespace iso::fastk)
#pragma once

// ---- gathers ------------------------------------------------------------------------------
struct W3 {
    double t0, t1, t2;
};

__device__ __forceinline__ double w3(const W3& w, int j)
{
    double r = 1.0;
    r *= ((j >> 2) & 1) ? w.t0 : (1 - w.t0);
    r *= ((j >> 1) & 1) ? w.t1 : (1 - w.t1);
    r *= (j & 1) ? w.t2 : (1 - w.t2);
    return r;
}

// six columns (Teff, logg, feh, Mbol, prior value, prior derivative) of one star from the compact hot table
// (the corner-packed tables are read by the wave-cooperative gathers of coop_gather.h only)
__device__ __forceinline__ void gather_star(const FastArgs& A, int i0, int i1, int i2, const W3& w,
                                            double* __restrict__ v)
{
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = 0.0;
    const int64_t cell = (int64_t)i0 * A.s0 + (int64_t)i1 * A.s1 + i2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t c = cell + (((j >> 2) & 1) ? A.s0 : 0) + (((j >> 1) & 1) ? A.s1 : 0) + (j & 1);
        const double2* __restrict__ p = reinterpret_cast<const double2*>(A.hot + c * HOT_COLS);
        const double2 u0 = p[0], u1 = p[1], u2 = p[2];
        const double ww = w3(w, j);
        v[0] += u0.x * ww;
        v[1] += u0.y * ww;
        v[2] += u1.x * ww;
        v[3] += u1.y * ww;
        v[4] += u2.x * ww;
        v[5] += u2.y * ww;
    }
}

struct W4 {
    double t0, t1, t2, t3;
};

__device__ __forceinline__ double w4(const W4& w, int j)
{
    double r = 1.0;
    r *= ((j >> 3) & 1) ? w.t0 : (1 - w.t0);
    r *= ((j >> 2) & 1) ? w.t1 : (1 - w.t1);
    r *= ((j >> 1) & 1) ? w.t2 : (1 - w.t2);
    r *= (j & 1) ? w.t3 : (1 - w.t3);
    return r;
}

template <int NB>
__device__ __forceinline__ void gather_bc(const FastArgs& A, int i0, int i1, int i2, int i3, const W4& w,
                                          double* __restrict__ v)
{
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = 0.0;
    const int64_t cell = (int64_t)i0 * A.bs0 + (int64_t)i1 * A.bs1 + (int64_t)i2 * A.bs2 + i3;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t c = cell + (((j >> 3) & 1) ? A.bs0 : 0) + (((j >> 2) & 1) ? A.bs1 : 0) +
                          (((j >> 1) & 1) ? A.bs2 : 0) + (j & 1);
        const double* __restrict__ p = A.bc + c * NB;
        const double ww = w4(w, j);
#pragma unroll
        for (int b = 0; b < NB; ++b) v[b] += p[b] * ww;
    }
}
