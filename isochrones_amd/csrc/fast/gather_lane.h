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

// six columns (Teff, logg, feh, Mbol, prior value, prior derivative) of one star
template <bool PACKED>
__device__ __forceinline__ void gather_star(const FastArgs& A, int i0, int i1, int i2, const W3& w,
                                            double* __restrict__ v)
{
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = 0.0;
    const int64_t cell = (int64_t)i0 * A.s0 + (int64_t)i1 * A.s1 + i2;
    if (PACKED) {
        const double2* __restrict__ p = reinterpret_cast<const double2*>(A.hotq + cell * PACK_ENTRY);
        double2 u[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) u[k] = p[k];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double ww = w3(w, j);
            v[0] += u[3 * j].x * ww;
            v[1] += u[3 * j].y * ww;
            v[2] += u[3 * j + 1].x * ww;
            v[3] += u[3 * j + 1].y * ww;
            v[4] += u[3 * j + 2].x * ww;
            v[5] += u[3 * j + 2].y * ww;
        }
    } else {
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

template <int NB, bool PACKED>
__device__ __forceinline__ void gather_bc(const FastArgs& A, int i0, int i1, int i2, int i3, const W4& w,
                                          double* __restrict__ v)
{
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = 0.0;
    const int64_t cell = (int64_t)i0 * A.bs0 + (int64_t)i1 * A.bs1 + (int64_t)i2 * A.bs2 + i3;
    if (PACKED) {
        const double* __restrict__ p = A.bcq + cell * (16 * NB);
        if ((NB & 1) == 0) {
            const double2* __restrict__ p2 = reinterpret_cast<const double2*>(p);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double ww = w4(w, j);
#pragma unroll
                for (int b = 0; b < NB; b += 2) {
                    const double2 u = p2[(j * NB + b) >> 1];
                    v[b] += u.x * ww;
                    v[b + 1] += u.y * ww;
                }
            }
        } else if (NB == 1) {
            const double2* __restrict__ p2 = reinterpret_cast<const double2*>(p);
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const double2 u = p2[j >> 1];
                v[0] += u.x * w4(w, j);
                v[0] += u.y * w4(w, j + 1);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double ww = w4(w, j);
#pragma unroll
                for (int b = 0; b < NB; ++b) v[b] += p[j * NB + b] * ww;
            }
        }
    } else {
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
}
