// K4: interp_mag, column-parallel over bands
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

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
// One sample's share of one lane (mags.py:35-61): the four stellar columns at (p0, p1, p2) - NaN where the point is not on
// the table - and, where `has0`, the magnitudes of the bands in BC columns c0 and c1.  Shared by the batch kernel below and
// the resident service wave (k_service.h).
template <int KIND>
__device__ __forceinline__ void interp_mag_point(const MagArgs& A, const double* lds, double p0, double p1, double p2, double dist,
                                                 double AV, bool has0, int c0, int c1, double* __restrict__ star, double& m0, double& m1)
{
    double x0, x1, x2;
    to_axes<KIND>(p0, p1, p2, x0, x1, x2);
    star[0] = star[1] = star[2] = star[3] = d_nan();
    Cell3 c3;
    if (locate3(A.g3, lds, x0, x1, x2, c3)) gather3<4>(A.g3, c3, star);
    if (has0) {
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
        m0 = star[3] + dm - b0;
        m1 = star[3] + dm - b1;
    }
}

template <int KIND>
__global__ __launch_bounds__(BLOCK, 2) void k_interp_mag(const MagArgs A)
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
        double star[4], m0 = d_nan(), m1 = d_nan();
        interp_mag_point<KIND>(A, lds, p0, p1, p2, dist, AV, A.mags && has0, c0, c1, star, m0, m1);
        if (sub == 0) {
            if (A.Teff) A.Teff[i] = star[0];
            if (A.logg) A.logg[i] = star[1];
            if (A.feh) A.feh[i] = star[2];
        }
        if (A.mags && has0) {
            double* o = A.mags + i * A.nb + 2 * sub;
            o[0] = m0;
            if (has1) o[1] = m1;
        }
    }
}
