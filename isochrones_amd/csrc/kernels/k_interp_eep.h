// (age, feh, mass) -> EEP on ragged per-track age arrays
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

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

// The reference's searchsorted (isochrones/interp.py:10-35) step for step: the index of the first element > x when no
// element equals x; on an exact hit the index of the equal element its bisection lands on - which, inside a run of
// repeated ages, is not the first of the run, so the probe sequence itself has to be the reference's.
__device__ __forceinline__ int64_t ref_searchsorted(const double* __restrict__ arr, double x, int64_t N)
{
    int64_t L = 0, R = N - 1;
    while (L <= R) {
        const int64_t m = (L + R) >> 1;
        const double xm = arr[m];
        if (xm < x) L = m + 1;
        else if (xm > x) R = m - 1;
        else if (xm == x) return m;
        else break;                         // NaN inside the track's length: the reference would spin; tables have none
    }
    return L;
}

// EEP of one star (interp.py:488-558) in three steps - the cell on the (feh, mass) axes, the age search on each of the four
// neighbouring tracks, the blend - shared by the batch kernel below (one lane does all of it) and the resident service wave
// (k_service.h: four lanes search one track each).
struct EepCell {
    bool ok;
    int64_t ind[4];
    double d0, d1;
};

__device__ __forceinline__ EepCell interp_eep_cell(const EepArgs& A, const double* lds, double x, double x0, double x1)
{
    EepCell c;
    c.ok = !(x != x || x0 != x0 || x1 != x1) && !out_of_axis(A.ax[0], lds, x0) && !out_of_axis(A.ax[1], lds, x1);
    c.d0 = c.d1 = 0.0;
    c.ind[0] = c.ind[1] = c.ind[2] = c.ind[3] = 0;
    if (c.ok) {
        int i0, i1;
        bracket(A.ax[0], lds, x0, i0, c.d0);
        bracket(A.ax[1], lds, x1, i1, c.d1);
        c.ind[0] = (int64_t)i0 * A.n1 + i1;
        c.ind[1] = (int64_t)i0 * A.n1 + i1 + 1;
        c.ind[2] = (int64_t)(i0 + 1) * A.n1 + i1;
        c.ind[3] = (int64_t)(i0 + 1) * A.n1 + i1 + 1;
    }
    return c;
}

__device__ __forceinline__ void interp_eep_track(const EepArgs& A, int64_t track, double x, int64_t& ie, int64_t& len)
{
    len = A.lengths[track];
    ie = ref_searchsorted(A.ages + track * A.n_eep, x, len);
}

__device__ __forceinline__ double interp_eep_blend(const EepArgs& A, const EepCell& c, const int64_t* ie, const int64_t* len)
{
    if (!c.ok) return d_nan();
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) bad |= ie[k] > A.n_eep - 1;
    if (bad) return d_nan();
    double e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = A.eep0 + (double)ie[k];
    if (ie[0] >= len[0]) e[0] = e[1];      // sequential substitution, as the reference
    if (ie[1] >= len[1]) e[1] = e[0];
    if (ie[2] >= len[2]) e[2] = e[3];
    if (ie[3] >= len[3]) e[3] = e[2];
    const double e_0 = (1 - c.d1) * e[0] + c.d1 * e[1];
    const double e_1 = (1 - c.d1) * e[2] + c.d1 * e[3];
    return (1 - c.d0) * e_0 + c.d0 * e_1;
}

__device__ __forceinline__ double interp_eep_point(const EepArgs& A, const double* lds, double x, double x0, double x1)
{
    const EepCell c = interp_eep_cell(A, lds, x, x0, x1);
    int64_t ie[4] = {0, 0, 0, 0}, len[4] = {0, 0, 0, 0};
    if (c.ok) {
#pragma unroll
        for (int k = 0; k < 4; ++k) interp_eep_track(A, c.ind[k], x, ie[k], len[k]);
    }
    return interp_eep_blend(A, c, ie, len);
}

__global__ __launch_bounds__(BLOCK, 2) void k_interp_eep(const EepArgs A)
{
    extern __shared__ double lds[];
    stage_axes<2>(A.ax, lds);
    __syncthreads();
    const int64_t stride_grid = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < A.n; i += stride_grid)
        A.out[i] = interp_eep_point(A, lds, A.x[i], A.x0[i], A.x1[i]);
}
