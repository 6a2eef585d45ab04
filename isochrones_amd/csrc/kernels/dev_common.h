// device helpers shared by the generic kernels: NaN/inf constants, LDS staging of axes, bracket search
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

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
