// posterior summaries of a stored chain (LDS bitonic sort)
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

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
            // every thread takes whole compare-exchange pairs: pair p -> lower index i (a zero inserted at bit
            // log2 j of p), partner i | j
            for (int p = threadIdx.x; p < (A.P >> 1); p += BLOCK) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int x = i | j;
                const double a = lds[i], b = lds[x];
                const bool asc = (i & k) == 0;
                if ((a > b) == asc) {
                    lds[i] = b;
                    lds[x] = a;
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
