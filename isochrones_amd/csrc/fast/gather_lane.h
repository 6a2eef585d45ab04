// interpolation weights and the lane-per-sample gathers on the corner-packed tables (the single-model sampler)
// (part of iso_fast_kernel.h: included inside namespace iso::fastk)
#pragma once

// ---- interpolation weights of a sample on the model table (3 axes) and the BC table (4 axes) ----
struct W3 {
    double t0, t1, t2;
};

struct W4 {
    double t0, t1, t2, t3;
};

// One lane's share of a sample: the weighted corners it loaded, in a fixed order of operations (explicit fused
// multiply-adds: the cooperative gathers and the lane-per-sample gathers of gather_lane.h then produce the same bits,
// whatever the compiler would have contracted).
__device__ __forceinline__ double corner_pair(double lo, double wlo, double hi, double whi)
{
    return fma(hi, whi, lo * wlo);
}
__device__ __forceinline__ double corner_quad(double2 x0, double wa0, double wb0, double2 x1, double wa1, double wb1)
{
    return fma(x1.y, wb1, fma(x1.x, wa1, fma(x0.y, wb0, x0.x * wa0)));
}

// ---- lane-per-sample gathers on the corner-packed tables -----------------------------------------------------------
// The wave-cooperative gathers (coop_gather.h) exist for bandwidth: four lanes share a sample so that a wave instruction
// covers whole cache lines.  A single star's fit has one workgroup on the chip and waits for nothing but its own
// dependent instructions; there the protocol (request slots, two wave barriers, DPP sums, a round per 16 samples) is
// the cost, and a lane that fetches its own cell - 24 (model) / 8 x bands (BC) independent 16-B loads of three
// consecutive lines - is done sooner.  Same pieces, same weights, same order of operations as the four lanes of a quad
// (per-lane share, then (p0 + p1) + (p2 + p3)): bit-identical results.
__device__ __forceinline__ double lane_quad_total(const double* p) { return (p[0] + p[1]) + (p[2] + p[3]); }

// The cooperative gathers receive the interpolation weights t through LDS, i.e. rounded to double, and form 1 - t from
// that.  A lane that keeps t in registers would have the multiplication that produced it, t = (x - a_i) * (1 / spacing),
// contracted into the subtraction (1 - t as one fused multiply-add of the unrounded product): a few results per hundred
// then differ in the last bit.  The weights pass through an empty asm statement here, which makes them the rounded values.
__device__ __forceinline__ double rounded(double t)
{
    asm("" : "+v"(t));
    return t;
}

__device__ __forceinline__ void lane_star(const FastArgs& A, bool need, uint32_t cell, const W3& w, double* __restrict__ v)
{
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = f_nan();
    if (!need) return;
    const double2* __restrict__ pc = reinterpret_cast<const double2*>(A.hotq + (size_t)cell * PACK_ENTRY);
    double2 u[4][6];
#pragma unroll
    for (int e = 0; e < 6; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j][e] = pc[4 * e + j];
    double part[6][4];
    const double t0 = rounded(w.t0), t1 = rounded(w.t1), t2 = rounded(w.t2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double g = ((j & 2) ? t1 : (1 - t1)) * ((j & 1) ? t2 : (1 - t2));
        const double wlo = (1 - t0) * g, whi = t0 * g;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            part[2 * q][j] = corner_pair(u[j][q].x, wlo, u[j][3 + q].x, whi);
            part[2 * q + 1][j] = corner_pair(u[j][q].y, wlo, u[j][3 + q].y, whi);
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = lane_quad_total(part[q]);
}

__device__ __forceinline__ void lane_pair(const double* __restrict__ tab, bool need, uint32_t cell, const W3& w,
                                          double* __restrict__ v)
{
    v[0] = v[1] = f_nan();
    if (!need) return;
    const double2* __restrict__ pc = reinterpret_cast<const double2*>(tab + (size_t)cell * 16);
    double pa[4], pb[4];
    const double t0 = rounded(w.t0), t1 = rounded(w.t1), t2 = rounded(w.t2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double2 lo = pc[j], hi = pc[4 + j];
        const double g = ((j & 2) ? t1 : (1 - t1)) * ((j & 1) ? t2 : (1 - t2));
        const double wlo = (1 - t0) * g, whi = t0 * g;
        pa[j] = corner_pair(lo.x, wlo, hi.x, whi);
        pb[j] = corner_pair(lo.y, wlo, hi.y, whi);
    }
    v[0] = lane_quad_total(pa);
    v[1] = lane_quad_total(pb);
}

template <int NB>
__device__ __forceinline__ void lane_bc(const FastArgs& A, bool need, uint32_t cell, const W4& w, double* __restrict__ v)
{
    // Every lane gathers, needed or not (a lane without a usable bracket reads cell 0, which always exists, and gets NaN at
    // the end): no divergent region around the 8 NB loads, i.e. one `if` less whose join block the register allocator can
    // fill with copies.  (Round 3's wrong (isochrone, 3 stars, 9 bands) sampler kernel was such a copy placed ahead of a join
    // block's exec restore - in the flux sum's |m| > 700 region, not here: profiles/r04/miscompile_hunt.md; csrc/isa_check.py
    // refuses a build that has the shape anywhere.)
    cell = need ? cell : 0u;
    const double2* __restrict__ pc = reinterpret_cast<const double2*>(A.bcq + (size_t)cell * (16 * NB));
    double part[NB][4];
    const double t0 = rounded(w.t0), t1 = rounded(w.t1), t2 = rounded(w.t2), t3 = rounded(w.t3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double g = ((j & 2) ? t1 : (1 - t1)) * ((j & 1) ? t2 : (1 - t2));
        const double wa0 = (1 - t0) * g * (1 - t3), wb0 = (1 - t0) * g * t3;
        const double wa1 = t0 * g * (1 - t3), wb1 = t0 * g * t3;
#pragma unroll
        for (int b = 0; b < NB; ++b) part[b][j] = corner_quad(pc[4 * b + j], wa0, wb0, pc[4 * (NB + b) + j], wa1, wb1);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = need ? lane_quad_total(part[b]) : f_nan();
}
