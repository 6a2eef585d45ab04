// bracket search on LDS-staged axes (with reciprocal spacings) and the O(1) EEP axis
// (part of iso_fast_kernel.h: included inside namespace iso::fastk)
#pragma once

// ---- brackets -----------------------------------------------------------------------------
__device__ __forceinline__ bool lds_oob(const double* lds, const FastAxis ax, double x)
{
    // (| and & instead of || and && in the bounds tests of this file and of lnpost_wave.h: a short-circuit makes every
    // operand a branch of its own, and the bound reads then go to LDS one after the other instead of together)
    return bool((x < lds[ax.off]) | (x > lds[ax.off + ax.n - 1]));
}

// Where the bisection of an axis starts (fast/axis_lut.h): a byte table indexed by the exponent and leading mantissa
// bits of x + c gives a node at or below x, and the bracket lies within the next lut_win(ax) nodes - the same integer
// #{a_j <= x} - 1 the reference's searchsorted (interp.py:10-35) arrives at, in 1-3 levels instead of 4-8.
__device__ __forceinline__ int lut_start(const double* lds, const FastAxis ax, double x)
{
    // The bucket number is clamped to the table (one v_med3): the brackets are computed for every lane, in range or not
    // (NaN, outside the axis - such a lane's result is never used), so that the table read does not wait behind the
    // bounds test.  For a_0 <= x <= a_last the clamp changes nothing (the bucket function is monotone).
    const uint8_t* tab = reinterpret_cast<const uint8_t*>(lds) + ax.lut;
    const int b = (__double2hiint(x + __hiloint2double(ax.chi, 0)) >> (ax.shw & 31)) - ax.b0;
    return (int)tab[max(0, min(b, (int)((unsigned)ax.shw >> 17)))];
}
__device__ __forceinline__ int lut_win(const FastAxis ax) { return (ax.shw >> 5) & 4095; }

__device__ __forceinline__ void lds_bracket(const double* lds, const FastAxis ax, double x, int& i, double& t)
{
    const double* a = lds + ax.off;
    int base = lut_start(lds, ax, x), len = lut_win(ax);
    while (len > 1) {
        const int half = len >> 1;
        base = (a[base + half] <= x) ? base + half : base;
        len -= half;
    }
    base = min(base, ax.n - 2);
    i = base;
    t = (x - a[base]) * a[ax.n + base];
}

// (Round 5, measured and not kept: the last level of the bisection and the reads of the bracket's node and reciprocal spacing
// in one LDS round trip, both candidates read and selected afterwards - 60 cycles off each bracket phase of a lone
// workgroup's half-step by the phase stamps, nothing on the clock: cfg 4 8.75 vs 8.78 us per step,
// profiles/r05/ab_spec_brackets.jsonl.)
// The same (windowed) bisection for several axes in lock-step: the LDS reads of one level are issued back to back,
// so a sample pays one LDS latency per level instead of one per level per axis (an axis that has
// converged re-reads its node, which changes nothing).
__device__ __forceinline__ void lds_bracket2(const double* lds, const FastAxis axa, const FastAxis axb, double xa,
                                             double xb, int& ia, int& ib, double& ta, double& tb)
{
    const double* a = lds + axa.off;
    const double* b = lds + axb.off;
    int ba = lut_start(lds, axa, xa), bb = lut_start(lds, axb, xb), la = lut_win(axa), lb = lut_win(axb);
    while ((la | lb) > 1) {
        const int ha = la >> 1, hb = lb >> 1;
        const double va = a[ba + ha], vb = b[bb + hb];
        ba = (va <= xa) ? ba + ha : ba;
        bb = (vb <= xb) ? bb + hb : bb;
        la -= ha;
        lb -= hb;
    }
    ba = min(ba, axa.n - 2);
    bb = min(bb, axb.n - 2);
    ia = ba;
    ib = bb;
    ta = (xa - a[ba]) * a[axa.n + ba];
    tb = (xb - b[bb]) * b[axb.n + bb];
}

__device__ __forceinline__ void lds_bracket4(const double* lds, const FastAxis ax0, const FastAxis ax1,
                                             const FastAxis ax2, const FastAxis ax3, double x0, double x1, double x2,
                                             double x3, int& i0, int& i1, int& i2, int& i3, double& t0, double& t1,
                                             double& t2, double& t3)
{
    const double* a0 = lds + ax0.off;
    const double* a1 = lds + ax1.off;
    const double* a2 = lds + ax2.off;
    const double* a3 = lds + ax3.off;
    int b0 = lut_start(lds, ax0, x0), b1 = lut_start(lds, ax1, x1), b2 = lut_start(lds, ax2, x2), b3 = lut_start(lds, ax3, x3);
    int l0 = lut_win(ax0), l1 = lut_win(ax1), l2 = lut_win(ax2), l3 = lut_win(ax3);
    while ((l0 | l1 | l2 | l3) > 1) {
        const int h0 = l0 >> 1, h1 = l1 >> 1, h2 = l2 >> 1, h3 = l3 >> 1;
        const double v0 = a0[b0 + h0], v1 = a1[b1 + h1], v2 = a2[b2 + h2], v3 = a3[b3 + h3];
        b0 = (v0 <= x0) ? b0 + h0 : b0;
        b1 = (v1 <= x1) ? b1 + h1 : b1;
        b2 = (v2 <= x2) ? b2 + h2 : b2;
        b3 = (v3 <= x3) ? b3 + h3 : b3;
        l0 -= h0;
        l1 -= h1;
        l2 -= h2;
        l3 -= h3;
    }
    b0 = min(b0, ax0.n - 2);
    b1 = min(b1, ax1.n - 2);
    b2 = min(b2, ax2.n - 2);
    b3 = min(b3, ax3.n - 2);
    i0 = b0; i1 = b1; i2 = b2; i3 = b3;
    t0 = (x0 - a0[b0]) * a0[ax0.n + b0];
    t1 = (x1 - a1[b1]) * a1[ax1.n + b1];
    t2 = (x2 - a2[b2]) * a2[ax2.n + b2];
    t3 = (x3 - a3[b3]) * a3[ax3.n + b3];
}

// Third model axis (EEP).  Exactly uniform (MIST's integer EEPs): O(1) index with an exact fix-up.  Anything else
// (A.e_axis != null; the reference bisects every axis alike, interp.py:10-35): every 8th node is staged in LDS with the
// other axes (A.ec), a branch-free bisection over those picks a window of 9 consecutive nodes, which come from the axis
// itself (a few KB, L2-resident: five independent loads, one round trip) and are counted against x in registers:
//   i = clamp(#{a_j <= x} - 1, 0, n - 2),  t = (x - a_i) / (a_{i+1} - a_i)        (a true division: no table of 1/spacing)
// - the rule of the LDS axes, i.e. searchsorted + find_indices bit for bit.  The branch is wave-uniform (a kernel
// argument), so the uniform case pays one scalar compare for it.
__device__ __forceinline__ int cvt_i32_saturating(double v)
{
    int r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

__device__ __forceinline__ bool eep_oob(const FastArgs& A, double x)
{
    return bool((x < A.e_a0) | (x > A.e_last));
}

__device__ __forceinline__ void eep_bracket(const FastArgs& A, const double* lds, double x, int& i, double& t)
{
    const int n = A.e_n;
    if (A.e_axis) {
        const double* c = lds + A.ec.off;
        int base = lut_start(lds, A.ec, x), len = lut_win(A.ec);
        while (len > 1) {
            const int half = len >> 1;
            base = (c[base + half] <= x) ? base + half : base;
            len -= half;
        }
        base = min(base * 8, n - 9);                       // window [base, base + 8]; the host guarantees n >= 9
        const double2* __restrict__ w2 = reinterpret_cast<const double2*>(A.e_axis + base);
        double v[9];
        if ((base & 1) == 0) {                              // 16-B aligned window: four 16-B loads + one 8-B load
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 u = w2[q];
                v[2 * q] = u.x;
                v[2 * q + 1] = u.y;
            }
            v[8] = A.e_axis[base + 8];
        } else {
#pragma unroll
            for (int q = 0; q < 9; ++q) v[q] = A.e_axis[base + q];
        }
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) cnt += (v[q] <= x) ? 1 : 0;
        const int k = max(0, min(cnt - 1, 7));
        double lo = v[0], hi = v[1];
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            lo = (k == q) ? v[q] : lo;
            hi = (k == q) ? v[q + 1] : hi;
        }
        i = base + k;
        t = (x - lo) / (hi - lo);
        return;
    }
    // brackets are computed for every lane, usable or not (lnpost_wave): x may be NaN or far outside the axis here.  A C++
    // cast of such a value is undefined (fptosi poison); the hardware conversion is not - v_cvt_i32_f64 saturates and
    // maps NaN to 0 - so the instruction is named directly: same single instruction, defined for every input
    int k = cvt_i32_saturating((x - A.e_a0) * A.e_inv);
    k = max(0, min(k, n - 2));
    const double lo = fma((double)k, A.e_step, A.e_a0);
    if (lo > x) --k;
    else if (k < n - 2 && fma((double)(k + 1), A.e_step, A.e_a0) <= x) ++k;
    k = max(0, min(k, n - 2));
    i = k;
    t = (x - fma((double)k, A.e_step, A.e_a0)) * A.e_inv;
}
