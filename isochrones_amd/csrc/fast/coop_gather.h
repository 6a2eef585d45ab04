// wave-cooperative gathers over the corner-packed tables (4 lanes per sample, LDS slots, DPP quad sums)
// (part of iso_fast_kernel.h: included inside namespace iso::fastk)
#pragma once

// ---- the kernel ---------------------------------------------------------------------------
// MULTI: every row carries the index of its own star (observations + priors) — the catalog /
// batched-ensemble form: S stars x W walkers in one launch.
// ---- wave-cooperative gathers over the corner-packed tables --------------------------------
// A lane-per-sample gather issues 24 + 8 x 16-B loads per lane with 64 unrelated addresses per
// wave instruction; measured ceiling of that pattern on MI355X: 4.4 TB/s of useful bytes
// (tools/gather_probe.hip).  Letting 4 lanes share one sample — each wave instruction then covers
// 16 samples x 64 contiguous bytes — reaches 7.1 TB/s.  The sample's owner lane publishes
// (cell, t0..t3) in a wave-private LDS slot; each group of 4 lanes serves one sample per iteration
// (4 iterations per wave), weights its share of the corners, sums over the group with two DPP
// quad permutes (no LDS traffic) and writes the result to the owner's response slot.  The packed
// tables are laid out for exactly this access (k_pack_star4 / k_pack_bc4 in iso_hip.hip):
//   model cell: 24 double2 "pieces"; piece (k, j) = index 4k+j holds columns (2q, 2q+1), q = k%3,
//               of corner c = 4*(k/3) + j  (c bit2/bit1/bit0 = +1 on axis 0/1/2);
//   BC cell:    piece ((k*NB + e)*4 + j) = {band e at Av node i3, band e at i3+1} of the corner
//               with axis-0 offset k and (axis-1, axis-2) offsets = the two bits of j.
// One slot per sample serves as request (header, t0..t3) and then as response (<= 8 values): a
// slot's request is read only in the iteration that serves it, and its response is written later
// in that same iteration, so the two may share storage.  Stride 9 doubles: conflict-free b64 access.
// (13 doubles when more than 8 bands are gathered.)
constexpr int slot_stride(int nb) { return nb <= 8 ? 9 : 13; }
constexpr int coop_lds_doubles(int nb) { return BLOCK * slot_stride(nb); }
constexpr int FAST_MAX_NB = 12;

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// sum over the 4 lanes of an aligned quad (every lane ends up with the total)
__device__ __forceinline__ double quad_sum(double x)
{
    x += dpp_f64<0xB1>(x);    // quad_perm [1,0,3,2]
    x += dpp_f64<0x4E>(x);    // quad_perm [2,3,0,1]
    return x;
}

// Cell numbers of the corner-packed tables are 32-bit (the host packs a table only if its cells fit): products of the
// low words, the same number the 64-bit expression gave after truncation - without 64-bit multiplies and with one
// scalar register per stride instead of two.
__device__ __forceinline__ uint32_t cell3(const FastArgs& A, int i0, int i1, int i2)
{
    return (uint32_t)i0 * (uint32_t)A.s0 + (uint32_t)i1 * (uint32_t)A.s1 + (uint32_t)i2;
}
__device__ __forceinline__ uint32_t cell4(const FastArgs& A, int j0, int j1, int j2, int j3)
{
    return (uint32_t)j0 * (uint32_t)A.bs0 + (uint32_t)j1 * (uint32_t)A.bs1 + (uint32_t)j2 * (uint32_t)A.bs2 + (uint32_t)j3;
}

struct CoopLds {
    double* req;    // this wave's 64 request slots
    double* rsp;    // this wave's 64 response slots (same storage)
    int lane;
    int stride;     // doubles per slot (compile-time constant after inlining)
};

// Model table: every lane may own one request (need, cell, w); returns the 6 interpolated columns
// of the lane's own sample in v (NaN if !need).  Must be called by all 64 lanes of the wave.
// `between` (optional) runs once, after the loads of the first two rounds have been issued and before their data is
// used: work that does not depend on the gather, for a kernel that has nothing else to hide the wait with (the
// single-model sampler; the first half is then never skipped).
struct NoWorkBetween {
    __device__ __forceinline__ void operator()() const {}
};
// DEEP: the loads of all four rounds (24 x 16 B per lane) are in flight before the first use.  Slower than two batches of two
// wherever the rounds of a wave serve one star (profiles/r04/coop_deep_batches_ab.jsonl) - the short batches overlap already -
// but in the one-star-per-lane form the lower rounds are the primaries' and the upper ones the companions': two batches would
// put the two stars' latencies one behind the other again.
// THREE: three lanes per sample, no cross-lane sums (below) - the form of a lone workgroup (lnpost_wave's LANE bit 3).
template <bool RUN_BETWEEN = false, class Between = NoWorkBetween, bool DEEP = false, bool THREE = (ISO_COOP_STAR3 != 0)>
__device__ __forceinline__ void coop_star(const FastArgs& A, const CoopLds& L, bool need, uint32_t cell, const W3& w,
                                          double* __restrict__ v, Between&& between = Between())
{
    double* mine = L.req + L.lane * L.stride;
    mine[0] = __hiloint2double(need ? 1 : 0, (int)cell);
    mine[1] = w.t0;
    mine[2] = w.t1;
    mine[3] = w.t2;
    __builtin_amdgcn_wave_barrier();
    const unsigned long long m = __ballot(need);
    const int j = L.lane & 3, grp = L.lane >> 2;
    ISO_STAMP_HERE(10);
    // two batches of two iterations: the 12 loads of a batch are in flight before the first use (DEEP: one batch of four)
    constexpr int PER = DEEP ? 4 : 2;
    if constexpr (THREE) {
    // THREE LANES PER SAMPLE, NO CROSS-LANE SUMS.  Lane q = 0..2 of a quad takes column pair q of all eight corners (the
    // pieces 12 (c / 4) + 4 q + c % 4 of the same corner-packed cell; lane 3 shadows lane 2 - same addresses, merged by the
    // texture unit) and forms, per column, exactly what the four lanes of a quad formed between them: the shares
    // fma(hi_j, whi_j, lo_j * wlo_j), j = 0..3, then (p0 + p1) + (p2 + p3) - the same bits, without the 24 DPP moves and
    // 12 additions per round of the quad form (95 -> 55 vector instructions per round; 8 loads of 16 B per lane instead
    // of 6, three quarters of the lanes active).  The weights of an unneeded slot are not zeroed: nobody reads its response.
    // Measured (profiles/r05/ab_star3_qbig.jsonl, same bits everywhere): a single star's fit 9.11 -> 8.73 us per step; 10^6-row
    // batches LOSE (cache-resident 45.8 -> 52 us, binary 106 -> 118 us: a third more load instructions through the texture
    // unit and four short batches instead of two), catalogs +2-3 %: so this is the lone workgroup's form only.
    // Rounds per batch (8 loads of 16 B per lane and round in flight before the first use): ISO_COOP_STAR3_PER, or all four (DEEP).
    const int qcol = j < 2 ? j : 2;
    // (a kernel that runs work `between` is a lone workgroup's latency form with registers to spare: both rounds of its first
    // half in flight at once)
    constexpr int PER3 = DEEP ? 4 : (RUN_BETWEEN ? 2 : ISO_COOP_STAR3_PER);
    constexpr unsigned long long BATCH_MASK = PER3 == 4 ? ~0ull : ((1ull << (16 * PER3)) - 1ull);
#pragma unroll
    for (int half = 0; half < 4 / PER3; ++half) {
        if (!DEEP && (!RUN_BETWEEN || half != 0) && ((m >> (16 * PER3 * half)) & BATCH_MASK) == 0) continue;          // wave-uniform
        constexpr int PER = PER3;
        double2 u[PER][8];
        double tt[PER][3];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if ((DEEP || PER > 1) && k > 0 && ((m >> (16 * (PER * half + k))) & 0xFFFFull) == 0) continue;      // wave-uniform: nobody owns these 16 slots
            const int src = 16 * (PER * half + k) + grp;
            const double* rq = L.req + src * L.stride;
            const double hdr = rq[0];
            const double t0 = rq[1], t1 = rq[2], t2 = rq[3];
            const bool nd = __double2hiint(hdr) != 0;
            const uint32_t c = nd ? (uint32_t)__double2loint(hdr) : 0u;      // cell 0 is always readable
            const double2* __restrict__ pc = reinterpret_cast<const double2*>(A.hotq + (size_t)c * PACK_ENTRY) + 4 * qcol;
#pragma unroll
            for (int e = 0; e < 8; ++e) u[k][e] = pc[12 * (e >> 2) + (e & 3)];
            tt[k][0] = t0;
            tt[k][1] = t1;
            tt[k][2] = t2;
        }
        if (RUN_BETWEEN && half == 0) between();
        if (half == 0) { ISO_STAMP(11, tt[PER - 1][0]); ISO_STAMP(12, u[0][0].x); ISO_STAMP(13, u[PER - 1][7].y); }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if ((DEEP || PER > 1) && k > 0 && ((m >> (16 * (PER * half + k))) & 0xFFFFull) == 0) continue;
            const int src = 16 * (PER * half + k) + grp;
            // (the eight weights are formed here, not next to the loads: 3 instead of 8 values per round kept across the wait)
            const double t0 = tt[k][0], t1 = tt[k][1], t2 = tt[k][2];
            const double a0 = 1 - t0, a1 = 1 - t1, a2 = 1 - t2;
            double px[4], py[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const double g = ((jj & 2) ? t1 : a1) * ((jj & 1) ? t2 : a2);
                const double wl = a0 * g, wh = t0 * g;
                px[jj] = corner_pair(u[k][jj].x, wl, u[k][4 + jj].x, wh);
                py[jj] = corner_pair(u[k][jj].y, wl, u[k][4 + jj].y, wh);
            }
            double* rs = L.rsp + src * L.stride + 2 * qcol;
            if (j < 3) {
                rs[0] = lane_quad_total(px);
                rs[1] = lane_quad_total(py);
            }
        }
    }
    } else {
#pragma unroll
    for (int half = 0; half < 4 / PER; ++half) {
        if (!DEEP && (!RUN_BETWEEN || half != 0) && ((m >> (32 * half)) & 0xFFFFFFFFull) == 0) continue;          // wave-uniform
        double2 u[PER][6];
        double wlo[PER], whi[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (DEEP && k > 0 && ((m >> (16 * k)) & 0xFFFFull) == 0) continue;      // wave-uniform: nobody owns these 16 slots
            const int src = 16 * (PER * half + k) + grp;
            const double* rq = L.req + src * L.stride;
            const double hdr = rq[0];
            const double t0 = rq[1], t1 = rq[2], t2 = rq[3];
            const bool nd = __double2hiint(hdr) != 0;
            const uint32_t c = nd ? (uint32_t)__double2loint(hdr) : 0u;      // cell 0 is always readable
            const double2* __restrict__ pc = reinterpret_cast<const double2*>(A.hotq + (size_t)c * PACK_ENTRY) + j;
#pragma unroll
            for (int e = 0; e < 6; ++e) u[k][e] = pc[4 * e];
            const double g = ((j & 2) ? t1 : (1 - t1)) * ((j & 1) ? t2 : (1 - t2));
            wlo[k] = nd ? (1 - t0) * g : 0.0;     // corners 0..3 (axis-0 offset 0)
            whi[k] = nd ? t0 * g : 0.0;           // corners 4..7
        }
        if (RUN_BETWEEN && half == 0) between();
        if (half == 0) { ISO_STAMP(11, whi[1]); ISO_STAMP(12, u[0][0].x); ISO_STAMP(13, u[1][5].y); }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (DEEP && k > 0 && ((m >> (16 * k)) & 0xFFFFull) == 0) continue;
            const int src = 16 * (PER * half + k) + grp;
            double part[6];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                part[2 * q] = quad_sum(corner_pair(u[k][q].x, wlo[k], u[k][3 + q].x, whi[k]));
                part[2 * q + 1] = quad_sum(corner_pair(u[k][q].y, wlo[k], u[k][3 + q].y, whi[k]));
            }
            double* rs = L.rsp + src * L.stride;
            // spread the six stores over the quad: lane j writes values j and j+4
            const double a0 = (j == 0) ? part[0] : (j == 1) ? part[1] : (j == 2) ? part[2] : part[3];
            rs[j] = a0;
            if (j < 2) rs[4 + j] = (j == 0) ? part[4] : part[5];
        }
    }
    }
    ISO_STAMP_HERE(14);
    __builtin_amdgcn_wave_barrier();
    const double* rs = L.rsp + L.lane * L.stride;
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = need ? rs[q] : f_nan();
    __builtin_amdgcn_wave_barrier();
}

// One column pair of the model table on its own corner-packed array ([cell][8 corners][2], 128 B per cell:
// the asteroseismic (nu_max, delta_nu) pair): same protocol as coop_star with two 16-B loads per lane.
__device__ __forceinline__ void coop_pair(const double* __restrict__ tab, const CoopLds& L, bool need, uint32_t cell,
                                          const W3& w, double* __restrict__ v)
{
    double* mine = L.req + L.lane * L.stride;
    mine[0] = __hiloint2double(need ? 1 : 0, (int)cell);
    mine[1] = w.t0;
    mine[2] = w.t1;
    mine[3] = w.t2;
    __builtin_amdgcn_wave_barrier();
    const unsigned long long m = __ballot(need);
    const int j = L.lane & 3, grp = L.lane >> 2;
    double2 lo[4], hi[4];
    double wlo[4], whi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double* rq = L.req + (16 * k + grp) * L.stride;
        const double hdr = rq[0];
        const double t0 = rq[1], t1 = rq[2], t2 = rq[3];
        const bool nd = __double2hiint(hdr) != 0;
        const uint32_t c = nd ? (uint32_t)__double2loint(hdr) : 0u;
        const double2* __restrict__ pc = reinterpret_cast<const double2*>(tab + (size_t)c * 16) + j;
        lo[k] = pc[0];
        hi[k] = pc[4];
        const double g = ((j & 2) ? t1 : (1 - t1)) * ((j & 1) ? t2 : (1 - t2));
        wlo[k] = nd ? (1 - t0) * g : 0.0;
        whi[k] = nd ? t0 * g : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (((m >> (16 * k)) & 0xFFFFull) == 0) continue;                      // wave-uniform
        const double a = quad_sum(corner_pair(lo[k].x, wlo[k], hi[k].x, whi[k]));
        const double b = quad_sum(corner_pair(lo[k].y, wlo[k], hi[k].y, whi[k]));
        double* rs = L.rsp + (16 * k + grp) * L.stride;
        if (j == 0) rs[0] = a;
        if (j == 1) rs[1] = b;
    }
    __builtin_amdgcn_wave_barrier();
    const double* rs = L.rsp + L.lane * L.stride;
    v[0] = need ? rs[0] : f_nan();
    v[1] = need ? rs[1] : f_nan();
    __builtin_amdgcn_wave_barrier();
}

// BC table: lane j of a quad handles the corners whose (axis-1, axis-2) offsets are the bits of j
template <int NB, bool DEEP = false>
__device__ __forceinline__ void coop_bc(const FastArgs& A, const CoopLds& L, bool need, uint32_t cell, const W4& w,
                                        double* __restrict__ v)
{
    double* mine = L.req + L.lane * L.stride;
    mine[0] = __hiloint2double(need ? 1 : 0, (int)cell);
    mine[1] = w.t0;
    mine[2] = w.t1;
    mine[3] = w.t2;
    mine[4] = w.t3;
    __builtin_amdgcn_wave_barrier();
    const unsigned long long m = __ballot(need);
    const int j = L.lane & 3, grp = L.lane >> 2;
    // batches sized so that <= 12 x 16-B loads per lane are in flight before the first use (DEEP, see coop_star: <= 32, and
    // never fewer than two rounds - a primary's and a companion's)
    // a batch of two is then rounds {r, r + 2} - one of the primaries', one of the companions')
    constexpr int BATCH = DEEP ? ((NB <= 4) ? 4 : (NB <= 8) ? 2 : 1) : ((NB <= 1) ? 4 : (NB <= 3) ? 2 : 1);
    constexpr bool PAIRED = DEEP && BATCH == 2;
#pragma unroll
    for (int r0 = 0; r0 < 4; r0 += BATCH) {
        if (PAIRED) {
            if ((((m >> (8 * r0)) | (m >> (32 + 8 * r0))) & 0xFFFFull) == 0) continue;                      // rounds r0/2 and 2 + r0/2
        } else if (((m >> (16 * r0)) & ((BATCH == 4) ? ~0ull : ((1ull << (16 * BATCH)) - 1ull))) == 0) continue;   // wave-uniform
        double2 x[BATCH][2 * NB];
        double wa[BATCH][2], wb[BATCH][2];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int src = 16 * (PAIRED ? (2 * k + (r0 >> 1)) : (r0 + k)) + grp;
            const double* rq = L.req + src * L.stride;
            const double hdr = rq[0];
            const double t0 = rq[1], t1 = rq[2], t2 = rq[3], t3 = rq[4];
            const bool nd = __double2hiint(hdr) != 0;
            const uint32_t c = nd ? (uint32_t)__double2loint(hdr) : 0u;
            const double2* __restrict__ pc = reinterpret_cast<const double2*>(A.bcq + (size_t)c * (16 * NB)) + j;
#pragma unroll
            for (int e = 0; e < 2 * NB; ++e) x[k][e] = pc[4 * e];           // e = kk*NB + band
            const double g = nd ? ((j & 2) ? t1 : (1 - t1)) * ((j & 1) ? t2 : (1 - t2)) : 0.0;
            wa[k][0] = (1 - t0) * g * (1 - t3);
            wb[k][0] = (1 - t0) * g * t3;
            wa[k][1] = t0 * g * (1 - t3);
            wb[k][1] = t0 * g * t3;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int src = 16 * (PAIRED ? (2 * k + (r0 >> 1)) : (r0 + k)) + grp;
            double* rs = L.rsp + src * L.stride;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const double part = quad_sum(corner_quad(x[k][b], wa[k][0], wb[k][0], x[k][NB + b], wa[k][1], wb[k][1]));
                if (j == (b & 3)) rs[b] = part;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const double* rs = L.rsp + L.lane * L.stride;
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = need ? rs[b] : f_nan();
    __builtin_amdgcn_wave_barrier();
}

// A tile of TB bands [b0, b0 + TB) of a BC cell that holds `nbt` bands (the pack's piece index of band e at axis-0
// offset kk is ((kk * nbt + e) * 4 + j)); bands beyond nbt - 1 repeat the last one (their terms are masked by the
// caller).  One round per 16 samples, 2 * TB loads of 16 B per lane in flight.
template <int TB>
__device__ __forceinline__ void coop_bc_tile(const FastArgs& A, const CoopLds& L, bool need, uint32_t cell, const W4& w,
                                             int nbt, int b0, double* __restrict__ v)
{
    double* mine = L.req + L.lane * L.stride;
    mine[0] = __hiloint2double(need ? 1 : 0, (int)cell);
    mine[1] = w.t0;
    mine[2] = w.t1;
    mine[3] = w.t2;
    mine[4] = w.t3;
    __builtin_amdgcn_wave_barrier();
    const unsigned long long m = __ballot(need);
    const int j = L.lane & 3, grp = L.lane >> 2;
#pragma unroll
    for (int r0 = 0; r0 < 4; ++r0) {
        if (((m >> (16 * r0)) & 0xFFFFull) == 0) continue;                       // wave-uniform
        const int src = 16 * r0 + grp;
        const double* rq = L.req + src * L.stride;
        const double hdr = rq[0];
        const double t0 = rq[1], t1 = rq[2], t2 = rq[3], t3 = rq[4];
        const bool nd = __double2hiint(hdr) != 0;
        const uint32_t c = nd ? (uint32_t)__double2loint(hdr) : 0u;
        const double2* __restrict__ pc = reinterpret_cast<const double2*>(A.bcq + (size_t)c * (size_t)(16 * nbt)) + j;
        double2 x[2 * TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int band = min(b0 + b, nbt - 1);
            x[b] = pc[4 * band];
            x[TB + b] = pc[4 * (nbt + band)];
        }
        const double g = nd ? ((j & 2) ? t1 : (1 - t1)) * ((j & 1) ? t2 : (1 - t2)) : 0.0;
        const double wa0 = (1 - t0) * g * (1 - t3), wb0 = (1 - t0) * g * t3, wa1 = t0 * g * (1 - t3), wb1 = t0 * g * t3;
        double* rs = L.rsp + src * L.stride;
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const double part = quad_sum(corner_quad(x[b], wa0, wb0, x[TB + b], wa1, wb1));
            if (j == (b & 3)) rs[b] = part;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const double* rs = L.rsp + L.lane * L.stride;
#pragma unroll
    for (int b = 0; b < TB; ++b) v[b] = need ? rs[b] : f_nan();
    __builtin_amdgcn_wave_barrier();
}

// ---- several requests per lane (round 6): the leaves of an observation tree side by side -------------------------------------
// The tree evaluator gathers its model stars one after the other: bracket, model cell, BC bracket, BC cell, then the next star -
// two dependent memory round trips per star.  For the mailbox wave of the per-point callback (ONE sample, nothing else on
// the CU to hide them) a
// lane publishes NREQ requests (slot q * 64 + lane of a wave's NREQ * 64 slots), and the rounds of all of them are served in
// FLIGHTS that mix the requests - the loads of round k of request 0 and of request 1 are in flight together - so that two stars
// cost one round trip per table.  The arithmetic of a round is coop_star's / coop_bc's (three lanes per sample for the model
// cell - same bits as the quad form): every value is the one the single-request functions give, bit for bit.
template <int NREQ>
__device__ __forceinline__ CoopLds coop_lds_multi(double* lds, int axes_len, int stride)
{
    const int base = (axes_len + 1) & ~1;
    const int wave = threadIdx.x >> 6;
    CoopLds L;
    L.req = lds + base + wave * 64 * NREQ * stride;
    L.rsp = L.req;
    L.stride = stride;
    L.lane = threadIdx.x & 63;
    return L;
}

template <int NREQ>
__device__ __forceinline__ void coop_star_multi(const FastArgs& A, const CoopLds& L, const bool* need, const uint32_t* cell, const W3* w,
                                                double (*v)[6])
{
#pragma unroll
    for (int q = 0; q < NREQ; ++q) {
        double* mine = L.req + (q * 64 + L.lane) * L.stride;
        mine[0] = __hiloint2double(need[q] ? 1 : 0, (int)cell[q]);
        mine[1] = w[q].t0;
        mine[2] = w[q].t1;
        mine[3] = w[q].t2;
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long m[NREQ];
#pragma unroll
    for (int q = 0; q < NREQ; ++q) m[q] = __ballot(need[q]);
    const int j = L.lane & 3, grp = L.lane >> 2;
    const int qcol = j < 2 ? j : 2;
    // virtual round vr = k * NREQ + q (round k of request q); a flight = FL consecutive virtual rounds
    constexpr int FL = 4, NVR = 4 * NREQ;
#pragma unroll
    for (int f = 0; f < NVR / FL; ++f) {
        bool any = false;
#pragma unroll
        for (int i = 0; i < FL; ++i) {
            const int vr = f * FL + i, k = vr / NREQ, q = vr % NREQ;
            any |= ((m[q] >> (16 * k)) & 0xFFFFull) != 0;
        }
        if (!any) continue;                                   // wave-uniform
        double2 u[FL][8];
        double tt[FL][3];
#pragma unroll
        for (int i = 0; i < FL; ++i) {
            const int vr = f * FL + i, k = vr / NREQ, q = vr % NREQ;
            if (((m[q] >> (16 * k)) & 0xFFFFull) == 0) continue;       // wave-uniform: nobody owns these 16 slots
            const int src = q * 64 + 16 * k + grp;
            const double* rq = L.req + src * L.stride;
            const double hdr = rq[0];
            const double t0 = rq[1], t1 = rq[2], t2 = rq[3];
            const bool nd = __double2hiint(hdr) != 0;
            const uint32_t c = nd ? (uint32_t)__double2loint(hdr) : 0u;      // cell 0 is always readable
            const double2* __restrict__ pc = reinterpret_cast<const double2*>(A.hotq + (size_t)c * PACK_ENTRY) + 4 * qcol;
#pragma unroll
            for (int e = 0; e < 8; ++e) u[i][e] = pc[12 * (e >> 2) + (e & 3)];
            tt[i][0] = t0;
            tt[i][1] = t1;
            tt[i][2] = t2;
        }
#pragma unroll
        for (int i = 0; i < FL; ++i) {
            const int vr = f * FL + i, k = vr / NREQ, q = vr % NREQ;
            if (((m[q] >> (16 * k)) & 0xFFFFull) == 0) continue;
            const int src = q * 64 + 16 * k + grp;
            const double t0 = tt[i][0], t1 = tt[i][1], t2 = tt[i][2];
            const double a0 = 1 - t0, a1 = 1 - t1, a2 = 1 - t2;
            double px[4], py[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const double g = ((jj & 2) ? t1 : a1) * ((jj & 1) ? t2 : a2);
                const double wl = a0 * g, wh = t0 * g;
                px[jj] = corner_pair(u[i][jj].x, wl, u[i][4 + jj].x, wh);
                py[jj] = corner_pair(u[i][jj].y, wl, u[i][4 + jj].y, wh);
            }
            double* rs = L.rsp + src * L.stride + 2 * qcol;
            if (j < 3) {
                rs[0] = lane_quad_total(px);
                rs[1] = lane_quad_total(py);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < NREQ; ++q) {
        const double* rs = L.rsp + (q * 64 + L.lane) * L.stride;
#pragma unroll
        for (int e = 0; e < 6; ++e) v[q][e] = need[q] ? rs[e] : f_nan();
    }
    __builtin_amdgcn_wave_barrier();
}

template <int NB, int NREQ>
__device__ __forceinline__ void coop_bc_multi(const FastArgs& A, const CoopLds& L, const bool* need, const uint32_t* cell, const W4* w,
                                              double (*v)[NB])
{
#pragma unroll
    for (int q = 0; q < NREQ; ++q) {
        double* mine = L.req + (q * 64 + L.lane) * L.stride;
        mine[0] = __hiloint2double(need[q] ? 1 : 0, (int)cell[q]);
        mine[1] = w[q].t0;
        mine[2] = w[q].t1;
        mine[3] = w[q].t2;
        mine[4] = w[q].t3;
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long m[NREQ];
#pragma unroll
    for (int q = 0; q < NREQ; ++q) m[q] = __ballot(need[q]);
    const int j = L.lane & 3, grp = L.lane >> 2;
    // flights of <= 32 loads of 16 B per lane (2 NB per round), at least one round of every request
    constexpr int NVR = 4 * NREQ;
    constexpr int FL = (NB <= 4) ? 4 : NREQ;
    static_assert(NVR % FL == 0, "flights tile the virtual rounds");
#pragma unroll
    for (int f = 0; f < NVR / FL; ++f) {
        bool any = false;
#pragma unroll
        for (int i = 0; i < FL; ++i) {
            const int vr = f * FL + i, k = vr / NREQ, q = vr % NREQ;
            any |= ((m[q] >> (16 * k)) & 0xFFFFull) != 0;
        }
        if (!any) continue;                                   // wave-uniform
        double2 x[FL][2 * NB];
        double wa[FL][2], wb[FL][2];
#pragma unroll
        for (int i = 0; i < FL; ++i) {
            const int vr = f * FL + i, k = vr / NREQ, q = vr % NREQ;
            if (((m[q] >> (16 * k)) & 0xFFFFull) == 0) continue;
            const int src = q * 64 + 16 * k + grp;
            const double* rq = L.req + src * L.stride;
            const double hdr = rq[0];
            const double t0 = rq[1], t1 = rq[2], t2 = rq[3], t3 = rq[4];
            const bool nd = __double2hiint(hdr) != 0;
            const uint32_t c = nd ? (uint32_t)__double2loint(hdr) : 0u;
            const double2* __restrict__ pc = reinterpret_cast<const double2*>(A.bcq + (size_t)c * (16 * NB)) + j;
#pragma unroll
            for (int e = 0; e < 2 * NB; ++e) x[i][e] = pc[4 * e];
            const double g = nd ? ((j & 2) ? t1 : (1 - t1)) * ((j & 1) ? t2 : (1 - t2)) : 0.0;
            wa[i][0] = (1 - t0) * g * (1 - t3);
            wb[i][0] = (1 - t0) * g * t3;
            wa[i][1] = t0 * g * (1 - t3);
            wb[i][1] = t0 * g * t3;
        }
#pragma unroll
        for (int i = 0; i < FL; ++i) {
            const int vr = f * FL + i, k = vr / NREQ, q = vr % NREQ;
            if (((m[q] >> (16 * k)) & 0xFFFFull) == 0) continue;
            const int src = q * 64 + 16 * k + grp;
            double* rs = L.rsp + src * L.stride;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const double part = quad_sum(corner_quad(x[i][b], wa[i][0], wb[i][0], x[i][NB + b], wa[i][1], wb[i][1]));
                if (j == (b & 3)) rs[b] = part;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < NREQ; ++q) {
        const double* rs = L.rsp + (q * 64 + L.lane) * L.stride;
#pragma unroll
        for (int b = 0; b < NB; ++b) v[q][b] = need[q] ? rs[b] : f_nan();
    }
    __builtin_amdgcn_wave_barrier();
}
