// fused stretch-move sampler kernels: step-wise and persistent
// (part of iso_fast_kernel.h: included inside namespace iso::fastk)
#pragma once

// -------------------------------------------------------------------------------------------
// Fused stretch-move half-step ("next" row f3: device-resident ensemble sampler).
// One lane = one walker of the active half of one star's ensemble: draw a partner from the
// complementary half (Philox4x32-10 counter RNG, keyed by seed, counter = (step, half, row)),
// propose y = x_j + z (x_k - x_j), evaluate lnpost(y) with the same device function as the batch
// kernel, accept / reject in place.  ASTERO: the instantiation for models with nu_max / delta_nu terms
// (reference starmodel.py:1603-1612), as in the batch kernel a template parameter, not a runtime branch.  The active half only *reads* the other half, so a half-step
// is race-free; two launches make one emcee-style iteration (Goodman & Weare 2010; the reference
// drives emcee.EnsembleSampler with one Python lnpost call per walker, starmodel.py:951-969).
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t* out)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One stretch move of walker k of the active half of one star's ensemble.  `pos` / `lnp` / `acc_cnt` are
// that star's [W][NP] / [W] / [W] arrays (global memory in the step-wise kernel, LDS in the persistent
// one), `chain_pos` / `chain_lnp` its slab of the stored chain for this step (or null): parameter q of its row
// lr goes to chain_pos[lr * S.chain_rs + q * S.chain_ps] (row-major [rows][NP]: rs = NP, ps = 1; parameter-major
// [NP][rows]: rs = 1, ps = rows - consecutive walkers then store consecutive doubles and a (star, parameter)
// pair of the chain is contiguous per step, which is what the summaries read).  The Philox
// counter is (step, half, global row): both kernels draw identical numbers for a given move.
// UNI: every row of the launch evaluates the one model A.m[0] (a single star's fit): its constants then come through scalar
// loads instead of one vector load per field and lane (a catalog row has to index its own star's block).
// DENSE_ONE (with UNI): the register-capped form with ONE ensemble per workgroup - the star's own block A.m[star] through
// scalar loads (`star` is workgroup-uniform there).
// STDP (with UNI): that model's priors are the reference's defaults - their families are compile-time constants.
// LANE: lnpost_wave's gather / overlap form (ISO_UNI_LANE for the single-model kernel, ISO_DENSE_LANE for the register-capped
// catalog kernel, 0 otherwise).
// SHAREDP: the priors the stars of the launch share are read from the first block through the constant address space
// (scalar loads, scalar branches on their families) - the register-capped catalog kernel, which the host only picks
// for launches whose stars do share them.
// (Round 5, measured and taken back: the random numbers of the wave's NEXT move - and the logarithms of z and u2 - drawn
// while the current move's BC gather is in flight, carried in registers to the next half-step.  Bit-identical chains, and
// 2.7 % SLOWER on cfg 4 (8.44 -> 8.67 us per step, profiles/r05/ab_rng_ahead.jsonl): the wait it was meant to fill is
// shorter than ten rounds of Philox, and nine more live registers across the evaluation cost more than the start of a
// half-step gains.)
template <int KIND, int NS, int NB, bool ASTERO, bool UNI = false, bool STDP = false, int LANE = 0, bool SHAREDP = false, bool DENSE_ONE = false>
__device__ __forceinline__ void stretch_move(const FastArgs& A, const StretchArgs& S, double* lds, const CoopLds& L,
                                             bool active, bool owner, int64_t star, int k, int half, uint32_t step,
                                             double* __restrict__ pos, double* __restrict__ lnp, int32_t* acc_cnt,
                                             double* __restrict__ chain_pos, double* __restrict__ chain_lnp)
{
    constexpr int NP = NS + 4;
    const int h = S.W >> 1;
    int lr = (half ? h : 0) + k;                        // row within the star's ensemble
    ISO_STAMP(0, lr);
    const int64_t row = star * S.W + lr;
    uint32_t rnd[4];
    // the key passes through an empty asm statement: otherwise the ten round keys are computed once, kept in scalar
    // registers the kernel does not have, and come back through v_readlane in every round (20 vector-slot instructions
    // per move instead of 20 scalar additions)
    uint32_t key0 = (uint32_t)S.seed, key1 = (uint32_t)(S.seed >> 32);
    asm volatile("" : "+s"(key0), "+s"(key1));
    philox4x32_10((uint32_t)(2u * step + (uint32_t)half), (uint32_t)row, (uint32_t)((uint64_t)row >> 32), 0x51u, key0, key1, rnd);
    const int j = (int)(((uint64_t)rnd[0] * (uint64_t)h) >> 32);          // uniform in [0, h)
    const int lp = (half ? 0 : h) + j;
    const double u1 = ((double)rnd[1] + (double)(rnd[2] & 0xFFFFu) * (1.0 / 65536.0)) * (1.0 / 4294967296.0);
    const double u2 = ((double)rnd[3] + (double)(rnd[2] >> 16) * (1.0 / 65536.0) + 0.5 / 65536.0) * (1.0 / 4294967296.0);
    const double zr = (S.a - 1.0) * u1 + 1.0;
    const double z = zr * zr / S.a;
    // helper lanes (no walker of their own) only read the complementary half, which nobody writes in this
    // half-step: they evaluate the partner's position and discard the result
    const int lsrc = active ? lr : lp;
    double xk[NP], y[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        xk[q] = pos[lsrc * NP + q];
        const double xj = pos[lp * NP + q];
        y[q] = fma(z, xk[q] - xj, xj);
    }
    ISO_STAMP(1, y[0]);
    const double lold = lnp[lsrc];
    // UNI: the block is read through the constant address space (same memory; tells the compiler that none of this
    // kernel's stores can touch it, which is what a scalar load needs)
    typedef const __attribute__((address_space(4))) DevModel* const_model_ptr;
    // DENSE + UNI (round 6): the register-capped catalog form when every workgroup owns ONE ensemble (258 and more walkers: the
    // reference's default 300) - `star` is the same number in every lane, so the star's block is read through the constant
    // address space as well (scalar loads instead of a vector load per field and lane): 1 250 reference-shape stars 11.0 ->
    // 10.45 ms, 10^4 71.6 -> 70.2 (profiles/r06/one_star_blocks_ab.jsonl)
    const int64_t star_u = ((int64_t)__builtin_amdgcn_readfirstlane((int)(star >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)star);
    const DevModel& M = (DENSE_ONE && UNI) ? *(const DevModel*)((const_model_ptr)(uintptr_t)(A.m + (S.multi ? star_u : 0)))
                                            : (UNI ? *(const DevModel*)((const_model_ptr)(uintptr_t)A.m) : A.m[S.multi ? star : 0]);
    double lnp_unused, lnl_unused;
    // UNI: a single star's fit (or a few ensembles of it) - one workgroup per CU at most, nothing to overlap with
    const DevModel& MP = SHAREDP ? *(const DevModel*)((const_model_ptr)(uintptr_t)A.m)
                                 : ((!UNI && S.multi && A.shared_priors) ? A.m[0] : M);      // catalogs: the priors all stars share
    const double lnew = lnpost_wave<KIND, NS, NB, ASTERO, true, false, STDP, LANE>(A, lds, L, active, M, MP, y, false, lnp_unused, lnl_unused);
    const double lnq = (NP - 1) * fast_log(z) + lnew - lold;
    // owner: the lane that decides and stores the move (= active, except in the one-star-per-lane form, where the companion's
    // lane evaluates the same proposal alongside and leaves the rest to the primary's)
    const bool acc = owner && isfinite(lnew) && (fast_log(u2) < lnq);
    if (acc) {
#pragma unroll
        for (int q = 0; q < NP; ++q) pos[lr * NP + q] = y[q];
        lnp[lr] = lnew;
        if (acc_cnt) acc_cnt[lr] += 1;
    }
    // chain recording: every move stores the row it owns (its value for this step)
    if (owner && chain_pos) {
#pragma unroll
        for (int q = 0; q < NP; ++q) chain_pos[lr * S.chain_rs + q * S.chain_ps] = acc ? y[q] : xk[q];
    }
    if (owner && chain_lnp) chain_lnp[lr] = acc ? lnew : lold;
    ISO_STAMP(8, lr);
}

// step-wise form: one launch = one half-step of every ensemble (grid over stars x W/2 walkers);
// the throughput form for catalogs large enough to fill the chip.
//
// Its occupancy is bound by registers, and what a move carries across the evaluation of its proposal - the walker's
// own position, the stretch factor, the acceptance uniform, the old lnpost, five array addresses - costs 32 of them
// (124 against the 92 of the batch kernel on the same model: 4 instead of 5 waves per SIMD, and the kernel is
// latency-bound: profiles/r02).  So the move is built twice from the thread index: once to form the proposal, and
// again after the evaluation (the index passes through an empty asm statement there, so nothing derived from it can
// be kept) to decide and store.  Same instructions on the same inputs both times: the chain is bit-identical to the
// persistent form's.
struct MoveIds {
    bool active;
    int64_t star;
    int lr, lp;          // own row / partner row inside the ensemble
    double z, u2;
};

__device__ __forceinline__ MoveIds move_ids(const StretchArgs& S, int tid)
{
    MoveIds m;
    const int64_t t0 = (int64_t)blockIdx.x * BLOCK + tid;
    m.active = t0 < S.n_active;
    const uint32_t t = (uint32_t)(m.active ? t0 : (S.n_active - 1));       // the host keeps n_active below 2^31 here
    const uint32_t h = (uint32_t)(S.W >> 1);
    const uint32_t star = t / h;
    const int k = (int)(t - star * h);
    m.star = star;
    m.lr = (S.half ? (int)h : 0) + k;
    const int64_t row = (int64_t)star * S.W + m.lr;
    uint32_t rnd[4];
    philox4x32_10((uint32_t)(2u * S.step + (uint32_t)S.half), (uint32_t)row, (uint32_t)((uint64_t)row >> 32), 0x51u,
                  (uint32_t)S.seed, (uint32_t)(S.seed >> 32), rnd);
    const int j = (int)(((uint64_t)rnd[0] * (uint64_t)h) >> 32);
    m.lp = (S.half ? 0 : (int)h) + j;
    const double u1 = ((double)rnd[1] + (double)(rnd[2] & 0xFFFFu) * (1.0 / 65536.0)) * (1.0 / 4294967296.0);
    m.u2 = ((double)rnd[3] + (double)(rnd[2] >> 16) * (1.0 / 65536.0) + 0.5 / 65536.0) * (1.0 / 4294967296.0);
    const double zr = (S.a - 1.0) * u1 + 1.0;
    m.z = zr * zr / S.a;
    return m;
}

// five waves per SIMD up to 7 bands (93-96 registers, at most 12 B of scratch); beyond that the evaluation itself
// needs 138-168 registers and the cap would only spill
constexpr int stretch_half_waves(int nb) { return nb <= 7 ? 5 : 2; }

template <int KIND, int NS, int NB, bool ASTERO = false>
__global__ __launch_bounds__(BLOCK, stretch_half_waves(NB)) void k_stretch_half(const FastArgs A, const StretchArgs S)
{
    extern __shared__ double lds[];
    for (int j = threadIdx.x; j < A.axes_len; j += BLOCK) lds[j] = A.axes_blob[j];
    __syncthreads();
    const CoopLds L = coop_lds<NB>(lds, A.axes_len);
    constexpr int NP = NS + 4;
    // ---- the proposal ----
    double y[NP];
    bool active;
    const DevModel* Mp;
    {
        const MoveIds m = move_ids(S, (int)threadIdx.x);
        const double* __restrict__ pos = S.pos + m.star * S.W * NP;
        const int lsrc = m.active ? m.lr : m.lp;      // helper lanes evaluate the partner's position and discard it
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double xk = pos[lsrc * NP + q], xj = pos[m.lp * NP + q];
            y[q] = fma(m.z, xk - xj, xj);
        }
        active = m.active;
        Mp = A.m + (S.multi ? m.star : 0);
    }
    double lnp_unused, lnl_unused;
    const DevModel* MPp = (S.multi && A.shared_priors) ? A.m : Mp;
    const double lnew = lnpost_wave<KIND, NS, NB, ASTERO, true>(A, lds, L, active, *Mp, *MPp, y, false, lnp_unused, lnl_unused);
    // ---- decide and store: everything about the move is rebuilt from the thread index ----
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const MoveIds m = move_ids(S, tid);
    if (!m.active) return;
    double* __restrict__ pos = S.pos + m.star * S.W * NP;
    double* __restrict__ lnp = S.lnp + m.star * S.W;
    double xk[NP], yy[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        xk[q] = pos[m.lr * NP + q];
        const double xj = pos[m.lp * NP + q];
        yy[q] = fma(m.z, xk[q] - xj, xj);
    }
    const double lold = lnp[m.lr];
    const double lnq = (NP - 1) * fast_log(m.z) + lnew - lold;
    const bool acc = isfinite(lnew) && (fast_log(m.u2) < lnq);
    if (acc) {
#pragma unroll
        for (int q = 0; q < NP; ++q) pos[m.lr * NP + q] = yy[q];
        lnp[m.lr] = lnew;
        if (S.accepted) S.accepted[m.star * S.W + m.lr] += 1;
    }
    if (S.chain_pos) {
        double* __restrict__ cp = S.chain_pos + (m.star * S.W + m.lr) * S.chain_rs;
#pragma unroll
        for (int q = 0; q < NP; ++q) cp[q * S.chain_ps] = acc ? yy[q] : xk[q];
    }
    if (S.chain_lnp) S.chain_lnp[m.star * S.W + m.lr] = acc ? lnew : lold;
}

// persistent form: ALL S.nsteps iterations in a single launch.  A workgroup owns G = max(1, BLOCK / (W/2))
// whole ensembles (one lane per walker of the active half, so a 32-walker catalog packs 16 stars into a
// workgroup; a large ensemble is walked in chunks of BLOCK).  Positions (and, except in the slim form, lnpost values
// and acceptance counters) live in LDS; the two half-steps of an iteration are separated by workgroup barriers instead
// of kernel boundaries, so an iteration costs two dependent evaluation chains instead of two launches.
// Same moves, same random numbers, bit-identical chains as the step-wise form.
// LDS: [axes][request/response slots][pos R*NP]( [lnp R][acc R (int32)] ),  R = G * W rows
__host__ __device__ constexpr int persist_group(int W) { return (W >> 1) >= BLOCK ? 1 : BLOCK / (W >> 1); }
// DENSE with at most 6 bands is "slim": FOUR workgroups per CU.  That takes 128 registers (the evaluation fits once
// MachineLICM no longer hoists its constants out of the iteration loop: build flag -disable-machine-licm, 167 -> 134
// registers at the 3-wave cap, 126-128 with 12-28 B of scratch at the 4-wave cap) and 40 KB of LDS: the gather slots
// shrink to 7 doubles (5 of request, <= 6 of response) and only the positions stay in LDS - a move reads its own
// lnpost from the global array at the start of its half-step and writes it and its acceptance counter back when it
// is accepted (16 B per move next to the 816 B its gathers move; every row has exactly one owner lane).
// (single stars only: a binary's evaluation spills 68-156 B per lane at 128 registers, a triple's 124-208 B)
__host__ __device__ constexpr bool persist_slim(bool dense, int nb, int ns) { return dense && nb <= 6 && ns == 1; }
__host__ __device__ constexpr int persist_slot_stride(bool dense, int nb, int ns) { return persist_slim(dense, nb, ns) ? 7 : slot_stride(nb); }
__host__ __device__ constexpr int persist_extra_doubles(int W, int np, bool slim = false)
{
    return slim ? persist_group(W) * W * np : persist_group(W) * W * (np + 1) + (persist_group(W) * W + 1) / 2;
}

// DENSE: registers capped so that 3 (slim: 4) workgroups share a CU; the uncapped form (2 workgroups per CU) is 10 %
// faster when latency is all that matters (every workgroup resident at once, e.g. a single star's fit).
// PAIR (single binaries, at most BLOCK / 2 moves per half-step): one star per lane - lanes l and l + 32 of a wave share a
// move, the primary's lane owns it (lnpost_wave's LANE bit 4).
// TRIPLE (single triples, any number of moves): one star per ROW of a wave - lanes l, l + 16, l + 32 share a move, row 0 owns
// it, row 3 idles (lnpost_wave's LANE bit 5); a half-step is walked in chunks of 64 moves (16 per wave).
template <int KIND, int NS, int NB, bool DENSE, bool ASTERO, bool UNI, bool STDP, bool PAIR, bool TRIPLE = false>
__device__ __forceinline__ void persist_body(const FastArgs& A, const StretchArgs& S)
{
    static_assert(!(PAIR && TRIPLE), "one form of one star per lane at a time");
    extern __shared__ double lds[];
    // NT: threads of the workgroup.  BLOCK, except that the register-capped form is launched with THREE waves for ensembles
    // of 129 ... 192 moves per half-step (the reference's default 300 walkers: 64 + 64 + 22 lanes): the fourth wave of such a
    // workgroup has nothing to do but holds a quarter of its register file, and three-wave workgroups fit five to a CU instead
    // of four (launch.h).  The LDS layout stays the one of BLOCK threads.
    const int NT = DENSE ? (int)blockDim.x : BLOCK;
    for (int j = threadIdx.x; j < A.axes_len; j += NT) lds[j] = A.axes_blob[j];
    constexpr bool SLIM = persist_slim(DENSE, NB, NS);
    constexpr int STRIDE = persist_slot_stride(DENSE, NB, NS);
    CoopLds L = coop_lds<NB>(lds, A.axes_len);
    L.req = L.rsp = lds + ((A.axes_len + 1) & ~1) + (threadIdx.x >> 6) * 64 * STRIDE;
    L.stride = STRIDE;
    constexpr int NP = NS + 4;
    const int W = S.W, h = W >> 1;
    const int GL = persist_group(W);                     // ensembles the LDS arrays are laid out for
    // ensembles this launch gives a workgroup: all GL, or fewer when the host spreads a small catalog over more CUs (a wave
    // that runs alone on its SIMD and its CU's L1 finishes a half-step sooner; the moves are keyed by (step, half, row),
    // so the chain does not depend on the split)
    const int G = (S.group > 0 && S.group < GL) ? S.group : GL;
    const int per = h < NT ? h : NT;                     // lanes one ensemble occupies per chunk
    const int64_t n_ens = S.n_active / h;
    const int64_t star0 = (int64_t)blockIdx.x * G;
    const int here = (int)((n_ens - star0) < G ? (n_ens - star0) : G);   // ensembles this workgroup owns
    const int R = here * W;
    const int64_t r0 = star0 * W;
    double* lpos = lds + ((A.axes_len + 1) & ~1) + BLOCK * STRIDE;
    double* llnp = SLIM ? S.lnp + r0 : lpos + GL * W * NP;                      // slim: the global arrays themselves
    int32_t* lacc = SLIM ? (S.accepted ? S.accepted + r0 : nullptr) : reinterpret_cast<int32_t*>(llnp + GL * W);
    for (int j = threadIdx.x; j < R * NP; j += NT) lpos[j] = S.pos[r0 * NP + j];
    if (!SLIM) {
        for (int j = threadIdx.x; j < R; j += NT) {
            llnp[j] = S.lnp[r0 + j];
            lacc[j] = 0;
        }
    }
    __syncthreads();
    const int64_t rows_total = n_ens * W;
    // lane -> (ensemble g, walker kk of its active half).  A workgroup that its ensembles do not fill (a single
    // star's fit: 256 walkers = 128 moves per half-step) spreads the moves evenly over its four waves in
    // multiples of 16: the cooperative gathers serve 16 samples per round and skip empty rounds, and a wave
    // issues alone on its SIMD, so 4 x 32 moves finish sooner than 2 x 64.  The random numbers are keyed by
    // (step, half, row), not by the lane, so the chain does not depend on this mapping.
    int g = (int)threadIdx.x / per, kk = (int)threadIdx.x - g * per;
    bool mine = g < here;
    bool owns = true;
    int pw_spread = 0;                                    // moves per wave when the moves are spread over the waves (below)
    if constexpr (PAIR) {
        // here * h <= BLOCK / 2 moves (the host's condition for this form): at most 32 per wave, in multiples of 16, in the
        // lower half of the wave; the upper half mirrors them for the companions
        const int total = here * h;
        int pw = (((total + 3) >> 2) + 15) & ~15;
        pw = pw < 16 ? 16 : pw;
        const int l5 = (int)threadIdx.x & 31;
        const int a = ((int)threadIdx.x >> 6) * pw + l5;
        mine = l5 < pw && a < total;
        g = mine ? a / h : 0;
        kk = mine ? a - g * h : 0;
        owns = ((int)threadIdx.x & 32) == 0;
    } else if constexpr (TRIPLE) {
        // move a of a chunk = 16 * wave + (lane & 15), in the three lower rows of the wave; g / kk are set per chunk below
        mine = ((int)threadIdx.x & 48) != 48;
        owns = ((int)threadIdx.x & 48) == 0;
        g = 0;
        kk = ((int)threadIdx.x >> 6) * 16 + ((int)threadIdx.x & 15);
    } else if (h < NT && here * h < NT) {
        const int total = here * h;
        const int nw = NT / 64;                               // (4; 3 in a three-wave launch of the register-capped form)
        int pw = (((total + nw - 1) / nw) + 15) & ~15;
        pw = pw < 16 ? 16 : pw;
        int wv = (int)threadIdx.x >> 6;
#if ISO_DENSE_PACKED
        // DENSE is the throughput form (more workgroups than the chip holds at once): there a wave costs its SIMD the same
        // time whatever its lanes do, so the moves are PACKED into as few waves as they fill (150 moves of a 300-walker
        // ensemble: 64 + 64 + 22 instead of 48 + 48 + 48 + 6); which waves those are rotates with the workgroup number (the
        // hardware already varies the SIMD a wave index lands on - tools/simd_probe.hip -, so this is belt and braces).
        if constexpr (DENSE) {
            pw = 64;
            wv = (wv + (int)blockIdx.x) % (NT / 64);
        }
#endif
        const int a = wv * pw + ((int)threadIdx.x & 63);
        mine = ((int)threadIdx.x & 63) < pw && a < total;
        g = mine ? a / h : 0;
        kk = mine ? a - g * h : 0;
        pw_spread = pw;
    }
    const int gs = mine ? g : 0;                          // idle lanes shadow a move of the first ensemble
    // When every ensemble of this workgroup lies inside one wavefront (W/2 divides 64 and the plain lane mapping is in
    // use), a half-step only depends on LDS rows its own wave wrote: the waves then need no workgroup barrier between
    // half-steps and run through the iterations independently - a wave that waits for memory no longer holds up the
    // other three.  (LDS operations of one wave complete in order; the fence keeps the compiler from moving them.)
    // (Round 6: the same holds when the moves of a partly filled workgroup are spread over its waves in runs that are whole
    // ensembles - a small catalog at 32 walkers, four stars per workgroup: one star per wave.)
    const bool wave_local = !PAIR && !TRIPLE && h <= 64 && (pw_spread > 0 ? (pw_spread % h) == 0 : ((64 % h) == 0 && !(h < NT && here * h < NT)));
    for (int it = 0; it < S.nsteps; ++it) {
        double* cp = S.chain_pos ? S.chain_pos + (int64_t)it * rows_total * NP + (r0 + gs * W) * S.chain_rs : nullptr;
        double* cl = S.chain_lnp ? S.chain_lnp + (int64_t)it * rows_total + r0 + gs * W : nullptr;
        for (int half = 0; half < 2; ++half) {
#if ISO_KERNARG_REREAD
            // both argument blocks are read again from the kernel-argument segment in every half-step (scalar loads through
            // a pointer the optimiser cannot see through): what a move needs of them then does not have to survive the
            // evaluation in scalar registers the kernel does not have (they were spilled to vector lanes and came back
            // through v_readlane, a vector-slot instruction each)
            typedef const __attribute__((address_space(4))) char* kernarg_ptr;
            kernarg_ptr kp = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kp));
            const FastArgs& A = *(const FastArgs*)(const __attribute__((address_space(4))) FastArgs*)kp;
            const StretchArgs& S = *(const StretchArgs*)(const __attribute__((address_space(4))) StretchArgs*)(kp + ((sizeof(FastArgs) + 7) & ~size_t(7)));
#endif
            if constexpr (TRIPLE) {
                const int total = here * h;
                for (int c0 = 0; c0 < total; c0 += 64) {
                    const int a = c0 + kk;
                    const bool active = mine && a < total;
                    const int ge = active ? a / h : 0;
                    const int k = active ? a - ge * h : h - 1;
                    double* cpe = S.chain_pos ? S.chain_pos + (int64_t)it * rows_total * NP + (r0 + ge * W) * S.chain_rs : nullptr;
                    double* cle = S.chain_lnp ? S.chain_lnp + (int64_t)it * rows_total + r0 + ge * W : nullptr;
                    if (__any(active))                    // wave-uniform
                        stretch_move<KIND, NS, NB, ASTERO, UNI, STDP, ISO_UNI_LANE | 32, false>(
                            A, S, lds, L, active, active && owns, star0 + ge, k, half, S.step + (uint32_t)it, lpos + ge * W * NP,
                            llnp + ge * W, lacc ? lacc + ge * W : nullptr, cpe, cle);
                }
            } else
            for (int k0 = 0; k0 < h; k0 += per) {
                const int k = k0 + kk;
                const bool active = mine && k < h;
                if (__any(active))                        // wave-uniform: idle waves go straight to the barrier
                    stretch_move<KIND, NS, NB, ASTERO, UNI, STDP,
                                 (DENSE ? ISO_DENSE_LANE : (UNI ? ISO_UNI_LANE : (STDP ? ISO_MULTI_STD_LANE : ISO_MULTI_LANE))) | (PAIR ? 16 : 0),
                                 (DENSE && ISO_DENSE_SHARED) || (!UNI && !DENSE && STDP), DENSE && UNI>(A, S, lds, L, active, active && owns, star0 + gs, active ? k : h - 1, half,
                                               S.step + (uint32_t)it, lpos + gs * W * NP, llnp + gs * W,
                                               lacc ? lacc + gs * W : nullptr, cp, cl);
            }
            if (wave_local) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
                __syncthreads();
            }
            ISO_STAMP_HERE(9);
        }
    }
    if (wave_local) __syncthreads();                      // the write-back below reads rows of the other waves
    for (int j = threadIdx.x; j < R * NP; j += NT) S.pos[r0 * NP + j] = lpos[j];
    if (!SLIM) {
        for (int j = threadIdx.x; j < R; j += NT) {
            S.lnp[r0 + j] = llnp[j];
            if (S.accepted) S.accepted[r0 + j] += lacc[j];
        }
    }
}

template <int KIND, int NS, int NB, bool DENSE, bool ASTERO = false, bool UNI = false, bool STDP = false>
__global__ __launch_bounds__(BLOCK, DENSE ? ((persist_slim(DENSE, NB, NS) && !STDP) ? 4 : 3) : 2) void k_stretch_persist(const FastArgs A, const StretchArgs S)
{
    persist_body<KIND, NS, NB, DENSE, ASTERO, UNI, STDP, false>(A, S);
}

// a single binary's fit (isochrone grid, no asteroseismic terms, at most BLOCK / 2 moves per half-step): one star per lane
// (models whose priors are the reference's default families: compile-time constants)
template <int NB>
__global__ __launch_bounds__(BLOCK, 2) void k_stretch_pair(const FastArgs A, const StretchArgs S)
{
    persist_body<ISO_KIND_ISO, 2, NB, false, false, true, true, true>(A, S);
}

// a single triple's fit (isochrone grid, no asteroseismic terms, default prior families): one star per row of a wave, 64 moves
// per chunk.  Measured (profiles/r06/triple_sweep.jsonl, us per step at 3 / 9 bands): 16 moves per half-step 19.7 -> 15.1 /
// 24.4 -> 21.4, 32 moves 20.8 -> 15.9 / 25.2 -> 22.8, 64 moves 21.1 -> 17.2 / 25.8 -> 23.9; beyond one chunk the form
// LOSES (128 moves: 23.3 -> 32.7 / 31.4 -> 46.0) - a chunk is a whole dependent chain again, and the lanes were not idle
// there - so the host takes it up to 64 moves (StretchArgs.triple_moves).
template <int NB>
__global__ __launch_bounds__(BLOCK, 2) void k_stretch_triple(const FastArgs A, const StretchArgs S)
{
    persist_body<ISO_KIND_ISO, 3, NB, false, false, true, true, false, true>(A, S);
}
