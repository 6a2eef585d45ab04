// posterior summaries of a stored chain (LDS bitonic sort)
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

// -------------------------------------------------------------------------------------------
// posterior summaries of a stored chain: one workgroup sorts the nsteps*W values of one
// (ensemble, parameter) pair in LDS (bitonic network on the next power of two, padded with +inf)
// and writes the requested quantiles (linear interpolation between order statistics)
// -------------------------------------------------------------------------------------------
struct QuantArgs {
    const double* chain;     // element (step t, row r, parameter d) at chain[t * ss + r * rs + d * ps]
    int64_t ss, rs, ps;      // row-major [nsteps][n_ens*W][D]: (rows*D, D, 1); parameter-major [nsteps][D][rows]: (rows*D, 1, rows)
    int64_t nsteps, n_ens;
    int W, D, nq, P;         // P = power of two >= nsteps*W
    double q[8];
    double* out;             // [n_ens][D][nq]
    int only_flagged;        // workgroup kernels: handle only the (ensemble, parameter) pairs the wave kernel flagged
};

// what k_chain_quantiles_wave leaves in out[..][0] of a pair it cannot select in its small per-wave storage
// (a NaN with a payload no arithmetic produces); the workgroup kernel launched after it picks those pairs up
constexpr unsigned long long QUANT_FLAG = 0x7ff8dead00000001ULL;
__device__ __forceinline__ bool quant_flagged(const QuantArgs& A, int64_t pair)
{
    return (unsigned long long)__double_as_longlong(A.out[pair * A.nq]) == QUANT_FLAG;
}

// numpy.percentile(method="linear") position of quantile level qq among m sorted values: lower index + fraction
__device__ __forceinline__ void quantile_position(double qq, int m, int& i0, int& i1, double& f)
{
#pragma clang fp contract(off)   // pos must be rounded before the fraction is taken (no fma(m-1, q, -i0))
    const double pos = (double)(m - 1) * qq;        // numpy's virtual index for method="linear": (n - 1) q
    i0 = (int)floor(pos);
    i0 = max(0, min(i0, m - 1));
    i1 = min(i0 + 1, m - 1);
    f = pos - (double)i0;
}

__device__ __forceinline__ double quantile_lerp(double a, double b, double f)
{
#pragma clang fp contract(off)   // numpy's _lerp, unfused: a + (b-a)t, from the upper end for t >= 0.5
    const double diff = b - a;
    return (f >= 0.5) ? b - diff * (1 - f) : a + diff * f;
}

__device__ __forceinline__ double chain_value(const QuantArgs& A, int64_t e, int d, int64_t rows, int i)
{
    const int t = i / A.W, w = i - t * A.W;
    const double v = A.chain[(int64_t)t * A.ss + (e * A.W + w) * A.rs + d * A.ps];
    return (v != v) ? d_inf() : v;                   // NaN sorts last (cannot occur in an accepted chain)
}

// full sort of the (ensemble, parameter) values in LDS; all threads of the workgroup
__device__ __forceinline__ void bitonic_sort_chain(const QuantArgs& A, int64_t e, int d, int64_t rows, int m, double* lds)
{
    for (int i = threadIdx.x; i < A.P; i += BLOCK) lds[i] = (i < m) ? chain_value(A, e, d, rows, i) : d_inf();
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
}

__global__ __launch_bounds__(BLOCK, 2) void k_chain_quantiles(const QuantArgs A)
{
    extern __shared__ double lds[];
    if (A.only_flagged && !quant_flagged(A, blockIdx.x)) return;      // workgroup-uniform
    const int64_t e = blockIdx.x / A.D;
    const int d = (int)(blockIdx.x - e * A.D);
    const int m = (int)(A.nsteps * A.W);
    bitonic_sort_chain(A, e, d, A.n_ens * A.W, m, lds);
    if ((int)threadIdx.x < A.nq) {
        int i0, i1;
        double f;
        quantile_position(A.q[threadIdx.x], m, i0, i1, f);
        A.out[(e * A.D + d) * A.nq + threadIdx.x] = quantile_lerp(lds[i0], lds[i1], f);
    }
}

// -------------------------------------------------------------------------------------------
// The same quantiles by SELECTION instead of a full sort: only 2 x nq order statistics are needed.
//   pass 1  min / max of the values
//   pass 2  histogram over QSEL_BINS equal-width value bins (monotone binning: every element of a lower bin is
//           <= every element of a higher bin), prefix sum -> the bin and the rank inside it of each target
//   pass 3  the elements of the (at most 2 nq) target bins are gathered into small LDS lists; inside a list the
//           order statistic is found by counting (an element's rank = number of elements that precede it in
//           (value, position) order)
// Three coalesced passes over the chain (L2-resident after the first) and a few dozen LDS operations, against
// 78 compare-exchange passes of the bitonic network.  A target bin that holds more than QSEL_CAP elements
// (a chain with hundreds of identical values, or NaN / inf) sends the workgroup to the full sort above.
// -------------------------------------------------------------------------------------------
constexpr int QSEL_BINS = 1024;
constexpr int QSEL_CAP = 128;      // elements per gathered list
constexpr int QSEL_RANKS = 16;     // 2 x nq, nq <= 8

__global__ __launch_bounds__(BLOCK, 2) void k_chain_quantiles_select(const QuantArgs A)
{
    extern __shared__ double lds[];
    // LDS map (the bitonic fallback reuses the whole area from offset 0):
    //   [0, QSEL_RANKS * QSEL_CAP) doubles   gathered lists
    //   then QSEL_BINS ints                  histogram -> exclusive prefix
    //   then QSEL_BINS bytes                 list slot of a bin (0xFF = not a target)
    //   then small per-rank records
    double* lists = lds;
    int* hist = reinterpret_cast<int*>(lds + QSEL_RANKS * QSEL_CAP);
    unsigned char* slot_of_bin = reinterpret_cast<unsigned char*>(hist + QSEL_BINS);
    int* rank_bin = reinterpret_cast<int*>(slot_of_bin + QSEL_BINS);      // [QSEL_RANKS]
    int* rank_local = rank_bin + QSEL_RANKS;                               // [QSEL_RANKS]
    int* rank_slot = rank_local + QSEL_RANKS;                              // [QSEL_RANKS]
    int* list_count = rank_slot + QSEL_RANKS;                              // [QSEL_RANKS]
    double* red = reinterpret_cast<double*>(list_count + QSEL_RANKS);      // [2 * waves] min / max partials
    int* flags = reinterpret_cast<int*>(red + 8);                          // [0] fallback, [1] number of lists
    double* result = reinterpret_cast<double*>(flags + 2);                 // [QSEL_RANKS]

    if (A.only_flagged && !quant_flagged(A, blockIdx.x)) return;          // workgroup-uniform
    const int64_t e = blockIdx.x / A.D;
    const int d = (int)(blockIdx.x - e * A.D);
    const int m = (int)(A.nsteps * A.W);
    const int64_t rows = A.n_ens * A.W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_ranks = 2 * A.nq;

    // ---- pass 1: min / max ----
    double mn = d_inf(), mx = -d_inf();
    for (int i = tid; i < m; i += BLOCK) {
        const double v = chain_value(A, e, d, rows, i);
        mn = fmin(mn, v);
        mx = fmax(mx, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, off));
        mx = fmax(mx, __shfl_xor(mx, off));
    }
    if (lane == 0) {
        red[wave] = mn;
        red[4 + wave] = mx;
    }
    for (int i = tid; i < QSEL_BINS; i += BLOCK) {
        hist[i] = 0;
        slot_of_bin[i] = 0xFF;
    }
    if (tid == 0) flags[0] = flags[1] = 0;
    __syncthreads();
    mn = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
    mx = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
    const double inv = (double)QSEL_BINS / (mx - mn);
    const bool degenerate = !(mx > mn) || !isfinite(inv) || !isfinite(mn);      // constant chain, inf, NaN
    auto bin_of = [&](double v) { return min(QSEL_BINS - 1, (int)((v - mn) * inv)); };
    if (!degenerate) {
        // ---- pass 2: histogram + exclusive prefix ----
        for (int i = tid; i < m; i += BLOCK) atomicAdd(&hist[bin_of(chain_value(A, e, d, rows, i))], 1);
        __syncthreads();
        {   // 1024 bins, 4 per thread: thread-local sum, wave scan of the sums, cross-wave offsets
            int c[QSEL_BINS / BLOCK];
            int tot = 0;
#pragma unroll
            for (int r = 0; r < QSEL_BINS / BLOCK; ++r) {
                c[r] = hist[tid * (QSEL_BINS / BLOCK) + r];
                tot += c[r];
            }
            int incl = tot;
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            int* wsum = reinterpret_cast<int*>(red);      // min / max partials are no longer needed
            __syncthreads();
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int base = incl - tot;
            for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
            for (int r = 0; r < QSEL_BINS / BLOCK; ++r) {
                hist[tid * (QSEL_BINS / BLOCK) + r] = base;     // exclusive prefix
                base += c[r];
            }
        }
        __syncthreads();
        // ---- the target ranks: which bin, which rank inside it ----
        if (tid < n_ranks) {
            int i0, i1;
            double f;
            quantile_position(A.q[tid >> 1], m, i0, i1, f);
            const int r = (tid & 1) ? i1 : i0;
            int lo = 0, hi = QSEL_BINS - 1;                      // last bin whose exclusive prefix is <= r
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (hist[mid] <= r) lo = mid;
                else hi = mid - 1;
            }
            rank_bin[tid] = lo;
            rank_local[tid] = r - hist[lo];
        }
        __syncthreads();
        if (tid == 0) {                                          // distinct target bins -> list slots
            int n_lists = 0;
            for (int k = 0; k < n_ranks; ++k) {
                const int b = rank_bin[k];
                if (slot_of_bin[b] == 0xFF) {
                    slot_of_bin[b] = (unsigned char)n_lists;
                    list_count[n_lists] = 0;
                    const int cnt = ((b + 1 < QSEL_BINS) ? hist[b + 1] : m) - hist[b];
                    if (cnt > QSEL_CAP) flags[0] = 1;
                    ++n_lists;
                }
                rank_slot[k] = slot_of_bin[b];
            }
            flags[1] = n_lists;
        }
        __syncthreads();
    }
    const bool fallback = degenerate ? (mx > mn || !isfinite(mn) || !isfinite(mx)) : (flags[0] != 0);
    if (degenerate && !fallback) {                               // every value equal
        if (tid < A.nq) A.out[(e * A.D + d) * A.nq + tid] = mn;
        return;
    }
    if (fallback) {                                              // workgroup-uniform
        __syncthreads();
        bitonic_sort_chain(A, e, d, rows, m, lds);
        if (tid < A.nq) {
            int i0, i1;
            double f;
            quantile_position(A.q[tid], m, i0, i1, f);
            A.out[(e * A.D + d) * A.nq + tid] = quantile_lerp(lds[i0], lds[i1], f);
        }
        return;
    }
    // ---- pass 3: gather the target bins ----
    for (int i = tid; i < m; i += BLOCK) {
        const double v = chain_value(A, e, d, rows, i);
        const int sl = slot_of_bin[bin_of(v)];
        if (sl != 0xFF) {
            const int pos = atomicAdd(&list_count[sl], 1);
            lists[sl * QSEL_CAP + pos] = v;
        }
    }
    __syncthreads();
    // ---- order statistic inside a list by counting; rank k of the workgroup's targets handled by wave k % 4 ----
    for (int k = wave; k < n_ranks; k += BLOCK / 64) {
        const int sl = rank_slot[k], cnt = list_count[sl], want = rank_local[k];
        const double* L = lists + sl * QSEL_CAP;
        for (int j = lane; j < cnt; j += 64) {
            const double x = L[j];
            int before = 0;
            for (int i = 0; i < cnt; ++i) {
                const double y = L[i];
                before += (y < x || (y == x && i < j)) ? 1 : 0;
            }
            if (before == want) result[k] = x;
        }
    }
    __syncthreads();
    if (tid < A.nq) {
        int i0, i1;
        double f;
        quantile_position(A.q[tid], m, i0, i1, f);
        A.out[(e * A.D + d) * A.nq + tid] = quantile_lerp(result[2 * tid], result[2 * tid + 1], f);
    }
}


// -------------------------------------------------------------------------------------------
// Chains of ANY length (more than 8 192 values per pair: the reference's default fit keeps 300 walkers x 100 iterations =
// 30 000 per parameter, starmodel.py:889-893): selection by REFINEMENT, nothing proportional to the chain in LDS.
//   pass 1   min / max of the finite values; counts of -inf and of +inf / NaN (they sort first / last)
//   pass 2   QBIG_BINS-bin histogram over [min, max], prefix, the bin and the rank inside it of each of the 2 nq targets
//   pass 3   the target bins that hold at most QBIG_CAP values are gathered into LDS lists; rank by counting
//   refine   a target whose bin holds more (a chain with many repeats of one value, a narrow posterior): the bin's values
//            are exactly those in [lo, hi] = their own min / max (the binning is monotone), so the same three steps run on
//            that interval with the whole list area (2 048 values) as capacity - until it fits, or lo == hi (all equal).
// Every pass streams the pair's values from memory (the W values of a step are contiguous in the parameter-major chain);
// a 30 000-value pair is 240 KB - L2 / Infinity Cache hits after the first pass.  Bit for bit numpy.percentile.
// (Round 6, measured and not kept: ONE pass that also keeps the values of sampled WINDOWS around each level in LDS and ranks
// from there - 99.9 % of the targets found, the chain read 1 1/8 times, and 6.2 ms where this form takes 4.3: the pass is
// bound by instruction issue, not by the chain's bytes, and 53 KB of LDS leave three workgroups per CU.
// docs/history/qbig_windows_experiment.patch, profiles/r06/quantile_big_ab_windows_experiment.jsonl.)
// -------------------------------------------------------------------------------------------
constexpr int QBIG_BINS = 4096;
constexpr int QBIG_CAP = 128;                              // values per gathered list in the common pass (36 KB of LDS: four workgroups per CU)
constexpr int QBIG_POOL = QSEL_RANKS * QBIG_CAP;           // = the single list of a refinement pass (2 048 values)
constexpr size_t QBIG_LDS = (size_t)QBIG_POOL * 8 + QBIG_BINS * 4 + QBIG_BINS + 6 * QSEL_RANKS * 4 + 16 * 8 + 8 * 4 + QSEL_RANKS * 8;
constexpr int QBIG_UN = 4;                                 // loads in flight per lane and round (2 and 8: the same time)
constexpr int QBIG_NS = 16;                                // sampled values per thread (up to 4 096 per pair)
struct QbigValues { double v[QBIG_UN]; };

__global__ __launch_bounds__(BLOCK, 4) void k_chain_quantiles_big(const QuantArgs A)
{
    extern __shared__ double lds[];
    double* lists = lds;
    int* hist = reinterpret_cast<int*>(lds + QBIG_POOL);
    unsigned char* slot_of_bin = reinterpret_cast<unsigned char*>(hist + QBIG_BINS);
    int* rank_bin = reinterpret_cast<int*>(slot_of_bin + QBIG_BINS);       // [QSEL_RANKS]
    int* rank_local = rank_bin + QSEL_RANKS;
    int* rank_slot = rank_local + QSEL_RANKS;                               // list slot, -1 = resolved, -2 = needs refinement
    int* list_count = rank_slot + QSEL_RANKS;
    int* rank_cnt = list_count + QSEL_RANKS;                                // values in the target's bin
    int* spare = rank_cnt + QSEL_RANKS;
    double* red = reinterpret_cast<double*>(spare + QSEL_RANKS);            // [16] reduction partials
    int* ired = reinterpret_cast<int*>(red + 16);                           // [8]
    double* result = reinterpret_cast<double*>(ired + 8);                   // [QSEL_RANKS]

    const int64_t e = blockIdx.x / A.D;
    const int d = (int)(blockIdx.x - e * A.D);
    const int m = (int)(A.nsteps * A.W);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_ranks = 2 * A.nq;
    const double* __restrict__ base = A.chain + e * A.W * A.rs + d * A.ps;
    // value i of the pair (step i / W, walker i % W); NaN sorts last, as +inf
    auto value = [&](int t, int w) {
        const double v = base[(int64_t)t * A.ss + (int64_t)w * A.rs];
        return (v != v) ? d_inf() : v;
    };
    // a pass over the pair's values: thread tid takes values tid, tid + BLOCK, ... FOUR AT A TIME - four independent loads in
    // flight per lane before the first is used (one at a time, a pass was bound by the latency of its single load: 8 waves x
    // 512 B in flight per CU, 1.9 TB/s over the three passes of a 10^4-star x 300 x 100 catalog).  (step, walker) of value
    // i: t = i / W by a multiply-high with floor(2^32 / W) and one correction (exact for i < 2^32).
    const uint32_t w_magic = A.W == 1 ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / (uint32_t)A.W);     // (2^32 itself does not fit: one short, corrected like any other)
    auto value_at = [&](int i) {
        uint32_t t = __umulhi((uint32_t)i, w_magic);
        uint32_t w = (uint32_t)i - t * (uint32_t)A.W;
        const bool up = w >= (uint32_t)A.W;
        t += up ? 1u : 0u;
        w -= up ? (uint32_t)A.W : 0u;
        return value((int)t, (int)w);
    };
    // The loads of the NEXT round are issued before this round's values are used: what is done with a value (LDS atomics, a table
    // look-up) runs under the next loads' latency instead of between two round trips.
    auto load_values = [&](int i) {
        QbigValues r;
#pragma unroll
        for (int u = 0; u < QBIG_UN; ++u) r.v[u] = value_at(min(i + u * BLOCK, m - 1));
        return r;
    };
    struct QbigCursor { int i; QbigValues cur, nxt; };
#define QBIG_FOR_VALUES(v)                                                                                       \
    for (QbigCursor it_ = {tid, load_values(tid), {}}; it_.i < m; it_.i += QBIG_UN * BLOCK, it_.cur = it_.nxt)    \
        if ((it_.nxt = load_values(it_.i + QBIG_UN * BLOCK)), true)                                               \
            _Pragma("unroll") for (int u_ = 0; u_ < QBIG_UN; ++u_)                                                \
                if (it_.i + u_ * BLOCK < m)                                                                       \
                if (const double v = it_.cur.v[u_]; true)

    auto block_minmax = [&](double& mn, double& mx) {        // workgroup reduction; every thread gets the result
        for (int off = 32; off > 0; off >>= 1) {
            mn = fmin(mn, __shfl_xor(mn, off));
            mx = fmax(mx, __shfl_xor(mx, off));
        }
        __syncthreads();
        if (lane == 0) {
            red[wave] = mn;
            red[4 + wave] = mx;
        }
        __syncthreads();
        mn = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
        mx = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
    };
    auto prefix_hist = [&]() {                               // hist -> exclusive prefix (all threads)
        constexpr int PER = QBIG_BINS / BLOCK;
        int c[PER];
        int tot = 0;
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            c[r] = hist[tid * PER + r];
            tot += c[r];
        }
        int incl = tot;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        __syncthreads();
        if (lane == 63) ired[wave] = incl;
        __syncthreads();
        int b0 = incl - tot;
        for (int w = 0; w < wave; ++w) b0 += ired[w];
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            hist[tid * PER + r] = b0;
            b0 += c[r];
        }
        __syncthreads();
    };
    auto find_bin = [&](int r) {                             // last bin whose exclusive prefix is <= r
        int lo = 0, hi = QBIG_BINS - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (hist[mid] <= r) lo = mid;
            else hi = mid - 1;
        }
        return lo;
    };
    // rank `want` of the `cnt` values of a list, by counting: an element's rank = elements before it in (value, position) order
    auto select_in_list = [&](const double* Lp, int cnt, int want, int k, int first, int stride) {
        for (int j = first; j < cnt; j += stride) {
            const double x = Lp[j];
            int before = 0;
            for (int i = 0; i < cnt; ++i) {
                const double y = Lp[i];
                before += (y < x || (y == x && i < j)) ? 1 : 0;
            }
            if (before == want) result[k] = x;
        }
    };

    // ---- pass 1: the range of the histogram - from a SAMPLE.  Any monotone binning gives the same selection (the target bins
    // are gathered and ranked exactly; a crowded bin is refined on its own min / max), so the range need not be the exact one:
    // the finite min / max of every 8th chunk of BLOCK values (an eighth of the traffic of a pass) spans the histogram, values
    // outside fall into its end bins, and the counts of the infinities come out of the histogram pass.  A sample without two
    // distinct finite values (a constant chain, a chain of infinities) takes the exact pass instead. ----
    // The sample's loads are all in flight at once (QBIG_NS per thread, one round trip; a loop over the sample took a round
    // trip per value: 15 for 30 000 values); a chain of more than QBIG_NS x 8 chunks is sampled at a wider stride.
    double mn = d_inf(), mx = -d_inf();
    int n_neg = 0, n_pos = 0;
    {
        const int n_chunks = (m + BLOCK - 1) / BLOCK;
        const int sample_stride = max(8, (n_chunks + QBIG_NS - 1) / QBIG_NS) * BLOCK;
        double sv[QBIG_NS];
#pragma unroll
        for (int u = 0; u < QBIG_NS; ++u) {
            const int i = tid + u * sample_stride;
            const double v = value_at(min(i, m - 1));
            sv[u] = (i < m) ? v : d_inf();                   // (an infinity is no part of the sample)
        }
#pragma unroll
        for (int u = 0; u < QBIG_NS; ++u) {
            const bool inf = (sv[u] == -d_inf()) | (sv[u] == d_inf());
            mn = inf ? mn : fmin(mn, sv[u]);
            mx = inf ? mx : fmax(mx, sv[u]);
        }
    }
    block_minmax(mn, mx);
    const bool exact_range = !(mx > mn);                     // workgroup-uniform (block_minmax gives every thread the same numbers)
    if (exact_range) {
        mn = d_inf();
        mx = -d_inf();
        QBIG_FOR_VALUES(v) {
            const bool neg = v == -d_inf(), pos = v == d_inf();
            n_neg += neg;
            n_pos += pos;
            mn = (neg | pos) ? mn : fmin(mn, v);
            mx = (neg | pos) ? mx : fmax(mx, v);
        }
        block_minmax(mn, mx);
    }
    for (int i = tid; i < QBIG_BINS; i += BLOCK) {
        hist[i] = 0;
        slot_of_bin[i] = 0xFF;
    }
    __syncthreads();
    const bool binned = mx > mn;                             // (exact range: false = every finite value equal, or none)
    const double inv = binned ? (double)QBIG_BINS / (mx - mn) : 0.0;
    // (clamped: values outside a sampled range belong to the end bins; the comparison form also keeps the conversion in range)
    auto bin_of = [&](double v) { return (int)fmin(fmax((v - mn) * inv, 0.0), (double)(QBIG_BINS - 1)); };
    if (binned) {
        // ---- pass 2: histogram over [mn, mx]; with a sampled range also the counts of the infinities ----
        QBIG_FOR_VALUES(v) {
            const bool neg = v == -d_inf(), pos = v == d_inf();
            if (!exact_range) {
                n_neg += neg;
                n_pos += pos;
            }
            if (!(neg | pos)) atomicAdd(&hist[bin_of(v)], 1);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        n_neg += __shfl_xor(n_neg, off);
        n_pos += __shfl_xor(n_pos, off);
    }
    __syncthreads();
    if (lane == 0) {
        ired[wave] = n_neg;
        ired[4 + wave] = n_pos;
    }
    __syncthreads();
    n_neg = ired[0] + ired[1] + ired[2] + ired[3];
    n_pos = ired[4] + ired[5] + ired[6] + ired[7];
    const int mfin = m - n_neg - n_pos;
    __syncthreads();
    // ---- the targets: order statistics i0, i1 of each level ----
    if (tid < n_ranks) {
        int i0, i1;
        double f;
        quantile_position(A.q[tid >> 1], m, i0, i1, f);
        const int r = (tid & 1) ? i1 : i0;
        rank_slot[tid] = -2;
        rank_local[tid] = r - n_neg;                         // rank among the finite values
        if (r < n_neg) { result[tid] = -d_inf(); rank_slot[tid] = -1; }
        else if (r >= n_neg + mfin) { result[tid] = d_inf(); rank_slot[tid] = -1; }
        else if (!binned) { result[tid] = mn; rank_slot[tid] = -1; }      // every finite value equal (exact range)
    }
    __syncthreads();
    if (mfin > 0 && binned) {                                // workgroup-uniform
        prefix_hist();
        if (tid < n_ranks && rank_slot[tid] == -2) {
            const int b = find_bin(rank_local[tid]);
            rank_bin[tid] = b;
            rank_cnt[tid] = ((b + 1 < QBIG_BINS) ? hist[b + 1] : mfin) - hist[b];
            rank_local[tid] -= hist[b];
        }
        __syncthreads();
        if (tid == 0) {                                      // distinct target bins that fit a list -> list slots
            int n_lists = 0;
            for (int k = 0; k < n_ranks; ++k) {
                if (rank_slot[k] != -2 || rank_cnt[k] > QBIG_CAP) continue;
                const int b = rank_bin[k];
                if (slot_of_bin[b] == 0xFF) {
                    slot_of_bin[b] = (unsigned char)n_lists;
                    list_count[n_lists] = 0;
                    ++n_lists;
                }
                rank_slot[k] = slot_of_bin[b];
            }
        }
        __syncthreads();
        // ---- pass 3: gather, select ----
        QBIG_FOR_VALUES(v) {
            if (v > -d_inf() && v < d_inf()) {
                const int sl = slot_of_bin[bin_of(v)];
                if (sl != 0xFF) lists[sl * QBIG_CAP + atomicAdd(&list_count[sl], 1)] = v;
            }
        }
        __syncthreads();
        for (int k = wave; k < n_ranks; k += BLOCK / 64) {
            const int sl = rank_slot[k];
            if (sl >= 0) select_in_list(lists + sl * QBIG_CAP, list_count[sl], rank_local[k], k, lane, 64);
        }
        __syncthreads();
        // ---- refinement of the targets whose bin did not fit ----
        for (int k = 0; k < n_ranks; ++k) {
            if (rank_slot[k] != -2) continue;                // workgroup-uniform (LDS word)
            // the bin's values: the finite v with bin_of(v) == rank_bin[k]; from the second round on: lo <= v <= hi
            int want = rank_local[k];
            const int b0 = rank_bin[k];
            double lo = d_inf(), hi = -d_inf();
            QBIG_FOR_VALUES(v) {
                if (v > -d_inf() && v < d_inf() && bin_of(v) == b0) {
                    lo = fmin(lo, v);
                    hi = fmax(hi, v);
                }
            }
            block_minmax(lo, hi);
            for (int round = 0; round < 64; ++round) {       // (each round splits the interval 4 096 ways: a few at most)
                if (!(hi > lo)) {
                    if (tid == 0) result[k] = lo;
                    break;
                }
                for (int i = tid; i < QBIG_BINS; i += BLOCK) hist[i] = 0;
                __syncthreads();
                const double inv2 = (double)QBIG_BINS / (hi - lo);
                auto bin2 = [&](double v) { return min(QBIG_BINS - 1, (int)((v - lo) * inv2)); };
                QBIG_FOR_VALUES(v) {
                    if (v >= lo && v <= hi) atomicAdd(&hist[bin2(v)], 1);
                }
                __syncthreads();
                const int total = hist[QBIG_BINS - 1];       // (read before the prefix overwrites it)
                __syncthreads();
                prefix_hist();
                const int b2 = find_bin(want);
                const int cnt2 = ((b2 + 1 < QBIG_BINS) ? hist[b2 + 1] : hist[QBIG_BINS - 1] + total) - hist[b2];
                const int want2 = want - hist[b2];
                __syncthreads();
                if (cnt2 <= QBIG_POOL) {
                    if (tid == 0) list_count[0] = 0;
                    __syncthreads();
                    QBIG_FOR_VALUES(v) {
                        if (v >= lo && v <= hi && bin2(v) == b2) lists[atomicAdd(&list_count[0], 1)] = v;
                    }
                    __syncthreads();
                    select_in_list(lists, list_count[0], want2, k, tid, BLOCK);
                    break;
                }
                double lo2 = d_inf(), hi2 = -d_inf();
                QBIG_FOR_VALUES(v) {
                    if (v >= lo && v <= hi && bin2(v) == b2) {
                        lo2 = fmin(lo2, v);
                        hi2 = fmax(hi2, v);
                    }
                }
                block_minmax(lo2, hi2);
                lo = lo2;
                hi = hi2;
                want = want2;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid < A.nq) {
        int i0, i1;
        double f;
        quantile_position(A.q[tid], m, i0, i1, f);
        A.out[(e * A.D + d) * A.nq + tid] = quantile_lerp(result[2 * tid], result[2 * tid + 1], f);
    }
#undef QBIG_FOR_VALUES
}


// -------------------------------------------------------------------------------------------
// Selection with ONE WAVEFRONT per (ensemble, parameter) pair, the values held in registers.
// The workgroup kernel above spends its time in barriers between short phases (~60 us per pair for 3 200
// values, 1 280 pairs in flight): a wave needs no barrier at all, keeps its <= 64 x IPL values in VGPRs after a
// single pass over the chain, and 16 waves per CU are in flight.  Same algorithm: min / max, 1 024-bin
// histogram in the wave's own LDS, prefix, the (at most 2 nq) target bins gathered into one small pool, rank by
// counting.  A pair whose target bins overflow the pool (hundreds of identical values) or whose values are
// not finite is flagged in `out` and left to the workgroup kernel, launched right after with only_flagged = 1.
// -------------------------------------------------------------------------------------------
constexpr int QW_IPL = 52;          // values per lane: up to 3 328 per pair (32 walkers x 100 steps = 3 200)
constexpr int QW_IPL_BIG = 104;     // second instantiation: up to 6 656 per pair at two waves per SIMD
constexpr int QW_POOL = 448;        // doubles of gathered target-bin elements per wave
constexpr int QW_WAVES = BLOCK / 64;
// per-wave LDS: histogram / slot map, pool, 16 x {bin, local rank, slot}, 16 x {offset, count, fill}, 16 results
constexpr int QW_LDS_PER_WAVE = QSEL_BINS * 4 + QW_POOL * 8 + 6 * QSEL_RANKS * 4 + QSEL_RANKS * 8;

__device__ __forceinline__ void qw_sync()
{
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
}

// (capping the registers at 128 for a fourth wave per SIMD spills 106 of them: measured 22 instead of 16.5 ns per pair)
template <int IPL>
__global__ __launch_bounds__(BLOCK, 2) void k_chain_quantiles_wave(const QuantArgs A)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pair = (int64_t)blockIdx.x * QW_WAVES + wave;
    if (pair >= A.n_ens * A.D) return;                              // wave-uniform; no workgroup barrier below
    char* base = reinterpret_cast<char*>(lds) + (size_t)wave * QW_LDS_PER_WAVE;
    double* pool = reinterpret_cast<double*>(base);
    double* result = pool + QW_POOL;
    int* hist = reinterpret_cast<int*>(result + QSEL_RANKS);
    int* rank_bin = hist + QSEL_BINS;
    int* rank_local = rank_bin + QSEL_RANKS;
    int* rank_slot = rank_local + QSEL_RANKS;
    int* list_off = rank_slot + QSEL_RANKS;
    int* list_cnt = list_off + QSEL_RANKS;
    int* list_fill = list_cnt + QSEL_RANKS;

    const int64_t e = pair / A.D;
    const int d = (int)(pair - e * A.D);
    const int m = (int)(A.nsteps * A.W);
    const int n_ranks = 2 * A.nq;

    // ---- the only pass over the chain: value i of the pair lives in lane i % 64, register i / 64 ----
    double v[IPL];
    double mn = d_inf(), mx = -d_inf();
    {
        // i = t W + w advances by 64 per register: (t, w) += (64 / W, 64 % W) with a carry, no division in the loop
        const int dq = 64 / A.W, dr = 64 - dq * A.W;
        int t = lane / A.W, w = lane - t * A.W;
        const double* __restrict__ src = A.chain + (e * A.W) * A.rs + d * A.ps;
        const int64_t step_stride = A.ss;
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            const bool have = k * 64 + lane < m;
            // unconditional load (element 0 of the pair for the padding lanes): the 52 loads issue back to back
            double x = src[have ? (int64_t)t * step_stride + w * A.rs : 0];
            x = (have && x == x) ? x : d_inf();                     // NaN sorts last, as chain_value
            v[k] = x;
            mn = fmin(mn, x);
            mx = have ? fmax(mx, x) : mx;
            t += dq;
            w += dr;
            if (w >= A.W) {
                w -= A.W;
                ++t;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, off));
        mx = fmax(mx, __shfl_xor(mx, off));
    }
    const double inv = (double)QSEL_BINS / (mx - mn);
    const bool degenerate = !(mx > mn) || !isfinite(inv) || !isfinite(mn);
    if (degenerate) {
        const bool flat = !(mx > mn) && isfinite(mn) && isfinite(mx);        // every value equal
        if (lane < A.nq) A.out[pair * A.nq + lane] = flat ? mn : __longlong_as_double((long long)QUANT_FLAG);
        return;
    }
    auto bin_of = [&](double x) { return min(QSEL_BINS - 1, (int)((x - mn) * inv)); };

    // ---- histogram in the wave's LDS, exclusive prefix (16 bins per lane) ----
#pragma unroll
    for (int r = 0; r < QSEL_BINS / 64; ++r) hist[r * 64 + lane] = 0;
    if (lane < QSEL_RANKS) list_fill[lane] = 0;
    qw_sync();
#pragma unroll
    for (int k = 0; k < IPL; ++k)
        if (k * 64 + lane < m) atomicAdd(&hist[bin_of(v[k])], 1);
    qw_sync();
    {
        constexpr int PER = QSEL_BINS / 64;
        int c[PER];
        int tot = 0;
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            c[r] = hist[lane * PER + r];
            tot += c[r];
        }
        int incl = tot;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        int run = incl - tot;
        qw_sync();
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            hist[lane * PER + r] = run;                            // exclusive prefix
            run += c[r];
        }
    }
    qw_sync();
    // ---- target ranks -> bin and rank inside the bin ----
    if (lane < n_ranks) {
        int i0, i1;
        double f;
        quantile_position(A.q[lane >> 1], m, i0, i1, f);
        const int r = (lane & 1) ? i1 : i0;
        int lo = 0, hi = QSEL_BINS - 1;                            // last bin whose exclusive prefix is <= r
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (hist[mid] <= r) lo = mid;
            else hi = mid - 1;
        }
        rank_bin[lane] = lo;
        rank_local[lane] = r - hist[lo];
    }
    qw_sync();
    // ---- distinct target bins -> slots of the pool (lane 0; at most 16 short iterations) ----
    int overflow = 0, n_lists = 0;
    if (lane == 0) {
        int used = 0;
        for (int k = 0; k < n_ranks; ++k) {
            const int b = rank_bin[k];
            int sl = -1;
            for (int j = 0; j < k; ++j)
                if (rank_bin[j] == b) { sl = rank_slot[j]; break; }
            if (sl < 0) {
                const int cnt = ((b + 1 < QSEL_BINS) ? hist[b + 1] : m) - hist[b];
                sl = n_lists++;
                list_off[sl] = used;
                list_cnt[sl] = cnt;
                used += cnt;
            }
            rank_slot[k] = sl;
        }
        overflow = used > QW_POOL;
    }
    overflow = __shfl(overflow, 0);
    n_lists = __shfl(n_lists, 0);
    if (overflow) {                                                  // wave-uniform
        if (lane < A.nq) A.out[pair * A.nq + lane] = __longlong_as_double((long long)QUANT_FLAG);
        return;
    }
    qw_sync();
    // ---- the histogram area becomes the bin -> slot map ----
#pragma unroll
    for (int r = 0; r < QSEL_BINS / 64; ++r) hist[r * 64 + lane] = -1;
    qw_sync();
    if (lane < n_ranks) hist[rank_bin[lane]] = rank_slot[lane];      // equal bins write equal slots
    qw_sync();
    // ---- gather the target bins' elements into the pool ----
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        if (k * 64 + lane < m) {
            const int sl = hist[bin_of(v[k])];
            if (sl >= 0) {
                const int pos = atomicAdd(&list_fill[sl], 1);
                pool[list_off[sl] + pos] = v[k];
            }
        }
    }
    qw_sync();
    // ---- order statistic inside a list by counting ----
    for (int k = 0; k < n_ranks; ++k) {
        const int sl = rank_slot[k], cnt = list_cnt[sl], want = rank_local[k];
        const double* L = pool + list_off[sl];
        for (int j = lane; j < cnt; j += 64) {
            const double x = L[j];
            int before = 0;
            for (int i = 0; i < cnt; ++i) {
                const double y = L[i];
                before += (y < x || (y == x && i < j)) ? 1 : 0;
            }
            if (before == want) result[k] = x;
        }
    }
    qw_sync();
    if (lane < A.nq) {
        int i0, i1;
        double f;
        quantile_position(A.q[lane], m, i0, i1, f);
        A.out[pair * A.nq + lane] = quantile_lerp(result[2 * lane], result[2 * lane + 1], f);
    }
}


// -------------------------------------------------------------------------------------------
// The same wave-per-pair selection for the common chain shapes, with everything the generic form decides per value
// decided at compile time.  FULL = floor(m / 64) registers are full in every lane (TAIL: one more register holds the
// remaining m % 64 values), and W divides 64, so value (t, w) of register k is dq = 64 / W steps behind register k + 1's:
// one per-lane pointer, a uniform advance per register, no predicate on the FULL registers anywhere.  The generic
// kernel spends 24 of its ~73 vector instructions per value on 64-bit index arithmetic for the load alone
// (profiles/r02: 3 806 per pair of 3 200 values, instruction-issue bound); this form needs ~20 per value in total.
// A NaN in the chain (cannot occur in an accepted chain) flags the pair for the workgroup kernel instead of being
// replaced value by value.  Host dispatch: iso_chain_quantiles_layout, sizes 12 / 25 / 50 / 100 x 64 (+ tail).
// -------------------------------------------------------------------------------------------
// registers: the values themselves are 2 x FULL; everything else must fit in what is left for 4 (FULL <= 50) or 2 waves per SIMD
#ifndef ISO_QEXACT_WAVES_50
#define ISO_QEXACT_WAVES_50 4
#endif
constexpr int qexact_waves(int full) { return full <= 25 ? 4 : full <= 50 ? ISO_QEXACT_WAVES_50 : 2; }
// LEAN (default): fewer fp64 instructions per value - the kernel is bound by vector instruction issue (4 waves per SIMD x
// ~2 100 instructions per pair against 25.6 KB of chain).  Any monotone binning selects the same order statistics, so
//   * the bin of x is (int) fma(x, inv, -mn * inv) clamped to [0, BINS - 1] by one v_med3_i32: two fp64 instructions instead
//     of three (subtract, multiply, convert), in the histogram pass and again in the gather pass;
//   * the range [mn, mx] the bins divide comes from every fourth register (a quarter of the values): what lies outside it
//     falls into the two edge bins (a handful of values), and two of the three fp64 instructions of the load pass go for
//     three registers out of four.  A pair whose sampled range is empty is flagged for the workgroup kernel (it may still
//     hold different values elsewhere); the wave-uniform "every value equal" answer needs the full range and stays with
//     the ISO_QEXACT_LEAN=0 form and the generic wave kernel.
// Measured (tools/quantile_timing.py, this form against the previous library, interleaved; profiles/r04/quantile_lean_ab.txt,
// quantile_order_ab.txt): 10^5 stars x 3 200 values 3.27-3.31 -> 3.21-3.25 ms, 6 400 values 3.89 -> 3.79, 3 232 values (tail)
// 3.50 -> 3.35 - 1-4 %, together with the scalar-base loads below.  ISO_QEXACT_SUBRANGE=1 (range from a quarter of the values)
// is not the default: the compiler then fetches the other three quarters in a second round trip behind the test of the
// range, and holding the loads in place costs 228-384 B of scratch per lane.  Also measured: consecutive waves on
// consecutive ensembles of one parameter (contiguous 1-KB runs per workgroup and step instead of four runs 25 MB apart):
// 3.22-3.25 against 3.21-3.24 ms - no difference, not kept.  The kernel is bound by neither address arithmetic nor DRAM
// locality; what is left is the length of a pair's dependent phases at four waves per SIMD.
#ifndef ISO_QEXACT_LEAN
#define ISO_QEXACT_LEAN 1
#endif
#ifndef ISO_QEXACT_SUBRANGE
#define ISO_QEXACT_SUBRANGE 0
#endif
// v_min_f64 / v_max_f64 as they are: fmin() / fmax() first canonicalise every operand that comes from memory (one more
// v_max_f64 x, x per value: 74 of the kernel's ~1 370 vector instructions) so that a signalling NaN is quieted.  Here a NaN of
// either kind only has to reach the "not finite" exit: a quiet one is skipped by the instruction and caught by the x != x test
// next to it, a signalling one makes the range NaN.
__device__ __forceinline__ double qmin_raw(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double qmax_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// the hardware's double -> int conversion, named directly: it saturates and gives 0 for NaN, where a C++ cast is undefined
__device__ __forceinline__ int qcvt_i32(double v)
{
#ifdef ISO_QCVT_CAST
    return (int)v;
#else
    int r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(v));
    return r;
#endif
}

template <int FULL, bool TAIL>
__global__ __launch_bounds__(BLOCK, qexact_waves(FULL)) void k_chain_quantiles_exact(const QuantArgs A)
{
    extern __shared__ double lds[];
    // (the wave index through readfirstlane: everything derived from the pair - its chain base above all - is then scalar)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t pair = (int64_t)blockIdx.x * QW_WAVES + wave;
    if (pair >= A.n_ens * A.D) return;                              // wave-uniform; no workgroup barrier below
    char* base = reinterpret_cast<char*>(lds) + (size_t)wave * QW_LDS_PER_WAVE;
    double* pool = reinterpret_cast<double*>(base);
    double* result = pool + QW_POOL;
    int* hist = reinterpret_cast<int*>(result + QSEL_RANKS);
    int* rank_bin = hist + QSEL_BINS;
    int* rank_local = rank_bin + QSEL_RANKS;
    int* rank_slot = rank_local + QSEL_RANKS;
    int* list_off = rank_slot + QSEL_RANKS;
    int* list_cnt = list_off + QSEL_RANKS;
    int* list_fill = list_cnt + QSEL_RANKS;

    const int64_t e = pair / A.D;
    const int d = (int)(pair - e * A.D);
    const int m = (int)(A.nsteps * A.W);                            // FULL * 64 (+ m % 64 with TAIL)
    const int n_ranks = 2 * A.nq;
    const int tail = m - FULL * 64;

    // ---- the only pass over the chain ----
    double v[FULL], vt = d_inf();
    double mn = d_inf(), mx = -d_inf();
    bool any_nan = false;
    {
        const int dq = 64 / A.W;
        const int t0 = lane / A.W, w0 = lane - t0 * A.W;
        // address = scalar base of the pair's register k + the lane's own 32-bit byte offset (the host sends a chain whose
        // lane offsets do not fit 32 bits to the generic wave kernel): the load takes its base from a scalar register pair
        // and no vector instruction computes an address - 50 64-bit vector additions, and as many register pairs, less
        const char* __restrict__ sbase = reinterpret_cast<const char*>(A.chain + (e * A.W) * A.rs + d * A.ps);
        const uint32_t loff = (uint32_t)(((int64_t)t0 * A.ss + (int64_t)w0 * A.rs) * 8);
        const int64_t adv = (int64_t)dq * A.ss * 8;
        // (the scalar base of register k is made opaque: otherwise the lane offset is folded into one 64-bit vector base and
        // every load gets a vector address again)
        auto at = [&](int k) {
            typedef const char __attribute__((address_space(1))) gchar;         // (global: an opaque generic pointer would
            typedef const double __attribute__((address_space(1))) gdouble;     //  make these flat loads)
            gchar* b = (gchar*)(sbase + k * adv);
            asm volatile("" : "+s"(b));
            return *(gdouble*)(b + loff);
        };
#pragma unroll
        for (int k = 0; k < FULL; ++k) {
            const double x = at(k);
            v[k] = x;
            any_nan |= (x != x);
            if (!ISO_QEXACT_SUBRANGE) {
                mn = qmin_raw(mn, x);
                mx = qmax_raw(mx, x);
            }
        }
        if (ISO_QEXACT_SUBRANGE) {
            // every load is issued before the range is reduced: the values the range does not need must not be fetched in a
            // second round trip behind the test of the range (which is where the compiler puts them when left alone)
#pragma unroll
            for (int k = 0; k < FULL; ++k) asm volatile("" : "+v"(v[k]));
#pragma unroll
            for (int k = 0; k < FULL; k += 4) {
                mn = qmin_raw(mn, v[k]);
                mx = qmax_raw(mx, v[k]);
            }
        }
        if (TAIL) {
            const bool have = lane < tail;
            const double x = have ? at(FULL) : d_inf();
            vt = x;
            any_nan |= (x != x);
            if (!ISO_QEXACT_SUBRANGE) {
                mn = qmin_raw(mn, x);
                mx = have ? qmax_raw(mx, x) : mx;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, off));
        mx = fmax(mx, __shfl_xor(mx, off));
    }
    const double inv = (double)QSEL_BINS / (mx - mn);
    const bool degenerate = !(mx > mn) || !isfinite(inv) || !isfinite(mn) || __any(any_nan);
    if (degenerate) {
        const bool flat = !ISO_QEXACT_SUBRANGE && !(mx > mn) && isfinite(mn) && isfinite(mx) && !__any(any_nan);      // every value equal
        if (lane < A.nq) A.out[pair * A.nq + lane] = flat ? mn : __longlong_as_double((long long)QUANT_FLAG);
        return;
    }
#if ISO_QEXACT_LEAN
    const double off = -mn * inv;
    auto bin_of = [&](double x) { return min(max(qcvt_i32(fma(x, inv, off)), 0), QSEL_BINS - 1); };
#else
    auto bin_of = [&](double x) { return min(QSEL_BINS - 1, (int)((x - mn) * inv)); };
#endif

    // ---- histogram in the wave's LDS, exclusive prefix (16 bins per lane) ----
#pragma unroll
    for (int r = 0; r < QSEL_BINS / 64; ++r) hist[r * 64 + lane] = 0;
    if (lane < QSEL_RANKS) list_fill[lane] = 0;
    qw_sync();
#pragma unroll
    for (int k = 0; k < FULL; ++k) atomicAdd(&hist[bin_of(v[k])], 1);
    if (TAIL && lane < tail) atomicAdd(&hist[bin_of(vt)], 1);
    qw_sync();
    {
        constexpr int PER = QSEL_BINS / 64;
        int c[PER];
        int tot = 0;
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            c[r] = hist[lane * PER + r];
            tot += c[r];
        }
        int incl = tot;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        int run = incl - tot;
        qw_sync();
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            hist[lane * PER + r] = run;                            // exclusive prefix
            run += c[r];
        }
    }
    qw_sync();
    if (lane < n_ranks) {
        int i0, i1;
        double f;
        quantile_position(A.q[lane >> 1], m, i0, i1, f);
        const int r = (lane & 1) ? i1 : i0;
        int lo = 0, hi = QSEL_BINS - 1;                            // last bin whose exclusive prefix is <= r
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (hist[mid] <= r) lo = mid;
            else hi = mid - 1;
        }
        rank_bin[lane] = lo;
        rank_local[lane] = r - hist[lo];
    }
    qw_sync();
    int overflow = 0, n_lists = 0;
    if (lane == 0) {
        int used = 0;
        for (int k = 0; k < n_ranks; ++k) {
            const int b = rank_bin[k];
            int sl = -1;
            for (int j = 0; j < k; ++j)
                if (rank_bin[j] == b) { sl = rank_slot[j]; break; }
            if (sl < 0) {
                const int cnt = ((b + 1 < QSEL_BINS) ? hist[b + 1] : m) - hist[b];
                sl = n_lists++;
                list_off[sl] = used;
                list_cnt[sl] = cnt;
                used += cnt;
            }
            rank_slot[k] = sl;
        }
        overflow = used > QW_POOL;
    }
    overflow = __shfl(overflow, 0);
    if (overflow) {                                                  // wave-uniform
        if (lane < A.nq) A.out[pair * A.nq + lane] = __longlong_as_double((long long)QUANT_FLAG);
        return;
    }
    qw_sync();
#pragma unroll
    for (int r = 0; r < QSEL_BINS / 64; ++r) hist[r * 64 + lane] = -1;       // the histogram area becomes the bin -> slot map
    qw_sync();
    if (lane < n_ranks) hist[rank_bin[lane]] = rank_slot[lane];      // equal bins write equal slots
    qw_sync();
    // the bins are computed again here, not carried over from the histogram pass (the compiler would keep all FULL of them
    // alive across the phases in between: 50-100 more registers, one or two waves per SIMD fewer): hide the scale from it
#if ISO_QEXACT_LEAN
    double mn2 = off, inv2 = inv;
    asm volatile("" : "+v"(mn2), "+v"(inv2));
    auto bin2 = [&](double x) { return min(max(qcvt_i32(fma(x, inv2, mn2)), 0), QSEL_BINS - 1); };
#else
    double mn2 = mn, inv2 = inv;
    asm volatile("" : "+v"(mn2), "+v"(inv2));
    auto bin2 = [&](double x) { return min(QSEL_BINS - 1, (int)((x - mn2) * inv2)); };
#endif
    auto gather = [&](double x) {
        const int sl = hist[bin2(x)];
        if (sl >= 0) {
            const int pos = atomicAdd(&list_fill[sl], 1);
            pool[list_off[sl] + pos] = x;
        }
    };
#pragma unroll
    for (int k = 0; k < FULL; ++k) gather(v[k]);
    if (TAIL && lane < tail) gather(vt);
    qw_sync();
    for (int k = 0; k < n_ranks; ++k) {
        const int sl = rank_slot[k], cnt = list_cnt[sl], want = rank_local[k];
        const double* L = pool + list_off[sl];
        for (int j = lane; j < cnt; j += 64) {
            const double x = L[j];
            int before = 0;
            for (int i = 0; i < cnt; ++i) {
                const double y = L[i];
                before += (y < x || (y == x && i < j)) ? 1 : 0;
            }
            if (before == want) result[k] = x;
        }
    }
    qw_sync();
    if (lane < A.nq) {
        int i0, i1;
        double f;
        quantile_position(A.q[lane], m, i0, i1, f);
        A.out[pair * A.nq + lane] = quantile_lerp(result[2 * lane], result[2 * lane + 1], f);
    }
}
