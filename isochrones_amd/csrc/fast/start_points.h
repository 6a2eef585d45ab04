// start points of a catalog fit: draw, evaluate and select on the device, one workgroup per star
// (part of iso_fast_kernel.h: included inside namespace iso::fastk)
#pragma once

// -------------------------------------------------------------------------------------------
// The reference starts every star's ensemble from draws of its priors that have a finite posterior
// (isochrones/starmodel.py:903-949: sample_from_prior until nwalkers rows are valid) - one Python lnpost call per
// draw.  The catalog path drew oversample x W candidates per star inside the parameter bounds and kept each star's
// best W with a dozen framework passes (sort / topk / gather over [stars, candidates]) around one catalog-kernel
// launch: 26 ms of a 0.3-s fit at 4 x 10^5 stars, and most of what a GPU that is handed 1 250 stars of a
// 10^4-star catalog spends.  Here one workgroup owns one star:
//   * every lane draws one candidate per chunk of BLOCK (Philox4x32-10, counter = (4 chunk + call, star, lane, 0x57),
//     key = seed): uniform inside the star's bounds, log-uniform in mass, distance within 4 sigma of the parallax
//     distance when the star has a positive parallax (log-uniform otherwise), EEPs of a multiple system in descending
//     order - the candidate distribution of the framework version;
//   * the wave evaluates its 64 candidates with the catalog kernels' own lnpost_wave (same instantiation parameters
//     as k_lnpost_fast<..., MULTI>: the star's block, masked bands, shared priors);
//   * the best W so far live in LDS ([W][NP + 1] records, two buffers); a chunk is merged by rank-by-counting over
//     the union (a lane counts the records that beat its candidate: LDS broadcast reads, no sort, ties broken by
//     age - kept records first, then lane order - so the result does not depend on scheduling);
//   * chunks_min chunks always (oversample x W candidates), then more while fewer than W are finite, up to
//     chunks_max; a star that never gets there is flagged in `failed` (its rows are NaN) - the per-star failure
//     isolation of the reference's try / except around each star (isochrones/starfit.py:155-159).
// No candidate ever leaves the chip: the framework version moved 4 GB of them at 4 x 10^5 stars.
// -------------------------------------------------------------------------------------------
// (a workgroup merges a chunk of BLOCK candidates into the W records kept so far; ensembles of more than BLOCK walkers - the
// reference's default is 300, starmodel.py:889 - walk their kept records in strides of BLOCK)
constexpr int START_MAX_W = 1024;
constexpr double START_MAX_DISTANCE = 1.0e5;     // pc: candidates of a star without a usable parallax are drawn log-uniform below this

struct StartArgs {
    double* best;         // [n_stars][W][NP]
    double* best_lnp;     // [n_stars][W]
    int32_t* failed;      // [n_stars]
    int64_t n_stars;
    int W;
    int chunks_min, chunks_max;
    uint64_t seed;
};

__host__ __device__ constexpr int start_extra_doubles(int W, int np) { return 2 * BLOCK + 2 * W * (np + 1) + 2; }

__device__ __forceinline__ double uniform53(uint32_t hi, uint32_t lo)
{
    return ((double)hi * 2097152.0 + (double)(lo & 0x1FFFFFu)) * (1.0 / 9007199254740992.0);      // [0, 1)
}

// (the batch kernel's 6-wave cap for one or two bands would spill here: the candidate, its maps and the merge live next to
// the evaluation; 9 bands and more get the registers of 3 waves,
// systems of two and three stars with more than a few bands the registers of 2: a one-off kernel that runs a few chunks per
// star has nothing to gain from occupancy bought with scratch)
constexpr int start_min_waves(int ns, int nb) { return ns == 1 ? (nb >= 9 ? 3 : 4) : (nb <= 4 ? 3 : 2); }

template <int KIND, int NS, int NB>
__global__ __launch_bounds__(BLOCK, start_min_waves(NS, NB)) void k_catalog_start(const FastArgs A, const StartArgs T)
{
    extern __shared__ double lds[];
    for (int j = threadIdx.x; j < A.axes_len; j += BLOCK) lds[j] = A.axes_blob[j];
    constexpr int NP = NS + 4, REC = NP + 1;
    const CoopLds L = coop_lds<NB>(lds, A.axes_len);
    double* new_lnp = lds + ((A.axes_len + 1) & ~1) + coop_lds_doubles(NB);
    double* sorted_new = new_lnp + BLOCK;                  // the chunk's keys in order (best first)
    double* kept = sorted_new + BLOCK;                     // [2][W][REC]: parameters, then lnpost
    const int W = T.W, tid = (int)threadIdx.x;
    const int64_t star = blockIdx.x;
    const DevModel& M = A.m[star];
    const DevModel& MP = A.shared_priors ? A.m[0] : M;
    __syncthreads();
    // per-parameter maps u -> a + b u (mass and a prior-drawn distance are exponentiated).  The ranges are the star's bounds
    // CUT TO THE TABLES: a prior without finite bounds (a log-normal mass prior lives on (0, inf)) would otherwise give
    // log(0) / inf - inf maps, and a candidate outside the table has no posterior anyway.  Distances are drawn below
    // START_MAX_DISTANCE, extinctions inside the BC table's A_V axis (below 10 mag for a model without bands).
    double a[NP], b[NP];
    {
        double lo[NP], hi[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            lo[q] = M.bound_lo[q];
            hi[q] = M.bound_hi[q];
        }
        auto cut = [&](int q, double t_lo, double t_hi) {
            lo[q] = fmax(lo[q], t_lo);
            hi[q] = fmin(hi[q], t_hi);
        };
        const double ax0_lo = lds[A.m0.off], ax0_hi = lds[A.m0.off + A.m0.n - 1], ax1_lo = lds[A.m1.off], ax1_hi = lds[A.m1.off + A.m1.n - 1];
        if (KIND == ISO_KIND_TRACK) {            // (mass, eep, feh, ...): table axes (feh, mass, eep)
            cut(0, ax1_lo, ax1_hi);
            cut(1, A.e_a0, A.e_last);
            cut(2, ax0_lo, ax0_hi);
        } else {                                 // (eep[, eep_1[, eep_2]], age, feh, ...): table axes (age, feh, eep)
#pragma unroll
            for (int s = 0; s < NS; ++s) cut(s, A.e_a0, A.e_last);
            cut(NS, ax0_lo, ax0_hi);
            cut(NS + 1, ax1_lo, ax1_hi);
        }
        cut(NS + 2, 0.0, START_MAX_DISTANCE);
        if (NB > 0) cut(NS + 3, lds[A.b3.off], lds[A.b3.off + A.b3.n - 1]);
        else cut(NS + 3, 0.0, 10.0);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            a[q] = lo[q];
            b[q] = hi[q] - lo[q];
        }
        constexpr int QD0 = NS + 2;
        if (KIND == ISO_KIND_TRACK) {
            b[0] = log(hi[0] / lo[0]);
            a[0] = log(lo[0]);
        }
        const double plx0 = M.plx_val, d00 = 1000.0 / plx0;
        const bool use_plx0 = M.has_parallax && plx0 > 0.0 && isfinite(d00);
        const double dlo = fmax(lo[QD0], 1.0);
        const double rel = fmin(fmax(sqrt(M.plx_unc2) / plx0, 1e-3), 0.3);
        a[QD0] = use_plx0 ? d00 * (1.0 - 4.0 * rel) : log(dlo);
        b[QD0] = use_plx0 ? 8.0 * rel * d00 : log(hi[QD0] / dlo);
    }
    constexpr int QD = NS + 2;
    const double plx = M.plx_val, d0 = 1000.0 / plx;
    const bool use_plx = M.has_parallax && plx > 0.0 && isfinite(d0);
    int nkept = 0, cur = 0;
    for (int chunk = 0; chunk < T.chunks_max; ++chunk) {
        double p[NP];
#pragma unroll
        for (int c = 0; c < (NP + 1) / 2; ++c) {
            uint32_t r[4];
            philox4x32_10((uint32_t)(4 * chunk + c), (uint32_t)star, (uint32_t)tid | ((uint32_t)((uint64_t)star >> 32) << 16), 0x57u,
                          (uint32_t)T.seed, (uint32_t)(T.seed >> 32), r);
            p[2 * c] = fma(b[2 * c], uniform53(r[0], r[1]), a[2 * c]);
            if (2 * c + 1 < NP) p[2 * c + 1] = fma(b[2 * c + 1], uniform53(r[2], r[3]), a[2 * c + 1]);
        }
        if (KIND == ISO_KIND_TRACK) p[0] = exp(p[0]);
        if (!use_plx) p[QD] = exp(p[QD]);
        if (NS == 2) {
            const double hi = fmax(p[0], p[1]), lo = fmin(p[0], p[1]);
            p[0] = hi; p[1] = lo;
        }
        if (NS == 3) {
            const double x = fmax(p[0], p[1]), y = fmin(p[0], p[1]);
            const double top = fmax(x, p[2]), rest = fmin(x, p[2]);
            p[0] = top; p[1] = fmax(y, rest); p[2] = fmin(y, rest);
        }
        double lnp_unused, lnl_unused;
        const double r = lnpost_wave<KIND, NS, NB, false, true>(A, lds, L, true, M, MP, p, false, lnp_unused, lnl_unused);
        const double key = isfinite(r) ? r : -f_inf();
        new_lnp[tid] = key;
        __syncthreads();
        const double* kc = kept + cur * W * REC;
        double* kn = kept + (cur ^ 1) * W * REC;
        // records that beat this lane's candidate: kept records win ties (they are older), then the lower lane.
        // Among the chunk's own candidates by counting (BLOCK broadcast reads); against the kept list - sorted, best first - by
        // bisection: #{kept >= key} is where key would go (9 reads instead of W); and the chunk's keys, put in order by their
        // own ranks (a permutation: ties are broken), let every kept record find #{new > its key} the same way (8 reads
        // instead of BLOCK).  Same ranks as counting everything against everything, a quarter of the instructions of a merge.
        int nrank = 0;
        for (int f = 0; f < BLOCK; ++f) {
            const double v = new_lnp[f];
            nrank += (int)((v > key) | ((v == key) & (f < tid)));
        }
        sorted_new[nrank] = key;
        int rank = nrank;
        {
            int lo = 0, hi = nkept;                        // first index whose kept key is < key
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (kc[mid * REC + NP] >= key) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < W) {
#pragma unroll
            for (int q = 0; q < NP; ++q) kn[rank * REC + q] = p[q];
            kn[rank * REC + NP] = key;
        }
        __syncthreads();                                   // sorted_new complete
        for (int t = tid; t < nkept; t += BLOCK) {         // (one pass unless the ensemble has more than BLOCK walkers)
            const double kv = kc[t * REC + NP];
            int lo = 0, hi = BLOCK;                        // first index whose new key is <= kv: #{new > kv}
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sorted_new[mid] > kv) lo = mid + 1;
                else hi = mid;
            }
            const int krank = t + lo;                      // the kept list is sorted: t records of it are ahead already
            if (krank < W) {
#pragma unroll
                for (int q = 0; q < NP; ++q) kn[krank * REC + q] = kc[t * REC + q];
                kn[krank * REC + NP] = kv;
            }
        }
        __syncthreads();
        nkept = min(W, nkept + BLOCK);
        cur ^= 1;
        // workgroup-uniform: every lane reads the same word - the worst record kept
        if (chunk + 1 >= T.chunks_min && nkept == W && kn[(W - 1) * REC + NP] > -f_inf()) break;
    }
    const double* kc = kept + cur * W * REC;
    const bool ok = nkept == W && kc[(W - 1) * REC + NP] > -f_inf();
    double* __restrict__ ob = T.best + star * (int64_t)W * NP;
    for (int e = tid; e < W * NP; e += BLOCK) {
        const int w = e / NP, q = e - w * NP;
        ob[e] = ok ? kc[w * REC + q] : f_nan();
    }
    for (int w = tid; w < W; w += BLOCK) T.best_lnp[star * W + w] = (w < nkept) ? kc[w * REC + NP] : -f_inf();
    if (tid == 0) T.failed[star] = ok ? 0 : 1;
}

template <int KIND, int NS>
inline bool launch_start_nb(int nb, const FastArgs& A, const StartArgs& T, hipStream_t s)
{
    const dim3 g((unsigned)T.n_stars), b(BLOCK);
    auto sh = [&](int n) {
        return (size_t)(((A.axes_len + 1) & ~1) + coop_lds_doubles(n) + start_extra_doubles(T.W, NS + 4)) * sizeof(double);
    };
    switch (nb) {
#define ISO_START_CASE(N)                                                                             \
    case N:                                                                                           \
        if (sh(N) > 160 * 1024) return false;                                                         \
        if (sh(N) > 64 * 1024 &&                     /* beyond what a launch gets without asking */   \
            hipFuncSetAttribute((const void*)k_catalog_start<KIND, NS, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh(N)) != hipSuccess) \
            return false;                                                                             \
        note_kernel("k_catalog_start<%d, %d, %d>", KIND, NS, N);                                      \
        hipLaunchKernelGGL((k_catalog_start<KIND, NS, N>), g, b, sh(N), s, A, T);                     \
        return true;
        ISO_START_CASE(1) ISO_START_CASE(2) ISO_START_CASE(3) ISO_START_CASE(4) ISO_START_CASE(5) ISO_START_CASE(6)
        ISO_START_CASE(7) ISO_START_CASE(8) ISO_START_CASE(9) ISO_START_CASE(10) ISO_START_CASE(11) ISO_START_CASE(12)
#undef ISO_START_CASE
    default: return false;
    }
}
