// kernel launch switches over the band count
// (part of iso_fast_kernel.h: included inside namespace iso::fastk)
//
// Every launcher names the instantiation it chose (note_kernel: a no-op unless iso_debug_trace_kernels(1) was called) in
// the spelling c++filt gives the kernel's symbol, so that tests/test_gpu_dispatch_table.py can tick off every kernel the
// build compiled (libiso_hip.resources.json) against the oracle.
#pragma once

// dynamic LDS of the persistent form; the host uses it to decide whether an ensemble fits
inline size_t stretch_persist_lds_bytes(int axes_len, int nb, int W, int np, bool dense = false)
{
    return (size_t)(((axes_len + 1) & ~1) + BLOCK * persist_slot_stride(dense, nb, np - 4) +
                    persist_extra_doubles(W, np, persist_slim(dense, nb, np - 4))) * sizeof(double);
}

// The persistent instantiation a run takes - ONE place decides it, for the occupancy query and for the launch alike
// (round 3 queried the catalog form even when the single-model forms, with other register counts, were launched):
//   dense                     register-capped: 4 (single stars, <= 6 bands) or 3 waves per SIMD, catalogs / many ensembles in rounds;
//                             + STDP (3 waves per SIMD) when the stars share the default priors and the launch runs in rounds anyway
//   multi                     catalog, every workgroup resident: uncapped registers; + STDP when the stars share the
//                             reference's default priors: shared block through scalar loads, compile-time families,
//                             lane BC gather (one band), table-free priors during the model gather
//   single model, std priors  UNI + STDP: model block through scalar loads, prior families compile-time constants
//   single model              UNI
//   single binary, std priors k_stretch_pair: one star per lane, when a half-step's moves fit half a workgroup
//   single triple, std priors k_stretch_triple: one star per row of a wave, up to 64 moves per half-step
// Models with asteroseismic terms (ASTERO) are single models by construction (iso_catalog_create refuses them) and
// have no register-capped form: a run of very many ensembles of such a model takes the UNI kernel in rounds.
struct PersistKernel {
    const void* fn;
    bool dense;
    char name[96];
};

template <int KIND, int NS, int N, bool ASTERO>
inline PersistKernel persist_kernel(const StretchArgs& S)
{
    PersistKernel k;
    bool dense = S.dense != 0, uni = false, stdp = false;
    if constexpr (ASTERO) {
        // one form: single model, priors read at run time (the compile-time-prior twin bought 13 % of a half-step and doubled
        // the asteroseismic instantiations: pruned in round 5, with the step-wise asteroseismic kernels)
        dense = false;
        uni = true;
        stdp = false;
        k.fn = (const void*)k_stretch_persist<KIND, NS, N, false, true, true, false>;
    } else if (dense) {
        // two register-capped forms: prior families read at run time (four waves per SIMD for single stars up to six bands),
        // or - the stars share the reference's default priors and the host asks for it (StretchArgs.dense_stdp) - compiled in,
        // at the registers of three waves per SIMD (the straight-line priors need them: 136 B of scratch at the cap of four)
        // (single stars with up to six bands only: the other shapes sit at three waves per SIMD either way, and there the
        // compiled-in form only adds scratch - 400-480 B per lane for systems with 11-12 bands)
        // (round 6) single stars, ONE ensemble per workgroup (258 and more walkers - the reference's default 300 -, or a launch
        // the host spreads one ensemble per workgroup): the star's block through scalar loads (DENSE + UNI, sampler.h)
        const int GL = persist_group(S.W);
        const bool one = NS == 1 && ((S.group > 0 && S.group < GL) ? S.group : GL) == 1;
        if constexpr (NS == 1) {
            if constexpr (persist_slim(true, N, NS)) {
                stdp = S.std_priors != 0 && S.dense_stdp != 0;
                if (one) k.fn = stdp ? (const void*)k_stretch_persist<KIND, NS, N, true, false, true, true>
                                     : (const void*)k_stretch_persist<KIND, NS, N, true, false, true, false>;
                else k.fn = stdp ? (const void*)k_stretch_persist<KIND, NS, N, true, false, false, true>
                                 : (const void*)k_stretch_persist<KIND, NS, N, true, false>;
            } else {
                if (one) k.fn = (const void*)k_stretch_persist<KIND, NS, N, true, false, true, false>;
                else k.fn = (const void*)k_stretch_persist<KIND, NS, N, true, false>;
            }
        } else {
            k.fn = (const void*)k_stretch_persist<KIND, NS, N, true, false>;      // (systems: at three waves per SIMD either way - persist_slim)
        }
        uni = one;
    } else if (S.multi) {
        // resident catalog: when its stars share their priors and those are the reference's defaults, the form that reads
        // them through scalar loads with the families as compile-time constants (STDP without UNI)
        stdp = S.std_priors != 0;
        if constexpr (N == 0) k.fn = nullptr;          // a catalog has 1-12 bands (iso_catalog_create)
        else k.fn = stdp ? (const void*)k_stretch_persist<KIND, NS, N, false, false, false, true>
                         : (const void*)k_stretch_persist<KIND, NS, N, false, false, false, false>;
    } else {
        uni = true;
        stdp = S.std_priors != 0;
        if constexpr (KIND == ISO_KIND_ISO && NS == 2) {
            // a single binary whose half-step fits half a workgroup: one star per lane (k_stretch_pair, sampler.h)
            const int h = S.W >> 1;
            const int64_t n_ens = S.n_active / h;
            const int GL = persist_group(S.W);
            const int G = (S.group > 0 && S.group < GL) ? S.group : GL;
            // measured (profiles/r04/pair_kernel_ab.jsonl, us per step): up to 16 moves per wave 13.2 -> 10.0 (1 band), 21.7 ->
            // 19.4 (12 bands); 17-32 moves per wave 14.4 -> 13.4 (1 band), nothing beyond 4 bands
            const int64_t moves = (n_ens < G ? n_ens : G) * h;
            // (round 6: the reference's default prior families only - the run-time-prior twins of the one-star-per-lane /
            // -per-row kernels were 39 instantiations for single fits with non-default priors, which keep the plain form)
            if (S.pair && stdp && moves <= BLOCK / 2 && (moves <= BLOCK / 4 || N <= 4)) {
                k.fn = (const void*)k_stretch_pair<N>;
                k.dense = false;
                snprintf(k.name, sizeof k.name, "k_stretch_pair<%d>", N);
                return k;
            }
        }
        if constexpr (KIND == ISO_KIND_ISO && NS == 3) {
            // a single triple (or a few ensembles of it - every workgroup resident at once): one star per row of a wave
            // (k_stretch_triple, sampler.h); a run of more workgroups than the chip holds is throughput-bound and keeps the
            // form in which a lane walks its three stars
            const int64_t n_ens = S.n_active / (S.W >> 1);
            const int GL = persist_group(S.W);
            const int G = (S.group > 0 && S.group < GL) ? S.group : GL;
            const int64_t moves = (n_ens < G ? n_ens : G) * (S.W >> 1);
            if (S.pair && stdp && moves <= S.triple_moves && (n_ens + G - 1) / G <= 512) {
                k.fn = (const void*)k_stretch_triple<N>;
                k.dense = false;
                snprintf(k.name, sizeof k.name, "k_stretch_triple<%d>", N);
                return k;
            }
        }
        k.fn = stdp ? (const void*)k_stretch_persist<KIND, NS, N, false, false, true, true>
                    : (const void*)k_stretch_persist<KIND, NS, N, false, false, true, false>;
    }
    k.dense = dense;
    snprintf(k.name, sizeof k.name, "k_stretch_persist<%d, %d, %d, %s, %s, %s, %s>", KIND, NS, N, tf(dense), tf(ASTERO), tf(uni), tf(stdp));
    return k;
}

template <int KIND, int NS, bool ASTERO = false>
inline bool launch_stretch_nb(int nb, const FastArgs& A, const StretchArgs& S, hipStream_t s)
{
    const dim3 b(BLOCK);
    if (S.nsteps > 0) {                                       // persistent: workgroups own whole ensembles
        const int64_t n_ens = S.n_active / (S.W >> 1);
        const int GL = persist_group(S.W);
        const int G = (S.group > 0 && S.group < GL) ? S.group : GL;
        const dim3 gp((unsigned)((n_ens + G - 1) / G));
        PersistKernel k;
        switch (nb) {
#define ISO_PERSIST_CASE(N) case N: k = persist_kernel<KIND, NS, N, ASTERO>(S); break;
            ISO_PERSIST_CASE(0) ISO_PERSIST_CASE(1) ISO_PERSIST_CASE(2) ISO_PERSIST_CASE(3) ISO_PERSIST_CASE(4) ISO_PERSIST_CASE(5)
            ISO_PERSIST_CASE(6) ISO_PERSIST_CASE(7) ISO_PERSIST_CASE(8) ISO_PERSIST_CASE(9) ISO_PERSIST_CASE(10)
            ISO_PERSIST_CASE(11) ISO_PERSIST_CASE(12)
#undef ISO_PERSIST_CASE
        default: return false;
        }
        if (!k.fn) return false;
        const size_t lds_bytes = stretch_persist_lds_bytes(A.axes_len, nb, S.W, NS + 4, k.dense);
        // with S.occupancy_query set: report resident workgroups per CU of this instantiation, launch nothing
        // (three-wave workgroups: only the register-capped form reads its thread count from the launch, sampler.h)
        const int threads = (k.dense && S.threads > 0) ? S.threads : BLOCK;
        // beyond the 64 KB a launch gets without asking (a 256-walker triple with nine bands: 67 KB; the CU has 160)
        if (lds_bytes > 64 * 1024 &&
            (lds_bytes > 160 * 1024 || hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess))
            return false;
        if (S.occupancy_query)
            return hipOccupancyMaxActiveBlocksPerMultiprocessor(S.occupancy_query, k.fn, threads, lds_bytes) == hipSuccess;
        note_kernel("%s", k.name);
        void* args[] = {const_cast<FastArgs*>(&A), const_cast<StretchArgs*>(&S)};
        return hipLaunchKernel(k.fn, gp, dim3(threads), args, lds_bytes, s) == hipSuccess;
    }
    // asteroseismic models: persistent form only (iso_sampler_create_model checks that the ensemble fits)
    if constexpr (ASTERO) {
        return false;
    } else {
    const dim3 g((unsigned)((S.n_active + BLOCK - 1) / BLOCK));
    auto sh = [&](int n) { return (size_t)(((A.axes_len + 1) & ~1) + coop_lds_doubles(n)) * sizeof(double); };
    switch (nb) {
#define ISO_HALF_CASE(N)                                                                              \
    case N:                                                                                           \
        note_kernel("k_stretch_half<%d, %d, %d, false>", KIND, NS, N);                                \
        hipLaunchKernelGGL((k_stretch_half<KIND, NS, N, false>), g, b, sh(N), s, A, S);               \
        return true;
        ISO_HALF_CASE(0) ISO_HALF_CASE(1) ISO_HALF_CASE(2) ISO_HALF_CASE(3) ISO_HALF_CASE(4) ISO_HALF_CASE(5) ISO_HALF_CASE(6)
        ISO_HALF_CASE(7) ISO_HALF_CASE(8) ISO_HALF_CASE(9) ISO_HALF_CASE(10) ISO_HALF_CASE(11) ISO_HALF_CASE(12)
#undef ISO_HALF_CASE
    default: return false;
    }
    }
}

template <int KIND, int NS, bool MULTI, bool ASTERO = false>
inline bool launch_nb(int nb, const FastArgs& A, hipStream_t s)
{
    const dim3 g((unsigned)((A.n + BLOCK - 1) / BLOCK)), b(BLOCK);
    auto sh = [&](int n) { return (size_t)(((A.axes_len + 1) & ~1) + coop_lds_doubles(n)) * sizeof(double); };
    switch (nb) {
#define ISO_FAST_CASE(N)                                                                              \
    case N:                                                                                           \
        if constexpr (MULTI && N == 0) return false;      /* a catalog has 1-12 bands (iso_catalog_create) */ \
        else {                                                                                        \
        note_kernel("k_lnpost_fast<%d, %d, %d, %s, %s>", KIND, NS, N, tf(MULTI), tf(ASTERO));         \
        hipLaunchKernelGGL((k_lnpost_fast<KIND, NS, N, MULTI, ASTERO>), g, b, sh(N), s, A);           \
        return true; }
        ISO_FAST_CASE(0) ISO_FAST_CASE(1) ISO_FAST_CASE(2) ISO_FAST_CASE(3) ISO_FAST_CASE(4) ISO_FAST_CASE(5) ISO_FAST_CASE(6)
        ISO_FAST_CASE(7) ISO_FAST_CASE(8) ISO_FAST_CASE(9) ISO_FAST_CASE(10) ISO_FAST_CASE(11) ISO_FAST_CASE(12)
#undef ISO_FAST_CASE
    default: return false;
    }
}
