// Resident mailbox kernel for the per-point callback: the reference is driven one lnpost(p) at a time by emcee / MultiNest
// (isochrones/starmodel.py:797,952,966), and a launch per call costs ~20 us of which the evaluation is ~4.  Here ONE wave
// stays resident on a CU for the model and polls a 64-byte request line in pinned, device-mapped host memory:
//
//   host                                            device (k_mailbox_lnpost, one wave)
//   req[1..NP] = parameters; req[0] = seq  ------>  one 64-B read over PCIe per poll: lanes 0-7 take the eight words of the
//                                                   line; a new sequence word = a request, whose parameters normally arrived
//                                                   with it (the host writes the sequence word LAST; x86 keeps the store
//                                                   order and a line is read from the host's cache in one piece).  Nothing
//                                                   in the memory model PROMISES the one piece: the sequence word carries a
//                                                   32-bit checksum of the parameter words, and a line that does not match
//                                                   it (new sequence word, stale parameters) is polled again
//                                                   evaluation: lnpost_wave, the batch kernel's device function (single-model
//                                                   form: model block through scalar loads, lane BC gather for one band)
//   spins on done[0] (its own cache)       <------  done[1..3] = lnpost, lnprior, lnlike; fence; done[0] = seq
//
// A call is a PCIe write, a PCIe read and a PCIe write instead of a launch: results are those of the launch path bit for bit
// (same device function, same instantiation flags as k_lnpost_fast except the gather form, which is bit-identical by
// construction - fast/lnpost_wave.h).  Requests of 2-128 rows put their rows behind the lines (`rows`), lane = row.
// The wave leaves when the host says so (`quit`: iso_model_destroy, a change of the model), after `idle_ticks` without a
// request (a device-wide synchronise elsewhere in the process - hipDeviceSynchronize, hipFree - waits for every kernel: it
// then waits at most that long), or after `life_ticks` whatever happens.  The host relaunches on the next call.
#include "iso_fast_kernel.h"

namespace iso {
namespace fastk {

__device__ __forceinline__ unsigned long long sys_load(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int KIND, int NS, int NB>
__global__ __launch_bounds__(64, 2) void k_mailbox_lnpost(const FastArgs A0, IsoMailbox* mb, unsigned long long idle_ticks,
                                                          unsigned long long life_ticks)
{
    extern __shared__ double lds[];
    for (int j = threadIdx.x; j < A0.axes_len; j += 64) lds[j] = A0.axes_blob[j];
    __syncthreads();
    const CoopLds L = coop_lds<NB>(lds, A0.axes_len);
    const int lane = (int)threadIdx.x;
    typedef const __attribute__((address_space(4))) DevModel* const_model_ptr;
    constexpr int NP = NS + 4;
    unsigned long long last = sys_load(&mb->done[0]);         // what the previous resident wave (or nobody) finished last
    const unsigned long long t_start = wall_clock64();
    unsigned long long t_idle = t_start;
    for (;;) {
        const unsigned long long w = sys_load(&mb->req[lane & 7]);
        const unsigned long long seq = __shfl(w, 0);
        if (seq == last) {
            const unsigned long long now = wall_clock64();
            const bool leave = (now - t_idle > idle_ticks) | (now - t_start > life_ticks) | (sys_load(&mb->ctl[1]) != 0);
            if (leave) break;                                  // (wave-uniform: every lane read the same words)
            continue;
        }
        // the argument block is read again from the kernel-argument segment for every request (scalar loads where a field is
        // used, through a pointer the optimiser cannot see through) instead of living in spilled scalar registers across the
        // polling loop - as the persistent samplers do (fast/sampler.h)
#ifdef ISO_MAILBOX_CLOCK
        const unsigned long long t_seen = wall_clock64();
#endif
        typedef const __attribute__((address_space(4))) char* kernarg_ptr;
        kernarg_ptr kp = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const FastArgs& A = *(const FastArgs*)(const __attribute__((address_space(4))) FastArgs*)kp;
        const DevModel& M = *(const DevModel*)((const_model_ptr)(uintptr_t)A.m);
        const int n = (int)(seq & 0xFF) + 1;                   // rows of the request (1..128)
        const bool parts = ((seq >> 8) & 1) != 0;              // lnprior / lnlike wanted as well
        if (n == 1) {
            unsigned long long words[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) words[q] = __shfl(w, 1 + q);
            if (mailbox_checksum(words, NP) != (uint32_t)(seq >> 32)) continue;      // torn line (wave-uniform): poll again
        } else {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // the rows behind the line are read after the sequence word
        }
        for (int r0 = 0; r0 < n; r0 += 64) {
            const int r = r0 + lane;
            const bool active = r < n;
            double p[NP];
            if (n == 1) {
#pragma unroll
                for (int q = 0; q < NP; ++q) p[q] = __longlong_as_double((long long)__shfl(w, 1 + q));
            } else {
                const int rr = active ? r : n - 1;
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    p[q] = __longlong_as_double((long long)sys_load(reinterpret_cast<const unsigned long long*>(&mb->rows[rr * NP + q])));
            }
            double lnp, lnl;
            const double post = lnpost_wave<KIND, NS, NB, false, false, false, false, ISO_UNI_LANE>(A, lds, L, active, M, p, parts, lnp, lnl);
            if (n == 1) {
                if (lane == 0) {
                    sys_store(&mb->done[1], (unsigned long long)__double_as_longlong(post));
                    sys_store(&mb->done[2], (unsigned long long)__double_as_longlong(lnp));
                    sys_store(&mb->done[3], (unsigned long long)__double_as_longlong(lnl));
                }
            } else if (active) {
                mb->out[r] = post;
                mb->out[ISO_MAILBOX_ROWS + r] = lnp;
                mb->out[2 * ISO_MAILBOX_ROWS + r] = lnl;
            }
        }
#ifdef ISO_MAILBOX_CLOCK
        if (lane == 0) sys_store(&mb->done[4], wall_clock64() - t_seen);
#endif
        __threadfence_system();                                // results before the sequence word
        if (lane == 0) sys_store(&mb->done[0], seq);
        last = seq;
        t_idle = wall_clock64();
    }
    __threadfence_system();
    if (lane == 0) sys_store(reinterpret_cast<unsigned long long*>(&mb->ctl[0]), 2ull);      // state: exited
}

template <int KIND, int NS>
static const void* mailbox_fn(int nb)
{
    switch (nb) {
#define ISO_MB_CASE(N) case N: return (const void*)k_mailbox_lnpost<KIND, NS, N>;
        ISO_MB_CASE(0) ISO_MB_CASE(1) ISO_MB_CASE(2) ISO_MB_CASE(3) ISO_MB_CASE(4) ISO_MB_CASE(5) ISO_MB_CASE(6)
        ISO_MB_CASE(7) ISO_MB_CASE(8) ISO_MB_CASE(9) ISO_MB_CASE(10) ISO_MB_CASE(11) ISO_MB_CASE(12)
#undef ISO_MB_CASE
    }
    return nullptr;
}

}  // namespace fastk

// start the model's resident wave on stream s; false = no instantiation for this shape
bool launch_mailbox(int kind, int n_stars, int n_bands, const FastArgs& A, IsoMailbox* d_mb, unsigned long long idle_ticks,
                    unsigned long long life_ticks, hipStream_t s)
{
    using namespace fastk;
    const void* fn = nullptr;
    if (kind == ISO_KIND_TRACK) fn = n_stars == 1 ? mailbox_fn<ISO_KIND_TRACK, 1>(n_bands) : nullptr;
    else if (n_stars == 1) fn = mailbox_fn<ISO_KIND_ISO, 1>(n_bands);
    else if (n_stars == 2) fn = mailbox_fn<ISO_KIND_ISO, 2>(n_bands);
    else if (n_stars == 3) fn = mailbox_fn<ISO_KIND_ISO, 3>(n_bands);
    if (!fn) return false;
    const size_t sh = (size_t)(((A.axes_len + 1) & ~1) + 64 * slot_stride(n_bands)) * sizeof(double);
    note_kernel("k_mailbox_lnpost<%d, %d, %d>", kind, n_stars, n_bands);
    void* args[] = {const_cast<FastArgs*>(&A), &d_mb, &idle_ticks, &life_ticks};
    return hipLaunchKernel(fn, dim3(1), dim3(64), args, sh, s) == hipSuccess;
}

}  // namespace iso
