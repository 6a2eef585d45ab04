// Resident service wave of the scalar accessors: interp_value / interp_mag / get_eep of ONE point without a launch
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

// -------------------------------------------------------------------------------------------
// The reference's accessors are called one point at a time from notebooks and from optimisers (DFInterpolator.__call__,
// interp.py:631-698; ModelGridInterpolator.interp_value / interp_mag, models.py:390-445; get_eep / interp_eep,
// models.py:501-542, interp.py:488-558 - `get_eep_accurate` minimises over interp_value calls).  A launch per call costs
// ~23 us through the C entry points; the evaluation itself is a few hundred instructions and a handful of table lines.
// As the per-point lnpost callback (iso_fast_mailbox.hip), ONE wave per context stays resident and polls a request in pinned,
// device-mapped host memory; here the request names what to evaluate:
//
//   host                                                   device (k_service, one wave)
//   req[1..7] = target, <= 5 coordinates, <= 8 column /
//               band numbers; req[0] = seq             -->  lanes 0-7 read the request line (one 64-B read over PCIe per poll); the
//   (9-32 columns: req[8..10] behind the line)               sequence word carries the opcode, the column count and a 32-bit checksum
//                                                            of the other words - a request whose words did not arrive together is
//                                                            polled again (iso_internal.h: mailbox_checksum)
//                                                            the target's axes are staged in LDS when the target changes (a
//                                                            target's record is only freed after the wave was told to leave, so
//                                                            its address names it)
//                                                            evaluation: interp_point / interp_mag_point / interp_eep_point -
//                                                            the per-sample bodies of k_interp / k_interp_mag / k_interp_eep,
//                                                            the same instructions on the same inputs: results are the
//                                                            launch path's bit for bit
//   spins on done[0]; reads out[]                     <--  out[0..] = values; fence; done[0] = seq
//
// The wave leaves when told (`quit`: the context goes), after `idle_ticks` without a request or after `life_ticks`.
// -------------------------------------------------------------------------------------------
constexpr int ISO_SVC_INTERP = 1, ISO_SVC_MAG = 2, ISO_SVC_EEP = 3;
constexpr int ISO_SVC_MAX_COLS = 32;       // columns / bands of one request (four words of bytes)
constexpr int ISO_SVC_LDS_DOUBLES = MAX_LDS_AXIS_DOUBLES;

// what a request addresses: device-resident, built once per table / interpolator / EEP table
struct SvcTarget {
    int op, ndim, kind, pad_;
    InterpArgs I;          // op INTERP (x / out / icols / k / n unused: they come with the request)
    MagArgs M;             // op MAG    (pars / outputs / bc_cols / nb unused)
    EepArgs E;             // op EEP    (x / out unused)
};

struct IsoSvcBox {
    unsigned long long req[16];   // [0] sequence word (checksum << 32 | counter << 16 | columns << 8 | opcode), [1] target (device
                                  // pointer), [2..6] coordinates / parameters, [7] column numbers 0-7 (a byte each); [8..10]
                                  // column numbers 8-31 of a request with more than eight
    unsigned long long done[8];   // [0] sequence word of the last finished request
    unsigned long long ctl[8];    // [0] state (0 none, 1 running, 2 exited), [1] quit
    double out[8 + ISO_SVC_MAX_COLS];
};

__device__ __forceinline__ unsigned long long svc_load(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void svc_store(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void svc_out(double* p, double v)
{
    svc_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}

__global__ __launch_bounds__(64, 1) void k_service(IsoSvcBox* mb, unsigned long long idle_ticks, unsigned long long life_ticks)
{
    extern __shared__ double lds[];
    const int lane = (int)threadIdx.x;
    unsigned long long last = svc_load(&mb->done[0]);
    unsigned long long staged = 0;                              // the target whose axes are in LDS
    const unsigned long long t_start = wall_clock64();
    unsigned long long t_idle = t_start;
    for (;;) {
        const unsigned long long w = svc_load(&mb->req[lane & 7]);
        const unsigned long long seq = __shfl(w, 0);
        if (seq == last) {
            const unsigned long long now = wall_clock64();
            const bool leave = (now - t_idle > idle_ticks) | (now - t_start > life_ticks) | (svc_load(&mb->ctl[1]) != 0);
            if (leave) break;                                  // (wave-uniform)
            continue;
        }
        const int op = (int)(seq & 0xFF), k = (int)((seq >> 8) & 0xFF);
        unsigned long long words[10];
#pragma unroll
        for (int q = 0; q < 7; ++q) words[q] = __shfl(w, 1 + q);
        words[7] = words[8] = words[9] = 0;
        if (k > 8) {                                           // (wave-uniform) the column numbers behind the line
            const unsigned long long w2 = svc_load(&mb->req[8 + (lane & 3)]);
#pragma unroll
            for (int q = 0; q < 3; ++q) words[7 + q] = __shfl(w2, q);
        }
        if (mailbox_checksum(words, 10) != (uint32_t)(seq >> 32)) continue;      // the words did not arrive together: poll again
        // the target's record through scalar loads (constant address space; the wave is told to leave before a record is freed)
        typedef const __attribute__((address_space(4))) SvcTarget* const_target_ptr;
        const unsigned long long tw = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(words[0] >> 32)) << 32) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(words[0] & 0xFFFFFFFFull));
        const SvcTarget* __restrict__ T = (const SvcTarget*)((const_target_ptr)(uintptr_t)tw);
        if (tw != staged) {
            __syncthreads();
            if (op == ISO_SVC_INTERP) stage_axes<ISO_MAX_DIM>(T->I.ax, lds);
            else if (op == ISO_SVC_MAG) {
                stage_axes<3>(T->M.g3.ax, lds);
                stage_axes<4>(T->M.g4.ax, lds);
            } else stage_axes<2>(T->E.ax, lds);
            __syncthreads();
            staged = tw;
        }
        double x[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) x[q] = __longlong_as_double((long long)words[1 + q]);
        auto col = [&](int c) { return (int)((words[6 + (c >> 3)] >> (8 * (c & 7))) & 0xFFull); };
        if (op == ISO_SVC_INTERP) {
            const int G = (k + 1) >> 1, sub = lane;            // lane `sub` owns the selected columns 2 sub, 2 sub + 1 (as k_interp)
            if (sub < G) {
                const int c0 = col(2 * sub);
                const bool two = (2 * sub + 1) < k;
                const int c1 = two ? col(2 * sub + 1) : c0;
                double v0 = d_nan(), v1 = d_nan();
                bool ok;
                switch (T->ndim) {
                case 2: ok = interp_point<2>(T->I, lds, x, c0, c1, v0, v1); break;
                case 3: ok = interp_point<3>(T->I, lds, x, c0, c1, v0, v1); break;
                default: ok = interp_point<4>(T->I, lds, x, c0, c1, v0, v1); break;
                }
                svc_out(&mb->out[2 * sub], ok ? v0 : d_nan());
                if (two) svc_out(&mb->out[2 * sub + 1], ok ? v1 : d_nan());
            }
        } else if (op == ISO_SVC_MAG) {
            const int nb = k, G = (nb + 1) >> 1 > 0 ? (nb + 1) >> 1 : 1, sub = lane;
            if (sub < G) {
                const bool has0 = (2 * sub) < nb, has1 = (2 * sub + 1) < nb;
                const int c0 = has0 ? col(2 * sub) : 0, c1 = has1 ? col(2 * sub + 1) : c0;
                double star[4], m0 = d_nan(), m1 = d_nan();
                if (T->kind == ISO_KIND_TRACK) interp_mag_point<ISO_KIND_TRACK>(T->M, lds, x[0], x[1], x[2], x[3], x[4], has0, c0, c1, star, m0, m1);
                else interp_mag_point<ISO_KIND_ISO>(T->M, lds, x[0], x[1], x[2], x[3], x[4], has0, c0, c1, star, m0, m1);
                if (sub == 0) {
                    svc_out(&mb->out[0], star[0]);
                    svc_out(&mb->out[1], star[1]);
                    svc_out(&mb->out[2], star[2]);
                }
                if (has0) svc_out(&mb->out[3 + 2 * sub], m0);
                if (has1) svc_out(&mb->out[3 + 2 * sub + 1], m1);
            }
        } else {
            // the four neighbouring tracks' age searches side by side (lanes 0-3; a search is up to eleven dependent reads)
            const EepCell c = interp_eep_cell(T->E, lds, x[0], x[1], x[2]);
            int64_t my_ie = 0, my_len = 0;
            if (c.ok && lane < 4) {
                const int64_t track = lane == 0 ? c.ind[0] : (lane == 1 ? c.ind[1] : (lane == 2 ? c.ind[2] : c.ind[3]));
                interp_eep_track(T->E, track, x[0], my_ie, my_len);
            }
            int64_t ie[4], len[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ie[q] = (int64_t)__shfl((long long)my_ie, q);
                len[q] = (int64_t)__shfl((long long)my_len, q);
            }
            if (lane == 0) svc_out(&mb->out[0], interp_eep_blend(T->E, c, ie, len));
        }
        __threadfence_system();                                // results before the sequence word
        if (lane == 0) svc_store(&mb->done[0], seq);
        last = seq;
        t_idle = wall_clock64();
    }
    __threadfence_system();
    if (lane == 0) svc_store(&mb->ctl[0], 2ull);              // state: exited
}
