// Device-resident stretch-move sampler for ANY evaluation: the persistent kernel body shared by the models that are not
// "BasicStarModel on the fused path with at most 12 bands" - observation trees (k_stretch_tree), IsoTrackModel
// (k_stretch_isotrack), 13-32 bands (k_stretch_wide).  Reference: StarModel.fit_mcmc drives emcee's stretch move around
// self.lnpost for every model class alike (isochrones/starmodel.py:886-972; Goodman & Weare 2010).
//
// Same move, same random numbers as fast/sampler.h (Philox4x32-10, counter (2 step + half, row_lo, row_hi, 0x51), key = seed;
// partner j, stretch factor z and acceptance uniform u2 built by the same arithmetic), so tests/_replay.py checks these
// chains move by move against the oracle exactly as it checks the fused kernels'.  What differs is the shape:
//   * the parameter count is a run-time number (a tree has sum(N_s + 4) of them, up to 24), so positions live in LDS as
//     [W][NP] rows and a proposal is never materialised: parameter q of the lane's proposal is rebuilt where the
//     evaluation asks for it, par(q) = fma(z, x_k[q] - x_j[q], x_j[q]) from the two LDS rows and z (a register) - the
//     rows a move reads are its own (written by nobody else) and one of the other half (written by nobody in this
//     half-step), so every read of a half-step sees the same numbers and the accepted position is bit for bit the
//     proposal that was evaluated;
//   * one workgroup owns ONE ensemble for all iterations of the launch (a fit of one model: nothing else runs; K
//     independent ensembles of the model are K workgroups);
//   * `lanes` of the workgroup's 256 take moves (a multiple of 64; fewer when the evaluator's per-lane LDS - the leaf values
//     of a tree with many stars - would not fit the 160 KB of a CU otherwise); a half-step of more moves runs in chunks.
// (included inside namespace iso::fastk, after iso_fast_kernel.h)
#pragma once

// (AnyStretchArgs: iso_internal.h - the host side fills it)

// LDS the sampler itself needs behind the evaluator's: positions, lnpost values, acceptance counters
inline size_t any_own_doubles(int W, int NP) { return (size_t)W * NP + W + (W + 1) / 2; }

// The kernels of this family take (AnyStretchArgs S, <the evaluator's arguments>...) - S FIRST: like the fused persistent
// kernels (sampler.h, ISO_KERNARG_REREAD) the loop reads its argument blocks again from the kernel-argument segment in every
// half-step, through a pointer the optimiser cannot see through - scalar loads where a field is used, instead of ~250
// scalar registers' worth of descriptors kept alive around the evaluation in vector lanes (v_readlane per use).
typedef const __attribute__((address_space(4))) char* kernarg_ptr;
template <class T>
__device__ __forceinline__ const T& kernarg_at(kernarg_ptr kp, size_t off)
{
    return *(const T*)(const __attribute__((address_space(4))) T*)(kp + off);
}
constexpr size_t kernarg_align8(size_t n) { return (n + 7) & ~size_t(7); }
constexpr size_t ANY_EVAL_ARGS = kernarg_align8(sizeof(AnyStretchArgs));      // offset of the evaluator's first argument

// EV: `double operator()(kernarg_ptr kp, bool active, Par par)` - lnpost of the lane's proposal; all lanes of a wave call it
// together; kp = this half-step's view of the kernel arguments.
template <class EV>
__device__ __forceinline__ void persist_any(EV& ev, const AnyStretchArgs& S0, double* lds)
{
    const AnyStretchArgs& S = S0;
    const int NP = S.NP, W = S.W, h = W >> 1;
    double* lpos = lds + S.own_off;
    double* llnp = lpos + W * NP;
    int32_t* lacc = reinterpret_cast<int32_t*>(llnp + W);
    const int64_t r0 = (int64_t)blockIdx.x * W;
    for (int j = threadIdx.x; j < W * NP; j += BLOCK) lpos[j] = S.pos[r0 * NP + j];
    for (int j = threadIdx.x; j < W; j += BLOCK) {
        llnp[j] = S.lnp[r0 + j];
        lacc[j] = 0;
    }
    __syncthreads();
    const int64_t rows_total = S.n_ens * W;
    // lane -> move of the chunk.  A half-step that does not fill the lanes is spread evenly over their waves in multiples of
    // 16 (the cooperative gathers serve 16 samples per round and skip empty rounds; a wave issues alone on its SIMD).  The
    // random numbers are keyed by (step, half, row), not by the lane: the chain does not depend on this mapping.
    const int lanes = S.lanes, nw = lanes >> 6;
    const int wave = (int)threadIdx.x >> 6, l6 = (int)threadIdx.x & 63;
    int per = h < lanes ? h : lanes, kk = (int)threadIdx.x;
    bool mine = kk < per;
    if (h < lanes) {
        int pw = (((h + nw - 1) / nw) + 15) & ~15;
        pw = pw > 64 ? 64 : pw;
        const int a = wave * pw + l6;
        mine = wave < nw && l6 < pw && a < h;
        kk = mine ? a : 0;
    }
    uint32_t key0 = (uint32_t)S.seed, key1 = (uint32_t)(S.seed >> 32);
    for (int it = 0; it < S0.nsteps; ++it) {
        for (int half = 0; half < 2; ++half) {
            kernarg_ptr kp = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kp));
            const AnyStretchArgs& S = kernarg_at<AnyStretchArgs>(kp, 0);
            for (int k0 = 0; k0 < h; k0 += per) {
                const int k = k0 + kk;
                const bool active = mine && k < h;
                if (!__any(active)) continue;                 // wave-uniform: idle waves go straight to the barrier
                const int lr = (half ? h : 0) + (active ? k : h - 1);
                ISO_STAMP_HERE(0);
                const int64_t row = r0 + lr;
                uint32_t rnd[4];
                asm volatile("" : "+s"(key0), "+s"(key1));    // (sampler.h: round keys as scalar additions)
                philox4x32_10((uint32_t)(2u * (S.step + (uint32_t)it) + (uint32_t)half), (uint32_t)row,
                              (uint32_t)((uint64_t)row >> 32), 0x51u, key0, key1, rnd);
                const int j = (int)(((uint64_t)rnd[0] * (uint64_t)h) >> 32);          // uniform in [0, h)
                const int lp = (half ? 0 : h) + j;
                const double u1 = ((double)rnd[1] + (double)(rnd[2] & 0xFFFFu) * (1.0 / 65536.0)) * (1.0 / 4294967296.0);
                const double u2 = ((double)rnd[3] + (double)(rnd[2] >> 16) * (1.0 / 65536.0) + 0.5 / 65536.0) * (1.0 / 4294967296.0);
                const double zr = (S.a - 1.0) * u1 + 1.0;
                const double z = zr * zr / S.a;
                // helper lanes (no move of their own) evaluate the partner's position and discard the result
                const double* xk = lpos + (active ? lr : lp) * NP;
                const double* xj = lpos + lp * NP;
                auto par = [&](int q) {
                    const double b = xj[q];
                    return fma(z, xk[q] - b, b);
                };
                ISO_STAMP_HERE(1);
                const double lnew = ev(kp, active, par);
                const double lold = llnp[lr];
                const double lnq = (NP - 1) * fast_log(z) + lnew - lold;
                const bool acc = active && isfinite(lnew) && (fast_log(u2) < lnq);
                double* mine_row = lpos + lr * NP;
                if (acc) {
                    for (int q = 0; q < NP; ++q) mine_row[q] = par(q);
                    llnp[lr] = lnew;
                    lacc[lr] += 1;
                }
                // chain recording: every move stores the row it owns (its value for this step)
                if (active && S.chain_pos) {
                    double* cp = S.chain_pos + (int64_t)it * rows_total * NP + row * S.chain_rs;
                    for (int q = 0; q < NP; ++q) cp[q * S.chain_ps] = mine_row[q];
                }
                if (active && S.chain_lnp) S.chain_lnp[(int64_t)it * rows_total + row] = acc ? lnew : lold;
                ISO_STAMP_HERE(9);
            }
            __syncthreads();
            ISO_STAMP_HERE(10);
        }
    }
    for (int j = threadIdx.x; j < W * NP; j += BLOCK) S.pos[r0 * NP + j] = lpos[j];
    for (int j = threadIdx.x; j < W; j += BLOCK) {
        S.lnp[r0 + j] = llnp[j];
        if (S.accepted) S.accepted[r0 + j] += lacc[j];
    }
}
