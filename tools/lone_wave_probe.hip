// Lone-wave latency probe: how fast does ONE wavefront run dependent work on an otherwise idle MI355X?
// (context for the persistent sampler kernel, which is instruction-latency bound).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lone_wave_probe tools/lone_wave_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void k_probe(double* out, long long* t, const int* chase, int iters)
{
    __shared__ int lchase[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lchase[i] = (i * 17 + 1) & 1023;
    __syncthreads();
    double x = out[threadIdx.x];
    long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) x = fma(x, 1.0000001, 1e-9);          // dependent fp64 FMA chain
    long long w1 = wall_clock64(), c1 = clock64();
    double y = x;
    for (int i = 0; i < iters / 16; ++i) y = log(y + 2.0);                 // dependent fp64 log chain
    long long w2 = wall_clock64();
    int p = threadIdx.x;
    for (int i = 0; i < iters / 16; ++i) p = lchase[p];                    // dependent LDS chain
    long long w3 = wall_clock64();
    int q = threadIdx.x;
    for (int i = 0; i < iters / 64; ++i) q = chase[q];                     // dependent global (L2-hit) chain
    long long w4 = wall_clock64();
    out[threadIdx.x] = x + y + p + q;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        t[0] = w1 - w0; t[1] = c1 - c0; t[2] = w2 - w1; t[3] = w3 - w2; t[4] = w4 - w3;
    }
}

int main()
{
    const int iters = 1 << 16;
    double* out; long long* t; int* chase;
    hipMalloc(&out, 256 * 8 * 4096); hipMalloc(&t, 64); hipMalloc(&chase, 65536 * 4);
    hipMemset(out, 0, 256 * 8 * 4096);
    int* h = new int[65536];
    for (int i = 0; i < 65536; ++i) h[i] = (i * 4099 + 7) & 65535;
    hipMemcpy(chase, h, 65536 * 4, hipMemcpyHostToDevice);
    int wclk = 0;
    hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    for (int blocks : {1, 1, 256, 4096}) {
        for (int rep = 0; rep < 2; ++rep) {
            k_probe<<<blocks, 64>>>(out, t, chase, iters);
            hipDeviceSynchronize();
        }
        long long r[5];
        hipMemcpy(r, t, 40, hipMemcpyDeviceToHost);
        const double ns = 1e6 / wclk;   // ns per wall-clock tick (rate in kHz)
        printf("blocks=%d wallclock_kHz=%d: fma %.2f ns/op (%.2f shader cycles -> %.0f MHz), log %.1f ns/op, "
               "lds chase %.1f ns, L2 chase %.1f ns\n", blocks, wclk, r[0] * ns / iters, (double)r[1] / iters,
               r[1] / (r[0] * ns) * 1e3, r[2] * ns / (iters / 16), r[3] * ns / (iters / 16), r[4] * ns / (iters / 64));
    }
    return 0;
}
