// Micro-benchmark (not part of the library): the HBM ceiling of the lnpost access pattern.
// Every "sample" reads one random 384-B cell (3 cache lines) of a 1.93 GB table + one random
// 128-B line of a 54 MB table + 40 B of streamed parameters, writes 8 B.  Variants:
//   lane-per-sample (24 + 8 x dwordx4 per lane) vs 8 lanes-per-sample (3 + 1 x dwordx4 per lane).
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o gather_probe ; ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void k_lane(const double2* __restrict__ big, uint32_t ncell, const double2* __restrict__ bc,
                                              uint32_t nbc, const double* __restrict__ pars, double* __restrict__ out, int n, uint32_t salt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int q = 0; q < 5; ++q) acc += pars[(size_t)q * n + i];
    const uint32_t c = hash32(i * 2654435761u + salt) % ncell;
    const double2* p = big + (size_t)c * 24;
    double2 u[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) u[k] = p[k];
#pragma unroll
    for (int k = 0; k < 24; ++k) acc += u[k].x + u[k].y;
    const uint32_t b = hash32(i * 40503u + salt + 17u) % nbc;
    const double2* pb = bc + (size_t)b * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const double2 v = pb[k]; acc += v.x + v.y; }
    out[i] = acc;
}

// 8 lanes cooperate on one sample: each lane loads 3 x 16 B of the cell and 1 x 16 B of the BC line
__global__ __launch_bounds__(256) void k_coop(const double2* __restrict__ big, uint32_t ncell, const double2* __restrict__ bc,
                                              uint32_t nbc, const double* __restrict__ pars, double* __restrict__ out, int n, uint32_t salt)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 3, sub = t & 7;
    if (i >= n) return;
    const uint32_t c = hash32(i * 2654435761u + salt) % ncell;
    const double2* p = big + (size_t)c * 24;
    const double2 a0 = p[sub], a1 = p[8 + sub], a2 = p[16 + sub];
    const uint32_t b = hash32(i * 40503u + salt + 17u) % nbc;
    const double2 v = bc[(size_t)b * 8 + sub];
    double acc = a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + v.x + v.y;
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    if (sub == 0) {
        for (int q = 0; q < 5; ++q) acc += pars[(size_t)q * n + i];
        out[i] = acc;
    }
}

// 4 lanes cooperate on one sample: instruction k covers bytes [64k, 64k+64) of the 384-B cell
__global__ __launch_bounds__(256) void k_coop4(const double2* __restrict__ big, uint32_t ncell, const double2* __restrict__ bc,
                                               uint32_t nbc, const double* __restrict__ pars, double* __restrict__ out, int n, uint32_t salt)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 2, sub = t & 3;
    if (i >= n) return;
    const uint32_t c = hash32(i * 2654435761u + salt) % ncell;
    const double2* p = big + (size_t)c * 24;
    double2 a[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = p[4 * k + sub];
    const uint32_t b = hash32(i * 40503u + salt + 17u) % nbc;
    const double2 v0 = bc[(size_t)b * 8 + sub], v1 = bc[(size_t)b * 8 + 4 + sub];
    double acc = v0.x + v0.y + v1.x + v1.y;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc += a[k].x + a[k].y;
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (sub == 0) {
        for (int q = 0; q < 5; ++q) acc += pars[(size_t)q * n + i];
        out[i] = acc;
    }
}

int main()
{
    const uint32_t ncell = 15u * 196u * 1710u, nbc = 70u * 26u * 18u * 13u;
    const int n = 1000000;
    double2 *big, *bc; double *pars, *out;
    CHECK(hipMalloc(&big, (size_t)ncell * 384));
    CHECK(hipMalloc(&bc, (size_t)nbc * 128));
    CHECK(hipMalloc(&pars, (size_t)n * 40));
    CHECK(hipMalloc(&out, (size_t)n * 8));
    CHECK(hipMemset(big, 0, (size_t)ncell * 384));
    CHECK(hipMemset(bc, 0, (size_t)nbc * 128));
    CHECK(hipMemset(pars, 0, (size_t)n * 40));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (384.0 + 128.0 + 48.0) * n;
    for (int variant = 0; variant < 3; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 100; ++it) {
                if (variant == 0) hipLaunchKernelGGL(k_lane, dim3((n + 255) / 256), dim3(256), 0, 0, big, ncell, bc, nbc, pars, out, n, (uint32_t)it);
                else if (variant == 1) hipLaunchKernelGGL(k_coop, dim3((n * 8 + 255) / 256), dim3(256), 0, 0, big, ncell, bc, nbc, pars, out, n, (uint32_t)it);
                else hipLaunchKernelGGL(k_coop4, dim3((n * 4 + 255) / 256), dim3(256), 0, 0, big, ncell, bc, nbc, pars, out, n, (uint32_t)it);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.2f us/launch, %.0f GB/s of useful bytes (%.2f of 8 TB/s)\n", variant == 0 ? "lane-per-sample   " : variant == 1 ? "8-lanes-per-sample" : "4-lanes-per-sample",
                            ms * 10.0, bytes / (ms * 1e-5) / 1e9, bytes / (ms * 1e-5) / 8e12);
        }
    }
    return 0;
}
