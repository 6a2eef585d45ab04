// Calibration probe for the memory-side PMC counters (not part of the library): kernels whose HBM byte counts are
// known exactly, run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/pmc_calibrate.sh) so that the
// counter -> byte factors used for the library's kernels are measured in the library's own access widths:
//   stream16     every lane loads 16 B, coalesced, each byte of a 2 GiB buffer once              (read  2 GiB)
//   cells384     quads of lanes read whole 384-B cells (6 x 64 contiguous bytes per quad), every cell of a 1.5 GiB
//                table at most once per launch (a permutation: no reuse for any cache to find)    (read  n x 384 B)
//   lines128     quads read whole 128-B lines (2 x 64 B), each line once                          (read  n x 128 B)
//   write8       every lane stores one double, coalesced (the library kernels' result store)      (write n x 8 B)
//   write16      every lane stores 16 B, coalesced                                               (write n x 16 B)
// Every launch works on a fresh region / permutation salt, so nothing survives in the 256 MiB Infinity Cache between
// launches of one kind.  Prints the byte counts per launch; tools/pmc_calibrate.sh joins them with the counters.
// Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_calibrate.hip -o tools/pmc_calibrate.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_stream16(const double2* __restrict__ src, size_t n16, double* __restrict__ sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    if (i < n16) { const double2 v = src[i]; acc = v.x + v.y; }
    if (acc == 1.2345e300) sink[0] = acc;                       // never true: keeps the load
}

// cell of sample i: an odd multiplier modulo a power of two is a bijection, so n <= ncell samples touch n distinct cells
__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t salt, uint32_t mask) { return ((i + salt) * 2654435761u) & mask; }

__global__ __launch_bounds__(256) void k_cells384(const double2* __restrict__ tab, uint32_t mask, uint32_t n, uint32_t salt,
                                                  double* __restrict__ sink)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = t >> 2, sub = t & 3;
    if (i >= n) return;
    const double2* p = tab + (size_t)perm(i, salt, mask) * 24;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { const double2 v = p[4 * k + sub]; acc += v.x + v.y; }
    if (acc == 1.2345e300) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_lines128(const double2* __restrict__ tab, uint32_t mask, uint32_t n, uint32_t salt,
                                                  double* __restrict__ sink)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = t >> 2, sub = t & 3;
    if (i >= n) return;
    const double2* p = tab + (size_t)perm(i, salt, mask) * 8;
    const double2 a = p[sub], b = p[4 + sub];
    const double acc = a.x + a.y + b.x + b.y;
    if (acc == 1.2345e300) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_write8(double* __restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (double)i;
}

__global__ __launch_bounds__(256) void k_write16(double2* __restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = make_double2((double)i, 1.0);
}

int main(int argc, char** argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 6;
    const size_t big = (size_t)6 << 30;                         // 6 GiB arena, far beyond the Infinity Cache
    char* arena; double* sink;
    CHECK(hipMalloc(&arena, big));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(arena, 0, big));
    CHECK(hipDeviceSynchronize());
    const size_t stream_bytes = (size_t)2 << 30;
    const uint32_t cells_log2 = 22, lines_log2 = 24;            // 4 Mi cells x 384 B = 1.5 GiB; 16 Mi lines x 128 B = 2 GiB
    const uint32_t n_cells = 1000000, n_lines = 4000000;
    const size_t n_w8_small = 1000000, n_w8_big = (size_t)64 << 20, n_w16 = (size_t)32 << 20;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int it = 0; it < launches; ++it) {
        const size_t off = (size_t)(it % 3) * stream_bytes;     // three disjoint 2 GiB regions in turn
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_stream16, dim3((unsigned)((stream_bytes / 16 + 255) / 256)), dim3(256), 0, 0,
                           (const double2*)(arena + off), stream_bytes / 16, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("stream16   launch %d: %.1f us, read %zu B (%.0f GB/s)\n", it, ms * 1e3, stream_bytes, stream_bytes / (ms * 1e-3) / 1e9);
    }
    for (int it = 0; it < launches; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_cells384, dim3((n_cells * 4 + 255) / 256), dim3(256), 0, 0, (const double2*)arena,
                           (1u << cells_log2) - 1u, n_cells, (uint32_t)it * n_cells, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("cells384   launch %d: %.1f us, read %zu B (%.0f GB/s)\n", it, ms * 1e3, (size_t)n_cells * 384, n_cells * 384.0 / (ms * 1e-3) / 1e9);
    }
    for (int it = 0; it < launches; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_lines128, dim3((n_lines * 4 + 255) / 256), dim3(256), 0, 0, (const double2*)(arena + ((size_t)3 << 30)),
                           (1u << lines_log2) - 1u, n_lines, (uint32_t)it * n_lines, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("lines128   launch %d: %.1f us, read %zu B (%.0f GB/s)\n", it, ms * 1e3, (size_t)n_lines * 128, n_lines * 128.0 / (ms * 1e-3) / 1e9);
    }
    for (int it = 0; it < launches; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_write8, dim3((unsigned)((n_w8_small + 255) / 256)), dim3(256), 0, 0, (double*)(arena + (size_t)it * (64 << 20)), n_w8_small);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("write8s    launch %d: %.1f us, wrote %zu B\n", it, ms * 1e3, n_w8_small * 8);
    }
    for (int it = 0; it < launches; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_write8, dim3((unsigned)((n_w8_big + 255) / 256)), dim3(256), 0, 0, (double*)(arena + (size_t)(it % 3) * stream_bytes), n_w8_big);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("write8     launch %d: %.1f us, wrote %zu B (%.0f GB/s)\n", it, ms * 1e3, n_w8_big * 8, n_w8_big * 8.0 / (ms * 1e-3) / 1e9);
    }
    for (int it = 0; it < launches; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_write16, dim3((unsigned)((n_w16 + 255) / 256)), dim3(256), 0, 0, (double2*)(arena + (size_t)(it % 3) * stream_bytes), n_w16);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("write16    launch %d: %.1f us, wrote %zu B (%.0f GB/s)\n", it, ms * 1e3, n_w16 * 16, n_w16 * 16.0 / (ms * 1e-3) / 1e9);
    }
    CHECK(hipDeviceSynchronize());
    printf("expected_bytes k_stream16 read %zu\nexpected_bytes k_cells384 read %zu\nexpected_bytes k_lines128 read %zu\n"
           "expected_bytes k_write8 write %zu (first %d launches) / %zu (next %d)\nexpected_bytes k_write16 write %zu\n",
           stream_bytes, (size_t)n_cells * 384, (size_t)n_lines * 128, n_w8_small * 8, launches, n_w8_big * 8, launches, n_w16 * 16);
    return 0;
}
