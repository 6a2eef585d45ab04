// Which SIMD of its CU does wave k of a 256-thread workgroup land on?  (decides whether rotating the busy waves of partially
// filled workgroups over the wave index evens out the SIMDs: sampler.h, ISO_DENSE_PACKED)
//   hipcc --offload-arch=gfx950 -O2 tools/simd_probe.hip -o variants/bin/simd_probe && variants/bin/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out, int spin)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    // keep the workgroup resident for a while so that several share a CU
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
}
int main()
{
    const int nb = 2048;
    unsigned* d;
    hipMalloc(&d, nb * 4 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 4; ++w) hist[w][(h[b * 4 + w] >> 4) & 3]++;
    printf("rows: wave index in the workgroup; columns: SIMD_ID (HW_ID bits 5:4)\n");
    for (int w = 0; w < 4; ++w) printf("wave %d: %5d %5d %5d %5d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("first workgroups (hw_id of waves 0..3):\n");
    for (int b = 0; b < 8; ++b) printf("  wg %d: %08x %08x %08x %08x\n", b, h[b * 4], h[b * 4 + 1], h[b * 4 + 2], h[b * 4 + 3]);
    return 0;
}
