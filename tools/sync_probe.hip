// Round-trip latency of "launch a tiny kernel, wait for its result on the host" under different wait methods:
//   (a) hipStreamSynchronize, (b) spin on hipStreamQuery, (c) spin on hipEventQuery,
//   (d) spin on a flag in pinned mapped memory written by the kernel itself,
//   (e) as (d) with the flag written by a second 1-thread kernel on the same stream.
// Build: hipcc --offload-arch=gfx950 -O2 -o sync_probe tools/sync_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_work(const double* in, double* out, volatile uint64_t* flag, uint64_t seq)
{
    double v = in[threadIdx.x];
    for (int i = 0; i < 40; ++i) v = v * 1.0000001 + 1e-9;
    out[threadIdx.x] = v;
    if (flag) {
        __syncthreads();
        if (threadIdx.x == 0) { __threadfence_system(); *flag = seq; }
    }
}
__global__ void k_signal(volatile uint64_t* flag, uint64_t seq) { __threadfence_system(); *flag = seq; }

int main(int argc, char** argv)
{
    double* h; CK(hipHostMalloc(reinterpret_cast<void**>(&h), 4096, hipHostMallocMapped));
    double* d; CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0));
    volatile uint64_t* hflag = reinterpret_cast<volatile uint64_t*>(h + 256);
    volatile uint64_t* dflag = reinterpret_cast<volatile uint64_t*>(d + 256);
    for (int i = 0; i < 64; ++i) h[i] = i;
    hipStream_t s = nullptr;
    if (argc > 1 && argv[1][0] == 'n') CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));   // "n": own non-blocking stream
    if (argc > 1 && argv[1][0] == 'b') CK(hipStreamCreate(&s));                                  // "b": own blocking stream
    printf("stream: %s\n", s ? (argv[1][0] == 'n' ? "non-blocking" : "blocking") : "null (legacy default)");
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int N = 5000;
    uint64_t seq = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = now();
            for (int i = 0; i < N; ++i) {
                ++seq;
                h[0] = static_cast<double>(i);
                switch (mode) {
                case 0: hipLaunchKernelGGL(k_work, 1, 64, 0, s, d, d + 64, nullptr, seq); CK(hipStreamSynchronize(s)); break;
                case 1: hipLaunchKernelGGL(k_work, 1, 64, 0, s, d, d + 64, nullptr, seq);
                        while (hipStreamQuery(s) == hipErrorNotReady) {} break;
                case 2: hipLaunchKernelGGL(k_work, 1, 64, 0, s, d, d + 64, nullptr, seq); CK(hipEventRecord(ev, s));
                        while (hipEventQuery(ev) == hipErrorNotReady) {} break;
                case 3: hipLaunchKernelGGL(k_work, 1, 64, 0, s, d, d + 64, dflag, seq);
                        while (*hflag != seq) {} break;
                case 4: hipLaunchKernelGGL(k_work, 1, 64, 0, s, d, d + 64, nullptr, seq);
                        hipLaunchKernelGGL(k_signal, 1, 1, 0, s, dflag, seq);
                        while (*hflag != seq) {} break;
                }
                if (h[64] < static_cast<double>(i)) { printf("stale result in mode %d\n", mode); return 1; }
            }
            auto t1 = now();
            if (rep == 1) {
                static const char* names[] = {"hipStreamSynchronize", "spin hipStreamQuery", "spin hipEventQuery",
                                              "spin mapped flag (same kernel)", "spin mapped flag (signal kernel)"};
                printf("%-36s %.2f us per round trip\n", names[mode], us(t0, t1) / N);
            }
        }
        CK(hipDeviceSynchronize());
    }
    return 0;
}
