// Can the host write a request word straight into device memory (large BAR), so that a resident wave polls LOCAL memory
// instead of reading pinned host memory across PCIe on every poll?  Probes, each in a child process (a host store to an
// unmapped device pointer is a SIGSEGV):
//   1. hipMalloc                      2. hipExtMallocWithFlags(hipDeviceMallocFinegrained)
//   3. hipExtMallocWithFlags(hipDeviceMallocUncached)
// and for every kind the host can write: the round trip host store -> resident wave sees it -> wave stores an answer to
// pinned host memory -> host sees it, against the same round trip with the request word in pinned host memory.
//   hipcc --offload-arch=gfx950 -O2 tools/bar_probe.hip -o /tmp/bar_probe && /tmp/bar_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>
#include <immintrin.h>

__global__ void k_echo(volatile unsigned long long* req, volatile unsigned long long* done, unsigned long long life_ticks)
{
    const unsigned long long t0 = wall_clock64();
    unsigned long long last = 0;
    for (;;) {
        const unsigned long long v = __hip_atomic_load((unsigned long long*)req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == ~0ull) break;
        if (v != last) {
            __hip_atomic_store((unsigned long long*)done, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            last = v;
        }
        if (wall_clock64() - t0 > life_ticks) break;
    }
}

static double round_trip_us(unsigned long long* req_host_view, unsigned long long* req_dev_view, int n)
{
    unsigned long long *done = nullptr, *d_done = nullptr;
    hipHostMalloc((void**)&done, 64, hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&d_done, done, 0);
    *done = 0;
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipLaunchKernelGGL(k_echo, dim3(1), dim3(1), 0, s, req_dev_view, d_done, 100000000ull * 5);      // 5 s of life
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= n; ++i) {
        __atomic_store_n(req_host_view, (unsigned long long)i, __ATOMIC_RELEASE);
        _mm_sfence();                                          // (a write-combined mapping keeps stores in the core's buffers)
        unsigned long long spins = 0;
        while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != (unsigned long long)i) {
            if ((++spins & 0xFFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(4)) {
                __atomic_store_n(req_host_view, ~0ull, __ATOMIC_RELEASE);
                hipStreamSynchronize(s);
                return -1.0;                                   // the wave never saw the store
            }
        }
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    __atomic_store_n(req_host_view, ~0ull, __ATOMIC_RELEASE);
    hipStreamSynchronize(s);
    return us;
}

static void probe(const char* name, int kind)
{
    fflush(stdout);
    const pid_t pid = fork();
    if (pid == 0) {
        unsigned long long* p = nullptr;
        hipError_t e = hipSuccess;
        if (kind == 0) e = hipMalloc((void**)&p, 4096);
        else if (kind == 1) e = hipExtMallocWithFlags((void**)&p, 4096, hipDeviceMallocFinegrained);
        else if (kind == 2) e = hipExtMallocWithFlags((void**)&p, 4096, hipDeviceMallocUncached);
        else {
            unsigned long long* h = nullptr;
            e = hipHostMalloc((void**)&h, 4096, hipHostMallocMapped);
            unsigned long long* d = nullptr;
            hipHostGetDevicePointer((void**)&d, h, 0);
            *h = 0;
            printf("%-34s round trip %.2f us\n", name, round_trip_us(h, d, 20000));
            fflush(stdout);
        _exit(0);
        }
        if (e != hipSuccess) {
            printf("%-34s allocation failed: %s\n", name, hipGetErrorString(e));
            fflush(stdout);
        _exit(0);
        }
        hipMemset(p, 0, 4096);
        hipDeviceSynchronize();
        printf("%-34s allocated at %p; host store ... ", name, (void*)p);
        fflush(stdout);
        __atomic_store_n(p, 0ull, __ATOMIC_RELEASE);                      // SIGSEGV here if the host has no mapping
        printf("ok; host load %llu; ", (unsigned long long)__atomic_load_n(p, __ATOMIC_ACQUIRE));
        printf("round trip %.2f us\n", round_trip_us(p, p, 20000));
        fflush(stdout);
        _exit(0);
    }
    int st = 0;
    waitpid(pid, &st, 0);
    if (WIFSIGNALED(st)) printf("-> signal %d (no host mapping)\n", WTERMSIG(st));
}

int main()
{
    probe("pinned host memory (today)", 3);
    probe("hipMalloc", 0);
    probe("hipExtMalloc fine-grained", 1);
    probe("hipExtMalloc uncached", 2);
    return 0;
}
