// What does pinned host memory cost to get?  hipHostMalloc against
// mmap (+ huge pages) + hipHostRegister, alone and from two threads.  Dev probe.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

static double now()
{
    return std::chrono::duration<double>(
        std::chrono::steady_clock::now().time_since_epoch()).count() * 1e3;
}

static void *pin_malloc(size_t bytes, double *ms)
{
    void *p = nullptr;
    double t0 = now();
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    *ms = now() - t0;
    return e == hipSuccess ? p : nullptr;
}

static void *pin_register(size_t bytes, bool huge, double *ms_map, double *ms_reg)
{
    double t0 = now();
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE,
                   MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    if (huge) madvise(p, bytes, MADV_HUGEPAGE);
    memset(p, 0, bytes);            // touch: the pages exist before pinning
    *ms_map = now() - t0;
    t0 = now();
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    *ms_reg = now() - t0;
    return e == hipSuccess ? p : nullptr;
}

int main()
{
    hipFree(nullptr);
    const size_t bytes = (size_t)300 << 20;
    double a, b, c;
    for (int r = 0; r < 2; r++) {
        void *p = pin_malloc(bytes, &a);
        printf("hipHostMalloc 300 MiB: %.1f ms%s\n", a, p ? "" : " FAILED");
        double t0 = now();
        if (p) hipHostFree(p);
        printf("  hipHostFree: %.1f ms\n", now() - t0);
    }
    for (int huge = 0; huge < 2; huge++) {
        void *p = pin_register(bytes, huge, &b, &c);
        printf("mmap%s + touch: %.1f ms, hipHostRegister: %.1f ms%s\n",
               huge ? " (MADV_HUGEPAGE)" : "", b, c, p ? "" : " FAILED");
        if (p) {
            // is it usable as a DMA target at full speed?
            void *d;
            hipMalloc(&d, bytes);
            hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost);
            double t0 = now();
            hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost);
            printf("  D2H into it: %.1f ms (%.1f GB/s)\n", now() - t0,
                   bytes / (now() - t0) / 1e6);
            hipFree(d);
            hipHostUnregister(p);
            munmap(p, bytes);
        }
    }
    {
        void *d;
        hipMalloc(&d, bytes);
        void *p = pin_malloc(bytes, &a);
        hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost);
        double t0 = now();
        hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost);
        printf("D2H into hipHostMalloc memory: %.1f ms (%.1f GB/s)\n", now() - t0,
               bytes / (now() - t0) / 1e6);
        hipHostFree(p);
        hipFree(d);
    }
    double m1, m2;
    double t0 = now();
    std::thread t1([&] { void *p = pin_malloc(bytes, &m1); (void)p; });
    std::thread t2([&] { void *p = pin_malloc(bytes, &m2); (void)p; });
    t1.join();
    t2.join();
    printf("two threads, 300 MiB each: %.1f / %.1f ms, wall %.1f ms\n", m1, m2,
           now() - t0);
    return 0;
}
