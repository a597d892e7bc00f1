// Can the host write straight into device memory (large BAR)?  For each kind of
// device allocation: CPU memcpy of a payload into it (timed), then a kernel
// that checks the bytes.  A fault is caught and reported.  Dev probe only.
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <vector>

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

__global__ void k_sum(const unsigned *p, size_t n, unsigned long long *out)
{
    unsigned long long s = 0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    atomicAdd(out, s);
}

// one dependent round trip: a single lane reads one word
__global__ void k_touch(const unsigned *p, unsigned *out) { *out = p[0]; }

static double now()
{
    return std::chrono::duration<double>(
        std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void probe(const char *name, void *p, size_t bytes)
{
    std::vector<unsigned> src(bytes / 4);
    for (size_t i = 0; i < src.size(); i++) src[i] = (unsigned)i * 2654435761u;
    unsigned long long want = 0;
    for (unsigned v : src) want += v;
    unsigned long long *d_out;
    hipMalloc(&d_out, 8);
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1)) {
        printf("%-28s host write FAULTS\n", name);
        fflush(stdout);
        return;
    }
    memcpy(p, src.data(), bytes);               // may fault
    const int reps = 200;
    double t0 = now();
    for (int r = 0; r < reps; r++) {
        memcpy(p, src.data(), bytes);
        __builtin_ia32_sfence();
    }
    double dt = (now() - t0) / reps;
    hipMemset(d_out, 0, 8);
    hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, 0, (const unsigned *)p,
                       bytes / 4, d_out);
    unsigned long long got = 0;
    hipMemcpy(&got, d_out, 8, hipMemcpyDeviceToHost);
    // write + launch + kernel reads it: wall per iteration
    unsigned *d_t;
    hipMalloc(&d_t, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 100; r++)
        hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, 0,
                           (const unsigned *)p, d_t);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s host write %6.1f us per %zu KiB (%.1f GB/s), kernel sees it: %s, "
           "touch kernel %.2f us\n", name, dt * 1e6, bytes >> 10,
           bytes / dt / 1e9, got == want ? "yes" : "NO", ms * 10.0);
    fflush(stdout);
}

int main()
{
    const size_t bytes = 64 << 10;
    void *p;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess) {
        void *d;
        hipHostGetDevicePointer(&d, p, 0);
        // host memory written by the host, read by the kernel over the link
        std::vector<unsigned> src(bytes / 4, 7u);
        memcpy(p, src.data(), bytes);
        probe("pinned host (coherent)", p, bytes);
    }
    if (hipHostMalloc(&p, bytes, hipHostMallocNonCoherent) == hipSuccess)
        probe("pinned host (non-coherent)", p, bytes);
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) == hipSuccess)
        probe("device fine-grained", p, bytes);
    else
        printf("device fine-grained: allocation refused\n");
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) == hipSuccess)
        probe("device uncached", p, bytes);
    else
        printf("device uncached: allocation refused\n");
    if (hipMallocManaged(&p, bytes) == hipSuccess)
        probe("managed", p, bytes);
    if (hipMalloc(&p, bytes) == hipSuccess)
        probe("device (hipMalloc)", p, bytes);
    return 0;
}
