// Pinned host -> device, 1 MB and 7 MB (a part / the whole of config 5's
// parameter-batch draws): a copy-engine transfer against a kernel that reads
// the pinned memory in place (what k_mh_screen does today).  Dev probe.
//   hipcc --offload-arch=gfx950 -O3 -o h2d_probe h2d_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

__global__ void k_read(const double2 *__restrict__ src, double2 *__restrict__ dst,
                       long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

static double now_us()
{
    return std::chrono::duration<double>(
        std::chrono::steady_clock::now().time_since_epoch()).count() * 1e6;
}

int main()
{
    hipStream_t s, s2;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (size_t bytes : {(size_t)140 << 10, (size_t)1 << 20, (size_t)7 << 20}) {
        void *pin = nullptr, *pin_dev = nullptr, *dev = nullptr;
        hipHostMalloc(&pin, bytes, hipHostMallocDefault);
        hipHostGetDevicePointer(&pin_dev, pin, 0);
        hipMalloc(&dev, bytes);
        memset(pin, 1, bytes);
        const long long n = (long long)(bytes / 16);
        for (int mode = 0; mode < 2; mode++) {
            float best = 1e9f;
            double host_us = 0.0;
            for (int rep = 0; rep < 12; rep++) {
                memset(pin, rep, 4096);     // (the host touches the block)
                hipEventRecord(e0, s);
                const double t0 = now_us();
                if (mode == 0)
                    hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s);
                else
                    hipLaunchKernelGGL(k_read, dim3((unsigned)((n + 255) / 256)),
                                       dim3(256), 0, s,
                                       (const double2 *)pin_dev, (double2 *)dev, n);
                const double t1 = now_us();
                hipEventRecord(e1, s);
                hipEventSynchronize(e1);
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms < best) best = ms;
                if (rep >= 2) host_us += (t1 - t0) / 10.0;
            }
            printf("%8zu KB  %-28s %8.1f us  (%.1f GB/s), enqueue %.1f us on the host\n",
                   bytes >> 10, mode == 0 ? "hipMemcpyAsync (copy engine)"
                                          : "kernel reads pinned in place",
                   best * 1e3, bytes / (best * 1e-3) / 1e9, host_us);
        }
        hipFree(dev);
        hipHostFree(pin);
    }
    return 0;
}
