// How long after a kernel has finished does hipStreamSynchronize return?
// The kernel's last act is a system-scope store to a pinned flag the host
// spins on; compared with the return of the synchronisation.  Dev probe.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <vector>

__global__ void k_flag(volatile unsigned *flag, unsigned v, int spin)
{
    // a little work, so that the launch is not empty
    unsigned x = v;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __threadfence_system();
        flag[0] = v | (x & 0u);
    }
}

static double now()
{
    return std::chrono::duration<double>(
        std::chrono::steady_clock::now().time_since_epoch()).count() * 1e6;
}

int main()
{
    unsigned *flag;
    hipHostMalloc(&flag, 64, hipHostMallocDefault);
    unsigned *dflag;
    hipHostGetDevicePointer((void **)&dflag, flag, 0);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int mode = 0; mode < 2; mode++) {
        std::vector<double> t_launch, t_flag, t_sync;
        for (unsigned r = 1; r <= 2000; r++) {
            flag[0] = 0;
            double t0 = now();
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, dflag, r, 2000);
            double t1 = now();
            double tf = 0;
            if (mode == 0) {
                while (*(volatile unsigned *)flag != r) {}
                tf = now();
            }
            hipStreamSynchronize(s);
            double t2 = now();
            if (r > 100) {
                t_launch.push_back(t1 - t0);
                t_flag.push_back(tf - t0);
                t_sync.push_back(t2 - t0);
            }
        }
        auto med = [](std::vector<double> &v) {
            std::sort(v.begin(), v.end());
            return v[v.size() / 2];
        };
        if (mode == 0)
            printf("spin on flag, then sync: launch call %.2f us, flag seen %.2f us, "
                   "sync returned %.2f us after the launch began\n",
                   med(t_launch), med(t_flag), med(t_sync));
        else
            printf("sync only:               launch call %.2f us, sync returned %.2f us\n",
                   med(t_launch), med(t_sync));
    }
    // mode 2: the kernel does NOT write the flag; the stream does, behind it
    // (hipStreamWriteValue32: a command-processor write, no launch) and the
    // host spins on that word instead of calling hipStreamSynchronize
    {
        std::vector<double> t_launch, t_flag;
        unsigned *other;
        hipHostMalloc(&other, 64, hipHostMallocDefault);
        unsigned *dother;
        hipHostGetDevicePointer((void **)&dother, other, 0);
        bool ok = true;
        for (unsigned r = 1; r <= 2000 && ok; r++) {
            flag[0] = 0;
            double t0 = now();
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, dother, r, 2000);
            if (hipStreamWriteValue32(s, dflag, r, 0) != hipSuccess) {
                printf("hipStreamWriteValue32 failed: %s\n",
                       hipGetErrorString(hipGetLastError()));
                ok = false;
                break;
            }
            double t1 = now();
            while (*(volatile unsigned *)flag != r) {}
            double tf = now();
            if (other[0] != r) {
                printf("stream write overtook the kernel's store at %u\n", r);
                ok = false;
            }
            if (r > 100) {
                t_launch.push_back(t1 - t0);
                t_flag.push_back(tf - t0);
            }
        }
        if (ok) {
            std::sort(t_launch.begin(), t_launch.end());
            std::sort(t_flag.begin(), t_flag.end());
            printf("stream write value, spin:  launch + write calls %.2f us, "
                   "flag seen %.2f us after the launch began\n",
                   t_launch[t_launch.size() / 2], t_flag[t_flag.size() / 2]);
        }
    }
    // mode 3 / 4: an event recorded behind the kernel, waited for with
    // hipEventSynchronize / polled with hipEventQuery; mode 5: hipStreamQuery
    {
        hipEvent_t ev;
        hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        unsigned *other;
        hipHostMalloc(&other, 64, hipHostMallocDefault);
        unsigned *dother;
        hipHostGetDevicePointer((void **)&dother, other, 0);
        for (int mode = 3; mode <= 5; mode++) {
            std::vector<double> t_done;
            for (unsigned r = 1; r <= 2000; r++) {
                double t0 = now();
                hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, dother, r, 2000);
                if (mode != 5) hipEventRecord(ev, s);
                if (mode == 3) hipEventSynchronize(ev);
                else if (mode == 4) while (hipEventQuery(ev) == hipErrorNotReady) {}
                else while (hipStreamQuery(s) == hipErrorNotReady) {}
                double t2 = now();
                if (other[0] != r) printf("result not visible at %u\n", r);
                if (r > 100) t_done.push_back(t2 - t0);
            }
            std::sort(t_done.begin(), t_done.end());
            printf("%s: returned %.2f us after the launch began\n",
                   mode == 3 ? "event record + hipEventSynchronize"
                   : mode == 4 ? "event record + hipEventQuery poll "
                               : "hipStreamQuery poll               ",
                   t_done[t_done.size() / 2]);
        }
    }
    return 0;
}
