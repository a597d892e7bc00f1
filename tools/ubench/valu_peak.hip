// dev microbenchmark: what does the FP64 vector pipe deliver for the inner
// block of k_ll (exec-masked v_add_f64 with SGPR addends)?  Not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_adds(double *out, int iters, double x, unsigned long long m1, unsigned long long m0)
{
    double a[8];
    for (int j = 0; j < 8; j++) a[j] = threadIdx.x * 1e-9 + j;
    double s0 = x, s1 = x * 2, s2 = x * 3, s3 = x * 4, s4 = x * 5, s5 = x * 6, s6 = x * 7, s7 = x * 8;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            asm volatile(
                "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %10\n v_add_f64 %3, %3, %11\n"
                "v_add_f64 %4, %4, %12\n v_add_f64 %5, %5, %13\n v_add_f64 %6, %6, %14\n v_add_f64 %7, %7, %15\n"
                "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %10\n v_add_f64 %3, %3, %11\n"
                "v_add_f64 %4, %4, %12\n v_add_f64 %5, %5, %13\n v_add_f64 %6, %6, %14\n v_add_f64 %7, %7, %15\n"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7));
        } else {
            asm volatile(
                "s_mov_b64 exec, %16\n"
                "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %10\n v_add_f64 %3, %3, %11\n"
                "v_add_f64 %4, %4, %12\n v_add_f64 %5, %5, %13\n v_add_f64 %6, %6, %14\n v_add_f64 %7, %7, %15\n"
                "s_mov_b64 exec, %17\n"
                "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %10\n v_add_f64 %3, %3, %11\n"
                "v_add_f64 %4, %4, %12\n v_add_f64 %5, %5, %13\n v_add_f64 %6, %6, %14\n v_add_f64 %7, %7, %15\n"
                "s_mov_b64 exec, -1\n"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7), "s"(m1), "s"(m0));
        }
    }
    double r = 0;
    for (int j = 0; j < 8; j++) r += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main()
{
    const int blocks = 256 * 8, iters = 20000;
    double *out;
    CHK(hipMalloc(&out, blocks * 256 * sizeof(double)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_adds<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-7, 0x5555aaaa5555aaaaull, 0xaa005500aa005500ull);
            else hipLaunchKernelGGL(k_adds<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-7, 0x5555aaaa5555aaaaull, 0xaa005500aa005500ull);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            double adds = (double)blocks * 256 * iters * 16;
            printf("mode %d (%s): %.3f ms  %.2f T lane-adds/s  (%.1f%% of 39.3T)\n", mode, mode ? "exec-masked" : "plain", ms, adds / ms / 1e9, adds / ms / 1e9 / 39.3 * 100);
        }
    }
    return 0;
}
