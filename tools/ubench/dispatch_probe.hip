// What does a small launch cost before any of its work?  Trivial kernels at
// the grid of the converged config-3 evaluation (880 workgroups x 256
// threads), by static LDS per workgroup and by a short dependent chain of
// scalar loads.  Dev probe (HIP events around 200 back-to-back launches).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_DOUBLES>
__global__ __launch_bounds__(256) void k_lds(double *out, int write)
{
    __shared__ double red[LDS_DOUBLES > 0 ? LDS_DOUBLES : 1];
    if (LDS_DOUBLES > 0) red[threadIdx.x % LDS_DOUBLES] = threadIdx.x;
    __syncthreads();
    if (write) out[(size_t)blockIdx.x * 256 + threadIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_chain(const double *__restrict__ t,
                                               int stages, double *out)
{
    // `stages` dependent rounds of wave-uniform loads (the scalar pipe)
    double s = 0.0;
    int at = blockIdx.x & 1023;
    for (int i = 0; i < stages; i++) {
        const double v = t[at];
        s += v;
        at = (at + 17 + (int)v) & 1023;     // v == 0: address known late
    }
    if (s == 12345.0) out[blockIdx.x] = s;
}

template <typename F>
static float timed(F launch, hipStream_t s)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 20; i++) launch();
    hipEventRecord(a, s);
    for (int i = 0; i < 200; i++) launch();
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 200.f;
}

int main()
{
    hipStream_t s;
    hipStreamCreate(&s);
    double *out, *tab;
    hipMalloc(&out, 4096 * 256 * sizeof(double));
    hipMalloc(&tab, 1024 * sizeof(double));
    hipMemset(tab, 0, 1024 * sizeof(double));
    for (int grid : {220, 880, 3520}) {
        printf("grid %d x 256 threads, us per launch (back to back):\n", grid);
        printf("  no LDS, no store      %.2f\n", timed([&] {
            hipLaunchKernelGGL(k_lds<0>, dim3(grid), dim3(256), 0, s, out, 0); }, s));
        printf("  no LDS, store         %.2f\n", timed([&] {
            hipLaunchKernelGGL(k_lds<0>, dim3(grid), dim3(256), 0, s, out, 1); }, s));
        printf("  8 KB LDS              %.2f\n", timed([&] {
            hipLaunchKernelGGL(k_lds<1024>, dim3(grid), dim3(256), 0, s, out, 0); }, s));
        printf("  32 KB LDS             %.2f\n", timed([&] {
            hipLaunchKernelGGL(k_lds<4096>, dim3(grid), dim3(256), 0, s, out, 0); }, s));
        printf("  64 KB LDS             %.2f\n", timed([&] {
            hipLaunchKernelGGL(k_lds<8192>, dim3(grid), dim3(256), 0, s, out, 0); }, s));
        for (int st : {4, 12, 24, 48})
            printf("  %2d dependent scalar-load stages  %.2f\n", st, timed([&] {
                hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, tab, st, out); }, s));
    }
    return 0;
}
