// dev microbenchmark (not product code): throughput of the OTHER mapping of
// the cells x clusters x mutations op that SURVEY.md section 7 lists and the
// north star sketches - lane <-> mutation, table slices resident per lane,
// the data word of a cell used as the EXEC mask, one 64-lane wavefront
// reduction per (cell, cluster).  Same FP64 adds as k_ll plus the reduction.
// Shape: C3 first sweep (5120 cells x 3152 clusters x 1024 mutations).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int W = 16;       // 64-bit words per row (M = 1024)
constexpr int KT = 2;       // clusters per wave (2*KT*W doubles = 64 per lane)

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_wavereduce(
    const ulonglong2 *__restrict__ rows /*[N][W]*/, const double *__restrict__ T /*[K][M][2]*/,
    int N, int K, int cells_per_wave, double *__restrict__ out /*[N][K]*/)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int kt = blockIdx.y;                       // cluster tile
    const int c0 = (blockIdx.x * 4 + wave) * cells_per_wave;
    if (c0 >= N) return;
    double L1[KT][W], L0[KT][W];
#pragma unroll
    for (int k = 0; k < KT; k++)
#pragma unroll
        for (int w = 0; w < W; w++) {
            const double2 v = *reinterpret_cast<const double2 *>(
                &T[((size_t)(kt * KT + k) * (W * 64) + w * 64 + lane) * 2]);
            L1[k][w] = v.x;
            L0[k][w] = v.y;
        }
    const int c1 = min(N, c0 + cells_per_wave);
    for (int c = c0; c < c1; c++) {
        const ulonglong2 *__restrict__ r = rows + (size_t)c * W;
        double acc[KT];
#pragma unroll
        for (int k = 0; k < KT; k++) acc[k] = 0.0;
#pragma unroll
        for (int w = 0; w < W; w++) {
            const ulonglong2 m = r[w];               // wave-uniform: s_load
            if (__builtin_amdgcn_inverse_ballot_w64(m.x)) {
#pragma unroll
                for (int k = 0; k < KT; k++) acc[k] += L1[k][w];
            }
            if (__builtin_amdgcn_inverse_ballot_w64(m.y)) {
#pragma unroll
                for (int k = 0; k < KT; k++) acc[k] += L0[k][w];
            }
        }
#pragma unroll
        for (int k = 0; k < KT; k++) {
            const double s = wave_sum(acc[k]);
            if (lane == 0) out[(size_t)c * K + kt * KT + k] = s;
        }
    }
}

int main()
{
    const int N = 5120, K = 3152, M = W * 64;
    std::vector<ulonglong2> rows((size_t)N * W);
    srand(1);
    for (auto &r : rows) {
        unsigned long long a = ((unsigned long long)rand() << 33) ^ ((unsigned long long)rand() << 11) ^ rand();
        unsigned long long b = ((unsigned long long)rand() << 33) ^ ((unsigned long long)rand() << 11) ^ rand();
        r.x = a & b;              // ~25 % ones
        r.y = ~a & (b | (a >> 1));
    }
    std::vector<double> T((size_t)K * M * 2);
    for (auto &t : T) t = -(double)(rand() % 1000) / 100.0;
    ulonglong2 *d_rows; double *d_T, *d_out;
    CHK(hipMalloc(&d_rows, rows.size() * sizeof(ulonglong2)));
    CHK(hipMalloc(&d_T, T.size() * sizeof(double)));
    CHK(hipMalloc(&d_out, (size_t)N * K * sizeof(double)));
    CHK(hipMemcpy(d_rows, rows.data(), rows.size() * sizeof(ulonglong2), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_T, T.data(), T.size() * sizeof(double), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int cpw : {64, 128, 256}) {
        dim3 grid((N / cpw + 3) / 4, K / KT);
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_wavereduce, grid, dim3(256), 0, 0, d_rows, d_T, N, K, cpw, d_out);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("wave-reduce mapping, %3d cells/wave: %.3f ms  %.2fe12 element-evals/s\n",
               cpw, best, (double)N * K * M / best / 1e9);
    }
    return 0;
}
