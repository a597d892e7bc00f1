// bnpc_codist.hip - posterior co-clustering distance.
//
//   differ[(i,j)] = #{ samples s : assignment[s][i] != assignment[s][j] },
//   i < j, condensed in scipy's pdist order
//   = the per-sample pdist(..., 'hamming') accumulation of
//     utils.get_dist (/root/reference/libs/utils.py:90-97); the mean distance
//     is differ / S.  Exact integers -> bit-exact parity.
//
// Work is S * N^2 / 2 label compares (4e10 at 3350 samples x 5000 cells): a
// workgroup owns a 64 x 64 tile of cell pairs, streams the samples through
// LDS 32 at a time (two 64-label rows per sample) and every thread keeps a
// 4 x 4 block of pair counters in registers.  int32 VALU, LDS-broadcast
// reads; only tiles on or above the diagonal do any work.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bnpc_hip.h"
#include "bnpc_internal.h"

#define SC 32       // samples staged per LDS round

__global__ __launch_bounds__(256) void k_codist(
    const int *__restrict__ assign, long long S, long long N,
    int *__restrict__ differ)
{
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (ti > tj) return;
    __shared__ int A[SC][64];
    __shared__ int B[SC][64];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    int cnt[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) cnt[a][b] = 0;

    for (long long s0 = 0; s0 < S; s0 += SC) {
        // stage SC samples x (64 + 64) labels: 4096 ints by 256 threads
#pragma unroll
        for (int r = 0; r < (2 * SC * 64) / 256; r++) {
            const int e = r * 256 + tid;
            const int which = e / (SC * 64);
            const int rem = e - which * (SC * 64);
            const int sc = rem >> 6, col = rem & 63;
            const long long s = s0 + sc;
            const long long cell = (long long)(which ? tj : ti) * 64 + col;
            int v = -1 - col;       // padding never equals a real label
            if (s < S && cell < N) v = assign[s * N + cell];
            if (which) B[sc][col] = v; else A[sc][col] = v;
        }
        __syncthreads();
        const int lim = (S - s0 < SC) ? (int)(S - s0) : SC;
        for (int sc = 0; sc < lim; sc++) {
            const int4 av = *reinterpret_cast<const int4 *>(&A[sc][ty * 4]);
            const int4 bv = *reinterpret_cast<const int4 *>(&B[sc][tx * 4]);
            const int a4[4] = {av.x, av.y, av.z, av.w};
            const int b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) cnt[a][b] += (a4[a] != b4[b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const long long i = (long long)ti * 64 + ty * 4 + a;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const long long j = (long long)tj * 64 + tx * 4 + b;
            if (i < j && j < N) {
                const long long idx = i * (2 * N - i - 1) / 2 + (j - i - 1);
                differ[idx] = cnt[a][b];
            }
        }
    }
}

extern "C" int bnpc_codist(int device, const int32_t *assignments, int64_t S,
                           int64_t N, int32_t *differ)
{
    if (!assignments || !differ || S < 1 || N < 2) {
        bnpc_set_error("bad argument: need S >= 1 samples of N >= 2 cells");
        return 2;
    }
#define CK(expr)                                                             \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                              \
            bnpc_set_error("%s failed: %s", #expr, hipGetErrorString(e_));   \
            if (d_a) (void)hipFree(d_a);                                     \
            if (d_d) (void)hipFree(d_d);                                     \
            return 1;                                                        \
        }                                                                    \
    } while (0)
    int *d_a = nullptr, *d_d = nullptr;
    const size_t pairs = (size_t)N * (N - 1) / 2;
    CK(hipSetDevice(device));
    CK(hipMalloc((void **)&d_a, (size_t)S * N * sizeof(int)));
    CK(hipMalloc((void **)&d_d, pairs * sizeof(int)));
    CK(hipMemcpy(d_a, assignments, (size_t)S * N * sizeof(int),
                 hipMemcpyHostToDevice));
    const unsigned nt = (unsigned)((N + 63) / 64);
    hipLaunchKernelGGL(k_codist, dim3(nt, nt), dim3(256), 0, 0, d_a,
                       (long long)S, (long long)N, d_d);
    CK(hipGetLastError());
    CK(hipMemcpy(differ, d_d, pairs * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(d_a);
    (void)hipFree(d_d);
#undef CK
    return 0;
}
