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
#include <math.h>
#include <stdint.h>
#include <algorithm>

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

// ---------------------------------------------------------------------------
// The posterior estimator as a pipeline: the pair
// counts STAY on the device, the mean distance goes to the host once (SciPy's
// Ward linkage needs it there), and every candidate cut of the tree is scored
// on the device in one pass over the counts.
//
// MPEAR of a clustering c (Fritsch & Ickstadt 2009, eq. 13;
// /root/reference/libs/utils.py:133-145) needs, with pi = 1 - differ / S,
//     I_sum  = #{pairs with c_i == c_j}            (from the label counts)
//     pi_sum = P - sum(differ) / S                  (one sum, all candidates)
//     index  = sum_{c_i == c_j} pi = I_sum - D_c / S,
//     D_c    = sum_{i < j, c_i == c_j} differ_ij   <- k_mpear_sums, int64
// i.e. exact integers, order-free; the float64 `pi` (10 GB at 50 000 cells)
// and the reference's pass over it per candidate are never made.
//
// k_mpear_sums: persistent workgroups walk the 64 x 64 pair tiles on or above
// the diagonal; a tile's counts sit in LDS (pairs with i >= j as zeros), the
// labels of its 64 + 64 cells under CP candidates beside them, candidate
// fastest (conflict-free: the lanes of a wave read consecutive candidates of
// one cell, the count of one pair is a broadcast).  Thread = (candidate,
// row part): it adds the counts of its pairs whose two cells share the
// candidate's label into ONE register accumulator that lives across all its
// tiles - no reduction inside the loop; at the end the parts are added
// through LDS and each workgroup makes one 64-bit atomic add per candidate.
// ---------------------------------------------------------------------------
#define MP_MAXCP 128

__global__ __launch_bounds__(256) void k_mpear_sums(
    const int *__restrict__ differ, long long N,
    const unsigned short *__restrict__ labels,  // [C][N]
    int c0, int C, int CP, unsigned long long *__restrict__ out)
{
    __shared__ int D[64][64];
    __shared__ unsigned short LA[64][MP_MAXCP];
    __shared__ unsigned short LB[64][MP_MAXCP];
    __shared__ unsigned long long red[256];
    const int tid = threadIdx.x;
    const int c = tid % CP, part = tid / CP, parts = 256 / CP;
    const long long nt = (N + 63) / 64;
    const long long tiles = nt * (nt + 1) / 2;
    unsigned long long acc = 0;
    for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
        // tile index -> (ti <= tj), rows of the upper triangle in order
        long long ti = (long long)((2.0 * nt + 1.0
            - sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)t))
            * 0.5);
        while (ti > 0 && ti * (2 * nt - ti + 1) / 2 > t) ti--;
        while ((ti + 1) * (2 * nt - ti) / 2 <= t) ti++;
        const long long tj = ti + (t - ti * (2 * nt - ti + 1) / 2);
        __syncthreads();                        // the previous tile is done
        for (int e = tid; e < 4096; e += 256) {
            const int a = e >> 6, b = e & 63;
            const long long i = ti * 64 + a, j = tj * 64 + b;
            int v = 0;
            if (i < j && j < N)
                v = differ[i * (2 * N - i - 1) / 2 + (j - i - 1)];
            D[a][b] = v;
        }
        for (int e = tid; e < 64 * CP; e += 256) {
            const int cell = e / CP, cc = e - cell * CP;
            const long long i = ti * 64 + cell, j = tj * 64 + cell;
            const bool live = c0 + cc < C;
            // cells past N / candidates past C: labels that match nothing
            LA[cell][cc] = (live && i < N)
                ? labels[(size_t)(c0 + cc) * N + i] : (unsigned short)0xfffe;
            LB[cell][cc] = (live && j < N)
                ? labels[(size_t)(c0 + cc) * N + j] : (unsigned short)0xffff;
        }
        __syncthreads();
        for (int a = part; a < 64; a += parts) {
            const unsigned short la = LA[a][c];
            unsigned sum = 0;                   // 64 counts <= S each
#pragma unroll 8
            for (int b = 0; b < 64; b++)
                sum += (LB[b][c] == la) ? (unsigned)D[a][b] : 0u;
            acc += sum;
        }
    }
    red[tid] = acc;
    __syncthreads();
    if (part == 0) {
        unsigned long long s = 0;
        for (int p = 0; p < parts; p++) s += red[p * CP + c];
        if (c0 + c < C && s) atomicAdd(&out[c0 + c], s);
    }
}

// sum of all pair counts (pi_sum), fixed grid, one atomic per workgroup
__global__ __launch_bounds__(256) void k_differ_sum(
    const int *__restrict__ differ, long long pairs,
    unsigned long long *__restrict__ out)
{
    unsigned long long acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pairs;
         i += (long long)gridDim.x * 256)
        acc += (unsigned long long)differ[i];
    __shared__ unsigned long long red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0]) atomicAdd(out, red[0]);
}

// mean distance differ / S as float64 (what SciPy's linkage takes): the same
// IEEE division NumPy performs on the host, 16 bytes per lane coalesced
__global__ __launch_bounds__(256) void k_differ_to_dist(
    const int *__restrict__ differ, long long pairs, double S,
    double *__restrict__ dist)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pairs;
         i += (long long)gridDim.x * 256)
        dist[i] = (double)differ[i] / S;
}

struct bnpc_post {
    int device = 0;
    int64_t S = 0, N = 0;
    int *differ = nullptr;                  // condensed, device
    unsigned long long *sums = nullptr;     // device scratch
};

#define PCK(expr)                                                            \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                              \
            bnpc_set_error("%s failed: %s", #expr, hipGetErrorString(e_));   \
            return 1;                                                        \
        }                                                                    \
    } while (0)

extern "C" int bnpc_post_destroy(bnpc_post *p)
{
    if (!p) return 0;
    (void)hipSetDevice(p->device);
    if (p->differ) (void)hipFree(p->differ);
    if (p->sums) (void)hipFree(p->sums);
    delete p;
    return 0;
}

extern "C" int bnpc_post_create(int device, const int32_t *assignments,
                                int64_t S, int64_t N, bnpc_post **out,
                                int64_t *differ_sum)
{
    if (!assignments || !out || S < 1 || N < 2) {
        bnpc_set_error("bad argument: need S >= 1 samples of N >= 2 cells");
        return 2;
    }
    *out = nullptr;
    PCK(hipSetDevice(device));
    bnpc_post *p = new bnpc_post();
    p->device = device;
    p->S = S;
    p->N = N;
    const size_t pairs = (size_t)N * (N - 1) / 2;
    int *d_a = nullptr;
    auto fail = [&]() {
        if (d_a) (void)hipFree(d_a);
        bnpc_post_destroy(p);
        return 1;
    };
#define PF(expr)                                                             \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                              \
            bnpc_set_error("%s failed: %s", #expr, hipGetErrorString(e_));   \
            return fail();                                                   \
        }                                                                    \
    } while (0)
    PF(hipMalloc((void **)&d_a, (size_t)S * N * sizeof(int)));
    PF(hipMalloc((void **)&p->differ, pairs * sizeof(int)));
    PF(hipMalloc((void **)&p->sums, (1024 + 1) * sizeof(unsigned long long)));
    PF(hipMemcpy(d_a, assignments, (size_t)S * N * sizeof(int),
                 hipMemcpyHostToDevice));
    const unsigned nt = (unsigned)((N + 63) / 64);
    hipLaunchKernelGGL(k_codist, dim3(nt, nt), dim3(256), 0, 0, d_a,
                       (long long)S, (long long)N, p->differ);
    PF(hipGetLastError());
    PF(hipMemsetAsync(p->sums, 0, sizeof(unsigned long long), 0));
    hipLaunchKernelGGL(k_differ_sum, dim3(1024), dim3(256), 0, 0, p->differ,
                       (long long)pairs, p->sums);
    PF(hipGetLastError());
    unsigned long long total = 0;
    PF(hipMemcpy(&total, p->sums, sizeof total, hipMemcpyDeviceToHost));
    (void)hipFree(d_a);
    d_a = nullptr;
#undef PF
    if (differ_sum) *differ_sum = (int64_t)total;
    *out = p;
    return 0;
}

// condensed pair counts / mean distances to the host (either may be NULL)
extern "C" int bnpc_post_fetch(bnpc_post *p, int32_t *differ, double *dist)
{
    if (!p) {
        bnpc_set_error("bad argument: NULL");
        return 2;
    }
    PCK(hipSetDevice(p->device));
    const size_t pairs = (size_t)p->N * (p->N - 1) / 2;
    if (differ)
        PCK(hipMemcpy(differ, p->differ, pairs * sizeof(int),
                      hipMemcpyDeviceToHost));
    if (dist) {
        // in slabs: the float64 form is twice the counts (10 GB at 50 000)
        const size_t slab = (size_t)64 << 20;           // elements
        double *d_slab = nullptr;
        PCK(hipMalloc((void **)&d_slab, std::min(slab, pairs) * sizeof(double)));
        for (size_t at = 0; at < pairs; at += slab) {
            const size_t n = std::min(slab, pairs - at);
            hipLaunchKernelGGL(k_differ_to_dist, dim3(2048), dim3(256), 0, 0,
                               p->differ + at, (long long)n, (double)p->S,
                               d_slab);
            hipError_t e = hipGetLastError();
            if (e == hipSuccess)
                e = hipMemcpy(dist + at, d_slab, n * sizeof(double),
                              hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                bnpc_set_error("mean distance: %s", hipGetErrorString(e));
                (void)hipFree(d_slab);
                return 1;
            }
        }
        (void)hipFree(d_slab);
    }
    return 0;
}

// same_differ[c] = sum over pairs i < j with labels[c][i] == labels[c][j] of
// differ_ij, for C candidate clusterings (labels < 65534)
extern "C" int bnpc_post_mpear(bnpc_post *p, const uint16_t *labels, int64_t C,
                               int64_t *same_differ)
{
    if (!p || !labels || !same_differ || C < 1 || C > 1024) {
        bnpc_set_error("bad argument: mpear sums of 1..1024 candidates");
        return 2;
    }
    PCK(hipSetDevice(p->device));
    unsigned short *d_lab = nullptr;
    const size_t bytes = (size_t)C * p->N * sizeof(unsigned short);
    PCK(hipMalloc((void **)&d_lab, bytes));
    hipError_t e = hipMemcpy(d_lab, labels, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemset(p->sums, 0, 1025 * sizeof(unsigned long long));
    const int CP = C <= 32 ? 32 : (C <= 64 ? 64 : 128);
    for (int c0 = 0; e == hipSuccess && c0 < C; c0 += CP) {
        hipLaunchKernelGGL(k_mpear_sums, dim3(1024), dim3(256), 0, 0,
                           p->differ, (long long)p->N, d_lab, c0, (int)C, CP,
                           p->sums + 1);
        e = hipGetLastError();
    }
    unsigned long long host[1024];
    if (e == hipSuccess)
        e = hipMemcpy(host, p->sums + 1, C * sizeof(unsigned long long),
                      hipMemcpyDeviceToHost);
    (void)hipFree(d_lab);
    if (e != hipSuccess) {
        bnpc_set_error("mpear sums: %s", hipGetErrorString(e));
        return 1;
    }
    for (int64_t c = 0; c < C; c++) same_differ[c] = (int64_t)host[c];
    return 0;
}

// ---------------------------------------------------------------------------
// Ward linkage of the mean distances on the device (the `linkage(dist,
// method='ward')` of /root/reference/libs/utils.py:104; SciPy is a pinned
// third-party dependency of the reference - scipy==1.10.1, requirements.txt:4
// - and what is restated here is its published algorithm for this call:
// scipy/cluster/_hierarchy.pyx `nn_chain` - the nearest-neighbour chain of
// Murtagh / Müllner with the Lance-Williams update `_ward`).
//
// At 50 000 cells the condensed distance vector is 10 GB and SciPy spends 62 s
// walking it row by row on one core; here the distances never leave the device
// (float64 from the resident pair counts, the same IEEE division, kept as a
// full symmetric matrix so that a row is contiguous), and the
// chain - inherently sequential: every step needs the previous one's result -
// runs inside ONE launch of ONE 1024-thread workgroup, so a step is two
// barriers, not a launch: the row scan for the nearest active neighbour
// (strictly smaller than the chain's previous element, smallest index among
// equals: the sequential scan's choice) as a workgroup reduction, the
// Lance-Williams pass over the merged cluster's row in parallel, the chain
// bookkeeping on thread 0.  Same merges, same heights bit for bit (sqrt, mul,
// add, div are IEEE on both sides, compiled without contraction); the final
// stable sort by height and the relabelling are done by the binding as SciPy
// does them.
// ---------------------------------------------------------------------------
#define WARD_T 1024
#define WARD_NONE 0x7fffffff

// the mean distances as a FULL symmetric matrix (rows contiguous: a row scan
// of the chain is coalesced; 20 GB at 50 000 cells), from the pair counts
__global__ __launch_bounds__(256) void k_differ_to_square(
    const int *__restrict__ differ, long long n, double S,
    double *__restrict__ F)
{
    const long long total = n * n;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        const long long i = e / n, j = e - i * n;
        double v = 0.0;
        if (i != j) {
            const long long a = i < j ? i : j, b = i < j ? j : i;
            v = (double)differ[a * (2 * n - a - 1) / 2 + (b - a - 1)] / S;
        }
        F[e] = v;
    }
}

__global__ __launch_bounds__(WARD_T) void k_ward_nnchain(
    double *__restrict__ F, long long n, int *__restrict__ size,
    int *__restrict__ chain, double *__restrict__ Z, int *__restrict__ err)
{
    __shared__ double r_d[WARD_T / 64];
    __shared__ int r_i[WARD_T / 64];
    __shared__ int s_x, s_y, s_done, s_nx, s_ny, s_len;
    __shared__ double s_min;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    for (long long i = tid; i < n; i += WARD_T) size[i] = 1;
    if (tid == 0) s_len = 0;
    __syncthreads();
    long long first_alive = 0;          // thread 0: sizes only ever drop to 0
    long long scans = 0;
    for (long long k = 0; k < n - 1; k++) {
        if (tid == 0 && s_len == 0) {
            while (first_alive < n && size[first_alive] == 0) first_alive++;
            chain[0] = (int)first_alive;
            s_len = 1;
        }
        for (;;) {
            if (tid == 0) {
                const int len = s_len;
                const int x = chain[len - 1];
                s_x = x;
                if (len > 1) {
                    s_y = chain[len - 2];
                    s_min = F[(size_t)x * n + s_y];
                } else {
                    s_y = -1;
                    s_min = INFINITY;
                }
            }
            __syncthreads();
            const int x = s_x;
            const double cmin = s_min;
            // nearest active neighbour of x: strictly below cmin, the
            // smallest index among equal distances.  Row x is contiguous;
            // four independent loads in flight per thread.
            const double *__restrict__ row = F + (size_t)x * n;
            double bd = cmin;
            int bi = WARD_NONE;
            long long i = tid;
            for (; i + 3 * WARD_T < n; i += 4 * WARD_T) {
                const int z0 = size[i], z1 = size[i + WARD_T],
                    z2 = size[i + 2 * WARD_T], z3 = size[i + 3 * WARD_T];
                const double d0 = row[i], d1 = row[i + WARD_T],
                    d2 = row[i + 2 * WARD_T], d3 = row[i + 3 * WARD_T];
                if (z0 && i != x && d0 < bd) {
                    bd = d0;
                    bi = (int)i;
                }
                if (z1 && i + WARD_T != x && d1 < bd) {
                    bd = d1;
                    bi = (int)(i + WARD_T);
                }
                if (z2 && i + 2 * WARD_T != x && d2 < bd) {
                    bd = d2;
                    bi = (int)(i + 2 * WARD_T);
                }
                if (z3 && i + 3 * WARD_T != x && d3 < bd) {
                    bd = d3;
                    bi = (int)(i + 3 * WARD_T);
                }
            }
            for (; i < n; i += WARD_T) {
                if (size[i] == 0 || i == x) continue;
                const double d = row[i];
                if (d < bd) {           // (this thread's i ascend)
                    bd = d;
                    bi = (int)i;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double od = __shfl_down(bd, off);
                const int oi = __shfl_down(bi, off);
                if (od < bd || (od == bd && oi < bi)) {
                    bd = od;
                    bi = oi;
                }
            }
            if (lane == 0) {
                r_d[wave] = bd;
                r_i[wave] = bi;
            }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < WARD_T / 64; w++)
                    if (r_d[w] < bd || (r_d[w] == bd && r_i[w] < bi)) {
                        bd = r_d[w];
                        bi = r_i[w];
                    }
                int y = s_y;
                if (bi != WARD_NONE) {
                    y = bi;
                    s_min = bd;
                }
                s_y = y;
                const int len = s_len;
                if (len > 1 && y == chain[len - 2]) {
                    s_done = 1;
                } else {
                    chain[len] = y;
                    s_len = len + 1;
                    s_done = 0;
                }
                if (++scans > 8 * n + 64 || y < 0) {   // cannot happen
                    *err = 1;
                    s_done = 2;
                }
            }
            __syncthreads();
            if (s_done) break;
        }
        if (s_done == 2) return;
        if (tid == 0) {
            s_len -= 2;
            int x = s_x, y = s_y;
            if (x > y) {
                const int t = x;
                x = y;
                y = t;
            }
            const int nx = size[x], ny = size[y];
            Z[k * 4 + 0] = (double)x;
            Z[k * 4 + 1] = (double)y;
            Z[k * 4 + 2] = s_min;
            Z[k * 4 + 3] = (double)(nx + ny);
            size[x] = 0;
            size[y] = nx + ny;
            s_x = x;
            s_y = y;
            s_nx = nx;
            s_ny = ny;
        }
        __syncthreads();
        {
            // Lance-Williams pass: rows x and y read and row y written
            // contiguously, column y (the symmetric entries) scattered
            const int x = s_x, y = s_y, nx = s_nx, ny = s_ny;
            const double dxy = s_min;
            const double *__restrict__ rx = F + (size_t)x * n;
            double *__restrict__ ry = F + (size_t)y * n;
            for (long long i = tid; i < n; i += WARD_T) {
                const int ni = size[i];
                if (ni == 0 || i == y) continue;
                const double dxi = rx[i];
                const double dyi = ry[i];
                // scipy's _ward
                const double t = 1.0 / (double)(nx + ny + ni);
                const double v = sqrt((double)(ni + nx) * t * dxi * dxi
                                      + (double)(ni + ny) * t * dyi * dyi
                                      - (double)ni * t * dxy * dxy);
                ry[i] = v;
                F[(size_t)i * n + y] = v;
            }
        }
        __syncthreads();
    }
}

// Z_raw[(N - 1) x 4]: the merges in the order the chain makes them (x < y:
// indices of the two clusters' slots, height, size) - what scipy's nn_chain
// holds before its final sort and relabelling.
extern "C" int bnpc_post_ward(bnpc_post *p, double *Z_raw)
{
    if (!p || !Z_raw) {
        bnpc_set_error("bad argument: NULL");
        return 2;
    }
    PCK(hipSetDevice(p->device));
    const long long n = p->N;
    double *d_D = nullptr, *d_Z = nullptr;
    int *d_size = nullptr, *d_chain = nullptr, *d_err = nullptr;
    // the full symmetric matrix: 8 N^2 bytes (20 GB at 50 000 cells).  Not
    // fitting is the ONE failure the caller may answer with SciPy's routine
    // on the condensed vector: it gets a return code of its own (5).
    {
        size_t free_b = 0, total_b = 0;
        PCK(hipMemGetInfo(&free_b, &total_b));
        const size_t need = (size_t)n * n * sizeof(double)
            + (size_t)n * 48 + ((size_t)1 << 20);
        if (need > free_b) {
            bnpc_set_error("ward linkage: the %lld x %lld distance matrix "
                           "needs %.1f GB, %.1f GB of device memory are free",
                           n, n, need / 1e9, free_b / 1e9);
            return 5;
        }
    }
    hipError_t e = hipMalloc((void **)&d_D, (size_t)n * n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_Z, (size_t)(n - 1) * 4 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_size, (size_t)n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&d_chain, (size_t)(n + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&d_err, sizeof(int));
    if (e == hipSuccess) e = hipMemset(d_err, 0, sizeof(int));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_differ_to_square, dim3(4096), dim3(256), 0, 0,
                           p->differ, n, (double)p->S, d_D);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_ward_nnchain, dim3(1), dim3(WARD_T), 0, 0, d_D, n,
                           d_size, d_chain, d_Z, d_err);
        e = hipGetLastError();
    }
    int err = 0;
    if (e == hipSuccess)
        e = hipMemcpy(&err, d_err, sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(Z_raw, d_Z, (size_t)(n - 1) * 4 * sizeof(double),
                      hipMemcpyDeviceToHost);
    if (d_D) (void)hipFree(d_D);
    if (d_Z) (void)hipFree(d_Z);
    if (d_size) (void)hipFree(d_size);
    if (d_chain) (void)hipFree(d_chain);
    if (d_err) (void)hipFree(d_err);
    if (e != hipSuccess) {
        bnpc_set_error("ward linkage: %s", hipGetErrorString(e));
        return 1;
    }
    if (err) {
        bnpc_set_error("ward linkage: the neighbour chain did not close "
                       "(non-finite distances?)");
        return 1;
    }
    return 0;
}
