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
#include <string.h>
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
    long long ward_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // row scans / chain steps of the
                                            // last bnpc_post_ward
};

#define PCK(expr)                                                            \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                              \
            bnpc_set_error("%s failed: %s", #expr, hipGetErrorString(e_));   \
            return 1;                                                        \
        }                                                                    \
    } while (0)

// row scans and chain steps of the last bnpc_post_ward (diagnostic)
extern "C" int bnpc_post_ward_stats(const bnpc_post *p, int64_t *scans,
                                    int64_t *steps)
{
    if (!p || !scans || !steps) {
        bnpc_set_error("bad argument: NULL");
        return 2;
    }
    *scans = p->ward_stats[0];
    *steps = p->ward_stats[1];
    return 0;
}

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
// full symmetric matrix so that a row is contiguous).
//
// The chain is sequential - every step needs the previous one's result - but
// the work inside a step is not, and ONE workgroup (round 3) cannot carry it:
// measured at 50 000 cells, 1.2 s of row scans and 4.2 s of Lance-Williams
// passes - a pass scatters an 8-byte store per cluster down a column of the
// matrix, and one compute unit hands the L2 one cache line per clock.  Round 4
// splits a step in two kernels that alternate on one stream (captured once as
// a graph of 128 pairs and replayed; the kernel boundary is the hand-off, all
// state lives in device memory, so there is no inter-workgroup protocol
// inside a launch to get wrong):
//
//   k_ward_chain   ONE wave takes in what the last piece of work left (the
//                  partial minima), walks the chain - a handful of dependent
//                  loads per step - until it needs work, posts it and ends:
//                  a merge, or the nearest neighbour of a stale row;
//   k_ward_work    the whole chip does it: ceil(N / 256) workgroups the
//                  Lance-Williams pass of the merge (row y contiguous,
//                  columns x and y scattered over every CU's path to the L2),
//                  ~N / 1500 workgroups a slice each of the row to be scanned
//                  - with a merge, the row the chain returns to (its
//                  neighbour was one of the two merged clusters, always), as
//                  the merge leaves it: column x gone, column y at the new
//                  distance, computed from values the chain kernel read
//                  beforehand, so the scan does not race with the pass.
//
// Work per step is cut as well:
//  * a row scan is a pure (minimum, smallest index) reduction: the diagonal,
//    the padding and the columns of merged-away clusters hold +inf, so there
//    is no activity test, 16-byte loads;
//  * every row keeps its nearest neighbour (nn_d, nn_i) = (minimum, smallest
//    index among equals) of its CURRENT entries, initialised for all rows by
//    one chip-wide pass (k_ward_init).  The Lance-Williams pass keeps that
//    exact - a row whose neighbour was one of the two merged clusters is
//    marked stale, any other row takes the merged cluster as its neighbour
//    iff (new distance, its index) is lexicographically smaller - and
//    computes the merged row's own neighbour on the way.  A chain step whose
//    top row has a valid neighbour needs no scan; a stale row is scanned when
//    (and only if) it reaches the top.  (With the many exact ties of
//    co-clustering distances the neighbours of a cluster's rows all point at
//    its smallest index, and its merges make them stale over and over: 2.6
//    scans per merge remain, 1 of them riding on the merge's own launch.
//    Keeping the k nearest per row instead was simulated: k = 8 still needs
//    1.3 scans per merge - not worth its bookkeeping.)
//
// N = 50 000: 4.2 s (round 3) -> 1.5 s; what is left is latency - 130 000
// launches of each kernel, ~5 us of dependent loads apiece.
//
// The sequential scan's choices are kept exactly: the nearest neighbour is
// the first index of the row's minimum, unless the chain's previous element
// is as near (`dist < current_min` is strict: the previous element wins
// ties); the distance to the previous element is the one recorded when it
// pushed the current top (neither has been merged since, so the entry is
// unchanged).  Same merges, same heights bit for bit (sqrt, mul, add, div are
// IEEE on both sides, compiled without contraction); the final stable sort by
// height and the relabelling are done by the binding as SciPy does them.
// ---------------------------------------------------------------------------
#define WARD_NONE 0x7fffffff

// the mean distances as a FULL symmetric matrix, rows `pitch` doubles apart
// (pitch even: rows are 16-byte aligned; 20 GB at 50 000 cells), from the
// pair counts; +inf on the diagonal and in the padding
__global__ __launch_bounds__(256) void k_differ_to_square(
    const int *__restrict__ differ, long long n, long long pitch, double S,
    double *__restrict__ F)
{
    const long long total = n * pitch;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        const long long i = e / pitch, j = e - i * pitch;
        double v = INFINITY;
        if (i != j && j < n) {
            const long long a = i < j ? i : j, b = i < j ? j : i;
            v = (double)differ[a * (2 * n - a - 1) / 2 + (b - a - 1)] / S;
        }
        F[e] = v;
    }
}

// (d, i) < (bd, bi) lexicographically
__device__ __forceinline__ void ward_take(double d, int i, double &bd, int &bi)
{
    if (d < bd || (d == bd && i < bi)) {
        bd = d;
        bi = i;
    }
}

// scipy's _ward: the distance of cluster i (ni cells) to the union of x and y
__device__ __forceinline__ double ward_lw(int ni, int nx, int ny, double dxi,
                                          double dyi, double dxy)
{
    const double t = 1.0 / (double)(nx + ny + ni);
    return sqrt((double)(ni + nx) * t * dxi * dxi
                + (double)(ni + ny) * t * dyi * dyi
                - (double)ni * t * dxy * dxy);
}

// state of the chain between launches (device memory)
struct WardState {
    long long merges, steps, scans, first_alive;
    int len, err, done;
    // the work k_ward_chain posted for k_ward_work: 1 = the merge (x, y) -> y,
    // x < y, plus - if b >= 0 - the nearest neighbour of row b AS THE MERGE
    // LEAVES IT (column x gone, column y at the new distance, computed from
    // nb, dxb, dyb); 2 = the nearest neighbour of row b as it stands
    int cmd, x, y, nx, ny, b, nb, pad_;
    double dxy, dxb, dyb;
};

#define WARD_B 256          // threads of a k_ward_work / k_ward_init workgroup
#define WARD_SCAN_U 4       // 16-byte loads per thread of a row-scan slice

__device__ __forceinline__ void ward_reduce_b(double &bd, int &bi, double *r_d,
                                              int *r_i)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_down(bd, off);
        const int oi = __shfl_down(bi, off);
        ward_take(od, oi, bd, bi);
    }
    if (lane == 0) {
        r_d[wave] = bd;
        r_i[wave] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < (int)(blockDim.x >> 6); w++)
            ward_take(r_d[w], r_i[w], bd, bi);
}

// (minimum, first index) of columns [2 * j0, 2 * j1) of one row: inactive
// columns, the diagonal and the padding hold +inf; columns skip_x / sub_y (or
// -1) are taken as +inf / sub_v instead of what memory holds
__device__ __forceinline__ void ward_scan_slice(
    const double *F, long long pitch, long long r, long long j0, long long j1,
    int skip_x, int sub_y, double sub_v, double &bd, int &bi)
{
    const double2 *row = (const double2 *)(F + (size_t)r * pitch);
    const double2 inf2 = {INFINITY, INFINITY};
    bd = INFINITY;
    bi = WARD_NONE;
    for (long long base = j0; base < j1;
         base += (long long)WARD_SCAN_U * blockDim.x) {
        double2 v[WARD_SCAN_U];
#pragma unroll
        for (int u = 0; u < WARD_SCAN_U; u++) {
            const long long j = base + (long long)u * blockDim.x + threadIdx.x;
            v[u] = j < j1 ? row[j] : inf2;
        }
#pragma unroll
        for (int u = 0; u < WARD_SCAN_U; u++) {
            const long long j = base + (long long)u * blockDim.x + threadIdx.x;
            const int c0 = (int)(2 * j), c1 = c0 + 1;
            double a0 = v[u].x, a1 = v[u].y;
            if (c0 == skip_x) a0 = INFINITY;
            if (c1 == skip_x) a1 = INFINITY;
            if (c0 == sub_y) a0 = sub_v;
            if (c1 == sub_y) a1 = sub_v;
            // (this thread's indices ascend: strict < keeps the first)
            if (a0 < bd) {
                bd = a0;
                bi = c0;
            }
            if (a1 < bd) {
                bd = a1;
                bi = c1;
            }
        }
    }
}

// every row's nearest neighbour, every cluster's size: one workgroup per row
__global__ __launch_bounds__(WARD_B) void k_ward_init(
    const double *__restrict__ F, long long n, long long pitch,
    int *__restrict__ size, double *__restrict__ nn_d, int *__restrict__ nn_i)
{
    __shared__ double r_d[WARD_B / 64];
    __shared__ int r_i[WARD_B / 64];
    for (long long r = blockIdx.x; r < n; r += gridDim.x) {
        double bd;
        int bi;
        ward_scan_slice(F, pitch, r, 0, pitch >> 1, -1, -1, 0.0, bd, bi);
        ward_reduce_b(bd, bi, r_d, r_i);
        if (threadIdx.x == 0) {
            size[r] = 1;
            nn_d[r] = bd;
            nn_i[r] = bi == WARD_NONE ? -1 : bi;
        }
        __syncthreads();
    }
}

// ONE wave: take in what the last k_ward_work left, then walk the chain until
// the next piece of work is known (a merge, or the scan of a stale row)
__global__ __launch_bounds__(64) void k_ward_chain(
    const double *__restrict__ F, long long n, long long pitch,
    int *__restrict__ size, int *__restrict__ chain,
    double *__restrict__ chain_d, double *__restrict__ nn_d,
    int *__restrict__ nn_i, double *__restrict__ Z,
    const double *__restrict__ pm_d, const int *__restrict__ pm_i, int Gm,
    const double *__restrict__ ps_d, const int *__restrict__ ps_i, int Gs,
    WardState *__restrict__ ws)
{
    const int lane = threadIdx.x;
    if (ws->err || ws->done) return;
    const int last = ws->cmd;
    if (last) {
        // the minima of the partial minima the workgroups left
        double md = INFINITY, sd = INFINITY;
        int mi = WARD_NONE, si = WARD_NONE;
        if (last == 1)
            for (int q = lane; q < Gm; q += 64) ward_take(pm_d[q], pm_i[q], md, mi);
        const int row = ws->b;
        if (row >= 0)
            for (int q = lane; q < Gs; q += 64) ward_take(ps_d[q], ps_i[q], sd, si);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double od = __shfl_down(md, off), pd = __shfl_down(sd, off);
            const int oi = __shfl_down(mi, off), pi = __shfl_down(si, off);
            ward_take(od, oi, md, mi);
            ward_take(pd, pi, sd, si);
        }
        if (lane == 0) {
            if (last == 1) {
                nn_d[ws->y] = md;
                nn_i[ws->y] = mi == WARD_NONE ? -1 : mi;
            }
            if (row >= 0) {
                nn_d[row] = sd;
                nn_i[row] = si == WARD_NONE ? -2 : si;
                if (si == WARD_NONE) ws->err = 1;   // nothing to merge with
            }
        }
    }
    if (lane != 0) return;
    if (ws->err) return;
    long long first_alive = ws->first_alive, steps = ws->steps,
        merges = ws->merges;
    int len = ws->len, cmd = 0;
    while (cmd == 0) {
        if (merges == n - 1) {
            ws->done = 1;
            break;
        }
        if (len == 0) {
            while (first_alive < n && size[first_alive] == 0) first_alive++;
            chain[0] = (int)first_alive;
            len = 1;
        }
        // (independent loads side by side: every one of them is an L2 round
        // trip, and the walk is nothing but their latencies)
        const int x = chain[len - 1];
        const int prev = chain[len > 1 ? len - 2 : 0];
        const int under = chain[len > 2 ? len - 3 : 0];
        const double dprev = chain_d[len - 1];
        const int cand = nn_i[x];
        const double dc = nn_d[x];
        if (cand < 0) {             // stale: scanned chip-wide
            ws->b = x;
            ws->scans++;
            cmd = 2;
            break;
        }
        int y = cand;
        double dmin = dc;
        if (len > 1 && !(dc < dprev)) {     // the previous element wins ties
            y = prev;
            dmin = dprev;
        }
        if (++steps > 8 * n + 64 || y < 0 || y >= n) {
            ws->err = 1;            // cannot happen
            break;
        }
        if (len > 1 && y == prev) {
            len -= 2;
            int a = x, b = y;
            if (a > b) {
                const int t = a;
                a = b;
                b = t;
            }
            const bool back = len > 0 && merges + 1 < n - 1;
            const int na = size[a], nb = size[b];
            const int top_n = back ? size[under] : 0;
            const double top_dx = back ? F[(size_t)a * pitch + under] : 0.0;
            const double top_dy = back ? F[(size_t)b * pitch + under] : 0.0;
            Z[merges * 4 + 0] = (double)a;
            Z[merges * 4 + 1] = (double)b;
            Z[merges * 4 + 2] = dmin;
            Z[merges * 4 + 3] = (double)(na + nb);
            size[a] = 0;
            size[b] = na + nb;
            merges++;
            ws->x = a;
            ws->y = b;
            ws->nx = na;
            ws->ny = nb;
            ws->dxy = dmin;
            // the row the chain returns to: its neighbour was one of the two
            ws->b = -1;
            if (back) {
                ws->b = under;
                ws->nb = top_n;
                ws->dxb = top_dx;
                ws->dyb = top_dy;
                ws->scans++;
            }
            cmd = 1;
            break;
        }
        chain[len] = y;
        chain_d[len] = dmin;
        len++;
    }
    ws->cmd = cmd;
    ws->first_alive = first_alive;
    ws->steps = steps;
    ws->merges = merges;
    ws->len = len;
}

// The posted work, on the whole chip.  Workgroups [0, Gm): the Lance-Williams
// pass of the merge (x, y) -> y, one cluster i per thread - rows x and y are
// read and row y written contiguously, columns x (now +inf) and y (the
// symmetric entries) are scattered over every compute unit's path to the L2;
// every row's nearest neighbour is kept exact on the way.  Workgroups
// [Gm, Gm + Gs): a slice each of the row whose neighbour is asked for.
__global__ __launch_bounds__(WARD_B) void k_ward_work(
    double *__restrict__ F, long long n, long long pitch,
    const int *__restrict__ size, double *__restrict__ nn_d,
    int *__restrict__ nn_i, double *__restrict__ pm_d,
    int *__restrict__ pm_i, int Gm, double *__restrict__ ps_d,
    int *__restrict__ ps_i, int Gs, const WardState *__restrict__ ws)
{
    __shared__ double r_d[WARD_B / 64];
    __shared__ int r_i[WARD_B / 64];
    const int cmd = ws->cmd;
    if (!cmd || ws->err || ws->done) return;
    const int x = ws->x, y = ws->y, nx = ws->nx, ny = ws->ny, rb = ws->b;
    const double dxy = ws->dxy;
    double bd = INFINITY;
    int bi = WARD_NONE;
    if ((int)blockIdx.x >= Gm) {
        if (rb < 0) return;
        const int s = (int)blockIdx.x - Gm;
        const long long n2 = pitch >> 1;
        const long long per = (n2 + Gs - 1) / Gs;
        const long long j0 = s * per, j1 = j0 + per < n2 ? j0 + per : n2;
        double vb = 0.0;
        if (cmd == 1) vb = ward_lw(ws->nb, nx, ny, ws->dxb, ws->dyb, dxy);
        ward_scan_slice(F, pitch, rb, j0, j1, cmd == 1 ? x : -1,
                        cmd == 1 ? y : -1, vb, bd, bi);
        ward_reduce_b(bd, bi, r_d, r_i);
        if (threadIdx.x == 0) {
            ps_d[s] = bd;
            ps_i[s] = bi;
        }
        return;
    }
    if (cmd != 1) return;
    const double *__restrict__ rx = F + (size_t)x * pitch;
    double *__restrict__ ry = F + (size_t)y * pitch;
    for (long long i = (long long)blockIdx.x * WARD_B + threadIdx.x; i < n;
         i += (long long)Gm * WARD_B) {
        if (i == x) {
            ry[i] = INFINITY;           // x is gone
            continue;
        }
        const int ni = size[i];
        if (ni == 0 || i == y) continue;
        const double v = ward_lw(ni, nx, ny, rx[i], ry[i], dxy);
        ry[i] = v;
        double *__restrict__ ri = F + (size_t)i * pitch;
        ri[y] = v;
        ri[x] = INFINITY;
        ward_take(v, (int)i, bd, bi);
        if (i == rb) continue;          // its neighbour is being recomputed
        // row i's nearest neighbour, kept exact
        const int ci = nn_i[i];
        if (ci == x || ci == y) {
            // its minimum was one of the merged entries: still the minimum
            // only if the new entry is no larger
            if (ci == y && v <= nn_d[i])
                nn_d[i] = v;
            else
                nn_i[i] = -1;
        } else if (ci >= 0) {
            const double cd = nn_d[i];
            if (v < cd || (v == cd && y < ci)) {
                nn_d[i] = v;
                nn_i[i] = y;
            }
        }
    }
    ward_reduce_b(bd, bi, r_d, r_i);
    if (threadIdx.x == 0) {
        pm_d[blockIdx.x] = bd;
        pm_i[blockIdx.x] = bi;
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
    if (n < 2) return 0;
    const long long pitch = (n + 1) & ~1ll;
    const int Gm = (int)std::min<long long>((n + WARD_B - 1) / WARD_B, 1024);
    const int Gs = (int)std::max<long long>(1, std::min<long long>(
        ((pitch >> 1) + WARD_B * 3 - 1) / (WARD_B * 3), 256));
    // the full symmetric matrix: 8 N^2 bytes (20 GB at 50 000 cells).  Not
    // fitting is the ONE failure the caller may answer with SciPy's routine
    // on the condensed vector: it gets a return code of its own (5).
    {
        size_t free_b = 0, total_b = 0;
        PCK(hipMemGetInfo(&free_b, &total_b));
        const size_t need = (size_t)n * pitch * sizeof(double)
            + (size_t)n * 72 + ((size_t)1 << 20);
        if (need > free_b) {
            bnpc_set_error("ward linkage: the %lld x %lld distance matrix "
                           "needs %.1f GB, %.1f GB of device memory are free",
                           n, n, need / 1e9, free_b / 1e9);
            return 5;
        }
    }
    double *d_D = nullptr, *d_Z = nullptr, *d_cd = nullptr, *d_nd = nullptr,
        *d_pd = nullptr;
    int *d_size = nullptr, *d_chain = nullptr, *d_ni = nullptr,
        *d_pi = nullptr;
    WardState *d_ws = nullptr;
    hipStream_t st = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    const int G = Gm + Gs;
    hipError_t e = hipMalloc((void **)&d_D, (size_t)n * pitch * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_Z, (size_t)(n - 1) * 4 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_size, (size_t)n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&d_chain, (size_t)(n + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&d_cd, (size_t)(n + 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_nd, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_ni, (size_t)n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&d_pd, (size_t)G * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_pi, (size_t)G * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&d_ws, sizeof(WardState));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemsetAsync(d_ws, 0, sizeof(WardState), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_differ_to_square, dim3(4096), dim3(256), 0, st,
                           p->differ, n, pitch, (double)p->S, d_D);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_ward_init,
                           dim3((unsigned)std::min<long long>(n, 4096)),
                           dim3(WARD_B), 0, st, d_D, n, pitch, d_size, d_nd,
                           d_ni);
        e = hipGetLastError();
    }
    // chain, work, chain, work ... on one stream: a graph of WARD_BATCH such
    // pairs, replayed until the chain reports that everything is merged (a
    // launch that finds nothing to do returns at once)
    const int WARD_BATCH = 128;
    auto pair = [&](hipStream_t s) {
        hipLaunchKernelGGL(k_ward_chain, dim3(1), dim3(64), 0, s, d_D, n,
                           pitch, d_size, d_chain, d_cd, d_nd, d_ni, d_Z,
                           d_pd, d_pi, Gm, d_pd + Gm, d_pi + Gm, Gs, d_ws);
        hipLaunchKernelGGL(k_ward_work, dim3((unsigned)G), dim3(WARD_B), 0, s,
                           d_D, n, pitch, d_size, d_nd, d_ni, d_pd, d_pi, Gm,
                           d_pd + Gm, d_pi + Gm, Gs, d_ws);
    };
    bool graphed = false;
    // (BNPC_WARD_DEVICE=plain: plain launches - rocprofv3's kernel trace does
    // not survive the replay of a captured graph on this stack)
    const char *wg = getenv("BNPC_WARD_DEVICE");
    if (e == hipSuccess && !(wg && !strcmp(wg, "plain"))
        && hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)
            == hipSuccess) {
        for (int k = 0; k < WARD_BATCH; k++) pair(st);
        if (hipStreamEndCapture(st, &graph) == hipSuccess && graph
            && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)
                == hipSuccess)
            graphed = true;
    }
    (void)hipGetLastError();
    WardState ws;
    memset(&ws, 0, sizeof ws);
    // about 2.6 pieces of work per merge (measured: the merge itself and 1.6
    // scans of stale rows); the state is looked at after every `look` batches
    long long batches = 0;
    const long long look = std::max<long long>(1, (n / WARD_BATCH) / 8);
    const long long cap = 16 * (n / WARD_BATCH + 2) + 64;
    while (e == hipSuccess && !ws.done && !ws.err && batches < cap) {
        const long long burst = batches == 0
            ? std::max<long long>(1, 2 * n / WARD_BATCH) : look;
        for (long long q = 0; q < burst && e == hipSuccess; q++) {
            if (graphed) {
                e = hipGraphLaunch(exec, st);
            } else {
                for (int k = 0; k < WARD_BATCH; k++) pair(st);
                e = hipGetLastError();
            }
        }
        batches += burst;
        if (e == hipSuccess)
            e = hipMemcpyAsync(&ws, d_ws, sizeof ws, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e == hipSuccess)
        e = hipMemcpy(Z_raw, d_Z, (size_t)(n - 1) * 4 * sizeof(double),
                      hipMemcpyDeviceToHost);
    p->ward_stats[0] = ws.scans;
    p->ward_stats[1] = ws.steps;
    const int err = ws.err || (e == hipSuccess && ws.merges != n - 1);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (st) (void)hipStreamDestroy(st);
    if (d_D) (void)hipFree(d_D);
    if (d_Z) (void)hipFree(d_Z);
    if (d_size) (void)hipFree(d_size);
    if (d_chain) (void)hipFree(d_chain);
    if (d_cd) (void)hipFree(d_cd);
    if (d_nd) (void)hipFree(d_nd);
    if (d_ni) (void)hipFree(d_ni);
    if (d_pd) (void)hipFree(d_pd);
    if (d_pi) (void)hipFree(d_pi);
    if (d_ws) (void)hipFree(d_ws);
    if (e == hipErrorOutOfMemory) {
        // an allocation that failed although the pre-check saw room
        // (fragmentation, other chains on the same GPU): the announced
        // out-of-memory code, so that the caller's host fallback applies
        (void)hipGetLastError();
        bnpc_set_error("ward linkage: out of device memory (%s)",
                       hipGetErrorString(e));
        return 5;
    }
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
