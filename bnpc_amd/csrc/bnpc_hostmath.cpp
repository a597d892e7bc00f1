// bnpc_hostmath.cpp - the per-element host arithmetic of the parameter moves,
// batched over clusters and spread over a team of host threads.
//
// What is evaluated is fixed by the reference: CRP.MH_cluster_params
// (/root/reference/libs/CRP.py:314-344) draws a truncated-normal proposal per
// mutation (SciPy: truncnorm.rvs = ppf of a uniform), and CRP._get_log_A
// (:347-383) adds its forward / reverse log-density, the Beta prior and the
// likelihood of the cluster's cells (here n1 * L1 + n0 * L0 from the device's
// column counts).  How it is evaluated is this build's: one call for all
// clusters of a step, no interpreter in the loop, threads over (cluster,
// mutation-chunk) tasks, the random draws of cluster g+1 generated while the
// team works on cluster g.
//
// Bits: every transcendental goes through the function SciPy / NumPy use
// themselves (bnpc_host_kernels, include/bnpc_hip.h); everything else is
// plain IEEE arithmetic in the dtypes and association order of the Python
// expressions (compiled with -ffp-contract=off).  The formulas follow
// scipy/stats/_continuous_distns.py (1.15): _log_gauss_mass, truncnorm_gen.
// _ppf / _logpdf, _norm_logpdf, beta_gen._logpdf, and scipy.special.logsumexp
// for two real terms.

#include <limits.h>
#include <linux/futex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "bnpc_hip.h"
#include "bnpc_internal.h"

// ---------------------------------------------------------------------------
// thread team: persistent workers, one job at a time; rebuilt after fork()
// ---------------------------------------------------------------------------
namespace {

// Workers park on a futex word that packs (job generation << 8 | ranks
// wanted); after a job they spin for a short while first, so that the batches
// of one sampler step (a few hundred microseconds apart) do not pay a kernel
// wake-up each.  No mutex on the wake-up path: all workers start in parallel.
//
// Threading contract (include/bnpc_hip.h): one job at a time.  run() holds
// the team's job lock from posting to completion, so entry points called from
// several host threads at once take turns on the team instead of corrupting
// the job word.  A job must not start another team job (it would wait for
// itself) - none does.
class Team {
public:
    Team() : pid_(getpid())
    {
        // (300 us: the ranks a converged step uses stay awake from one of its
        // batches to the next; a chain that shares its NUMA node with others
        // is given 5 by its driver - bnpc_amd.mcmc, bench.py)
        const char *e = getenv("BNPC_HOST_SPIN_US");
        spin_ns_ = (e ? atol(e) : 300) * 1000L;
    }
    // a chain whose driver gave it the short spin (it has fewer than a
    // handful of CPUs to itself) is frugal with ranks as well: every rank it
    // wakes is taken from a neighbour
    bool frugal() const { return spin_ns_ < 50000; }
    int size() const { return ranks_.load(std::memory_order_acquire); }
    pid_t pid() const { return pid_; }

    // Workers up to `ranks` ranks in all (the caller is rank 0).  The team
    // grows IN PLACE: a late starter begins with seen = 0 and takes the word
    // it finds for a job, but the tickets of a job that is over are spent (or
    // carry another generation), so it runs nothing old.  A thread the system
    // refuses (EAGAIN
    // under a pids / ulimit cap) ends the growth: the team keeps the ranks it
    // has.  Returns the size reached.
    int grow(int ranks)
    {
        std::lock_guard<std::mutex> hold(job_mu_);
        if (ranks > 255) ranks = 255;
        while (size() < ranks) {
            const int rank = size();
            try {
                threads_.emplace_back([this, rank] { loop(rank); });
            } catch (const std::system_error &) {
                break;
            }
            ranks_.store(rank + 1, std::memory_order_release);
        }
        return size();
    }

    // fn(rank) on `n` ranks (the caller is rank 0); returns when all are
    // done, with the number of ranks that ran
    int run(int n, const std::function<void(int)> &fn)
    {
        std::lock_guard<std::mutex> hold(job_mu_);
        if (n > size()) n = size();
        if (n > 255) n = 255;
        if (n <= 1) {
            fn(0);
            return 1;
        }
        job_.store(&fn, std::memory_order_relaxed);
        pending_.store(n - 1, std::memory_order_relaxed);
        gen_ = (gen_ + 1) & 0xffffff;
        // ranks 1 .. n-1 are handed out by ticket to whichever workers turn
        // up (the generation in the upper half keeps a late worker of the
        // previous job from drawing a rank of this one)
        ticket_.store(((uint64_t)gen_ << 32) | 1u, std::memory_order_relaxed);
        // store the word, THEN look for sleepers - in that order for every
        // observer (a full fence: x86 may otherwise satisfy the load before
        // the store is visible, while a worker that has just announced itself
        // still reads the old word in futex_wait, and nobody wakes it)
        word_.store((gen_ << 8) | (uint32_t)n, std::memory_order_seq_cst);
        std::atomic_thread_fence(std::memory_order_seq_cst);
        // as many sleepers as ranks are wanted, not the whole team: any worker
        // can take any rank, and a 32-thread team that is woken for a 3-rank
        // job costs 29 pointless context switches
        if (sleepers_.load(std::memory_order_seq_cst) > 0)
            syscall(SYS_futex, (uint32_t *)&word_, FUTEX_WAKE_PRIVATE, n - 1,
                    nullptr, nullptr, 0);
        fn(0);
        while (pending_.load(std::memory_order_acquire) != 0) cpu_relax();
        job_.store(nullptr, std::memory_order_relaxed);
        return n;
    }

private:
    static void cpu_relax()
    {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    static long now_ns()
    {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1000000000L + ts.tv_nsec;
    }
    void loop(int /* creation order: ranks are drawn per job */)
    {
        uint32_t seen = 0;      // the word before the first job (a worker
                                // may start after that job was posted)
        bool worked = false;    // a rank the last job did not want parks at
                                // once: only the ranks in use spin for the
                                // next job (a wake-up reaches every sleeper)
        int last_slot = 0;      // ... and only the first few of them for long:
                                // the small batches of a converged step use
                                // ranks 1-7 again within the spin; the two
                                // dozen more of a wide batch would burn a
                                // core each until the next one
        for (;;) {
            uint32_t w;
            const long t0 = now_ns();
            const long spin = !worked ? 0
                : (last_slot < 8 || spin_ns_ < 50000 ? spin_ns_ : 50000);
            int polls = 0;
            while ((w = word_.load(std::memory_order_acquire)) == seen) {
                cpu_relax();
                if ((++polls & 63) == 0 && now_ns() - t0 > spin) {
                    sleepers_.fetch_add(1, std::memory_order_seq_cst);
                    // re-checked by the kernel: returns at once if the word
                    // has moved on
                    syscall(SYS_futex, (uint32_t *)&word_, FUTEX_WAIT_PRIVATE,
                            seen, nullptr, nullptr, 0);
                    sleepers_.fetch_sub(1, std::memory_order_acq_rel);
                }
            }
            seen = w;
            // draw a rank of THIS job (generation w >> 8), if one is left
            const int n = (int)(w & 0xff);
            int slot = -1;
            uint64_t t = ticket_.load(std::memory_order_acquire);
            while ((uint32_t)(t >> 32) == (w >> 8) && (int)(uint32_t)t < n) {
                if (ticket_.compare_exchange_weak(t, t + 1,
                        std::memory_order_acq_rel)) {
                    slot = (int)(uint32_t)t;
                    break;
                }
            }
            worked = slot > 0;
            if (worked) last_slot = slot;
            if (worked) {
                // (published before the word: the acquire load above pairs
                // with the seq_cst store in run())
                (*job_.load(std::memory_order_relaxed))(slot);
                pending_.fetch_sub(1, std::memory_order_release);
            }
        }
    }

    pid_t pid_;
    long spin_ns_ = 50000;
    std::mutex job_mu_;
    std::vector<std::thread> threads_;
    std::atomic<int> ranks_{1};         // the caller + threads_.size()
    std::atomic<uint32_t> word_{0};
    std::atomic<uint64_t> ticket_{0};   // generation << 32 | next free rank
    std::atomic<int> pending_{0}, sleepers_{0};
    uint32_t gen_ = 0;
    std::atomic<const std::function<void(int)> *> job_{nullptr};
};

// The team of this PROCESS.  A forked child inherits the object but not the
// threads: it abandons the parent's team (never destroyed - its std::threads
// are not joinable there, and its job lock may be held by a thread that does
// not exist in the child) and starts its own on first use.
Team *g_team = nullptr;
std::mutex g_team_mu;

void team_after_fork_child()
{
    // the registry lock may have been held by another thread of the parent
    new (&g_team_mu) std::mutex();
    g_team = nullptr;
}

Team *team_for(int threads)
{
    static const int hooked = pthread_atfork(nullptr, nullptr,
                                             team_after_fork_child);
    (void)hooked;
    Team *t;
    {
        std::lock_guard<std::mutex> lk(g_team_mu);
        if (g_team && g_team->pid() != getpid()) g_team = nullptr;
        if (!g_team) g_team = new Team();
        t = g_team;
    }
    if (t->size() < threads) t->grow(threads);
    return t;
}

// ---------------------------------------------------------------------------
// the aside: ONE more persistent thread per process, for a job that runs NEXT
// TO the calling thread instead of with it (bnpc_aside_start / _wait,
// bnpc_internal.h): the draws of the next parameter batch taken ahead on a
// copy of the stream while the caller waits for the sweep's kernel and walks
// its loop.  Parks on a futex word after the team's spin; rebuilt after
// fork() like the team.
// ---------------------------------------------------------------------------
class Aside {
public:
    Aside() : pid_(getpid())
    {
        const char *e = getenv("BNPC_HOST_SPIN_US");
        spin_ns_ = (e ? atol(e) : 300) * 1000L;
    }
    pid_t pid() const { return pid_; }

    // post fn (copied); false: no thread to be had - nothing was posted.
    // One job at a time: the caller has waited for the previous one.
    bool start(const std::function<void()> &fn)
    {
        wait();
        if (!started_) {
            try {
                thread_ = std::thread([this] { loop(); });
            } catch (const std::system_error &) {
                return false;
            }
            started_ = true;
        }
        job_ = fn;
        running_.store(1, std::memory_order_release);
        word_.fetch_add(1, std::memory_order_seq_cst);
        std::atomic_thread_fence(std::memory_order_seq_cst);
        if (sleeping_.load(std::memory_order_seq_cst) > 0)
            syscall(SYS_futex, (uint32_t *)&word_, FUTEX_WAKE_PRIVATE, 1,
                    nullptr, nullptr, 0);
        return true;
    }

    void wait()
    {
        for (int spins = 0; running_.load(std::memory_order_acquire);
             spins++) {
            if (spins < 2048) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            } else {
                std::this_thread::yield();
            }
        }
    }

private:
    static long now_ns()
    {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1000000000L + ts.tv_nsec;
    }
    void loop()
    {
        uint32_t seen = 0;
        for (;;) {
            uint32_t w;
            const long t0 = now_ns();
            int polls = 0;
            while ((w = word_.load(std::memory_order_acquire)) == seen) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                if ((++polls & 63) == 0 && now_ns() - t0 > spin_ns_) {
                    sleeping_.fetch_add(1, std::memory_order_seq_cst);
                    syscall(SYS_futex, (uint32_t *)&word_, FUTEX_WAIT_PRIVATE,
                            seen, nullptr, nullptr, 0);
                    sleeping_.fetch_sub(1, std::memory_order_acq_rel);
                }
            }
            seen = w;
            job_();
            running_.store(0, std::memory_order_release);
        }
    }

    pid_t pid_;
    long spin_ns_ = 300000;
    bool started_ = false;
    std::thread thread_;
    std::function<void()> job_;
    std::atomic<uint32_t> word_{0};
    std::atomic<int> running_{0}, sleeping_{0};
};

Aside *g_aside = nullptr;

// (the calling thread of a chain only: no lock - and a forked child starts
// its own, the parent's object is abandoned like its team)
Aside *aside()
{
    if (g_aside && g_aside->pid() != getpid()) g_aside = nullptr;
    if (!g_aside) g_aside = new Aside();
    return g_aside;
}

}  // namespace

bool bnpc_aside_start(const std::function<void()> &fn)
{
    return aside()->start(fn);
}

void bnpc_aside_wait()
{
    if (g_aside && g_aside->pid() == getpid()) g_aside->wait();
}

// ranks a team of `threads` will really have: the team is capped at 255 and
// keeps what it has when the system refuses a thread (the team is grown here)
int bnpc_team_ranks(int threads)
{
    if (threads > 255) threads = 255;
    if (threads <= 1) return 1;
    const int have = team_for(threads)->size();
    return have < threads ? have : threads;
}

int bnpc_team_run(int threads, const std::function<void(int)> &fn)
{
    threads = bnpc_team_ranks(threads);
    if (threads <= 1) {
        fn(0);
        return 1;
    }
    return team_for(threads)->run(threads, fn);
}

extern "C" int bnpc_team_size(int threads)
{
    return bnpc_team_ranks(threads);
}

// Rows of a sample trace written for the first time - which is what makes
// their pages exist: row r of dst (dst_stride bytes apart) receives `copy`
// bytes of row r of src (src_stride apart; none if copy == 0) and zeros up to
// `width` bytes.  A contiguous block of rows per rank: fresh anonymous memory
// costs a fault and a cleared page per 4 KiB (~65 us per MB on one thread:
// 6-16 ms for config 5's 99 MB parameter trace, inside whichever step takes
// the first post-burn-in sample), and faults of different threads in
// different pages of one mapping do not wait for each other.
extern "C" int bnpc_rows_copy_zero(void *dst, int64_t dst_stride,
                                   const void *src, int64_t src_stride,
                                   int64_t rows, int64_t copy, int64_t width,
                                   int threads)
{
    if (rows < 0 || copy < 0 || width < copy || dst_stride < width
            || (rows > 0 && !dst) || (copy > 0 && (!src || src_stride < copy))) {
        bnpc_set_error("bad argument: rows_copy_zero");
        return 2;
    }
    if (rows == 0 || width == 0) return 0;
    // (a rank per 2 MiB at least: waking a rank costs what 30 pages do)
    const int64_t per_rank = (2 << 20) / width + 1;
    int64_t want = (rows + per_rank - 1) / per_rank;
    if (want > threads) want = threads;
    const int ranks = bnpc_team_ranks((int)(want < 1 ? 1 : want));
    auto work = [&](int rank) {
        const int64_t r0 = rows * rank / ranks, r1 = rows * (rank + 1) / ranks;
        for (int64_t r = r0; r < r1; r++) {
            char *d = (char *)dst + r * dst_stride;
            if (copy) memcpy(d, (const char *)src + r * src_stride, copy);
            if (width > copy) memset(d + copy, 0, width - copy);
        }
    };
    if (ranks <= 1) work(0);
    else team_for(ranks)->run(ranks, work);
    return 0;
}

extern "C" int bnpc_team_stress(int64_t jobs, int max_tasks, int ranks,
                                uint64_t seed, int64_t *done,
                                int64_t *expected)
{
    if (!done || !expected || jobs < 0 || max_tasks < 1 || ranks < 1) {
        bnpc_set_error("bad argument: team_stress");
        return 2;
    }
    std::atomic<int64_t> total(0);
    int64_t want = 0;
    uint64_t x = seed * 0x9e3779b97f4a7c15ull + 1;
    for (int64_t j = 0; j < jobs; j++) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        const int64_t tasks = 1 + (int64_t)(x % (uint64_t)max_tasks);
        const int r = 1 + (int)((x >> 32) % (uint64_t)ranks);
        std::atomic<int64_t> next(0);
        int64_t seen_by[256] = {0};
        // beside every other team job a job on the aside thread (the walker's
        // thread: posted, run next to the team, awaited), counting on its own
        int64_t beside = 0;
        const int64_t beside_n = (j & 1) ? 1 + (int64_t)(x % 97) : 0;
        bool posted = false;
        if (beside_n)
            posted = bnpc_aside_start([&beside, beside_n]() {
                for (int64_t i = 0; i < beside_n; i++) beside++;
            });
        bnpc_team_run(r, [&](int rank) {
            for (;;) {
                const int64_t t = next.fetch_add(1, std::memory_order_relaxed);
                if (t >= tasks) break;
                seen_by[rank]++;        // plain: one writer per rank
            }
        });
        if (posted) bnpc_aside_wait();
        int64_t got = posted ? beside : beside_n;
        want += beside_n;
        for (int i = 0; i < 256; i++) got += seen_by[i];
        total.fetch_add(got, std::memory_order_relaxed);
        want += tasks;
    }
    *done = total.load();
    *expected = want;
    return 0;
}

namespace {

inline void uloop(bnpc_uloop f, void *data, const double *in, double *out,
                  intptr_t n)
{
    if (n <= 0) return;
    char *args[2] = {(char *)in, (char *)out};
    intptr_t dims[1] = {n};
    intptr_t steps[2] = {(intptr_t)sizeof(double), (intptr_t)sizeof(double)};
    f(args, dims, steps, data);
}

constexpr int BLK = 128;       // elements per task
constexpr int64_t SMALL_BATCH = 4096;   // elements: draw before the team starts

struct Consts {
    double log_sd[8];          // np.log(sd[i])
    double log_m1, log_m2;     // np.log(1.0), np.log(2.0)
    double betaln_pq;
    float tmin32, tmax32;
};

// scipy _log_gauss_mass(a, b) of one interval; false: leave it to SciPy
inline bool gauss_mass(const bnpc_host_kernels *k, double a, double b,
                       double *out)
{
    if (a <= 0.0 && b > 0.0) {          // case_central
        *out = k->sc_log1p(-k->ndtr(a, 0) - k->ndtr(-b, 0), 0);
        return true;
    }
    if (b <= 0.0 && k->left_ok) {       // case_left: _log_diff of the cdfs
        *out = bnpc_log_diff_pi1(k->log_ndtr(b, 0), k->log_ndtr(a, 0));
        return true;
    }
    return false;                       // case_right / NaN
}

// beta._logpdf + rv_continuous.logpdf's support mask, x float32 -> float64
inline double beta_logpdf1(const bnpc_host_kernels *k, float x32, double p,
                           double q, double betaln_pq)
{
    const double x = (double)x32;
    if (!((0.0 < x) && (x < 1.0))) return -INFINITY;
    double l = k->xlog1py(q - 1.0, -x, 0) + k->xlogy(p - 1.0, x, 0);
    l -= betaln_pq;
    return l;
}

// n elements of cluster g - positions idx[0..n) of its row, or, idx == NULL,
// the run [m0, m0 + n); false: an element needs SciPy's own path
// accept_known: the device screen has established that these proposals are
// accepted (include/bnpc_hip.h, screen == 2): the proposal's bits and the
// prior density of the result are all that is evaluated
bool mh_block(const bnpc_host_kernels *k, const bnpc_mh_args *a,
              const Consts &c, int64_t g, const int32_t *idx, int64_t m0,
              int n, bool accept_known = false)
{
    const size_t row = (size_t)g * a->M;
    // gathered inputs (a dense run is gathered too: one code path)
    float old[BLK];
    int32_t n1[BLK], n0[BLK], si[BLK], at[BLK];
    double U[BLK], lu[BLK];
    for (int i = 0; i < n; i++) {
        const int32_t m = idx ? idx[i] : (int32_t)(m0 + i);
        at[i] = m;
        old[i] = a->old_theta[row + m];
        n1[i] = a->n1[row + m];
        n0[i] = a->n0[row + m];
        si[i] = a->sd_idx[row + m];
        U[i] = a->U[row + m];
        lu[i] = a->u[row + m];
    }

    double std_[BLK], lsd[BLK], lo[BLK], hi[BLK], lgm[BLK];
    double t0[BLK], t1[BLK], t2[BLK], t3[BLK];
    float nw[BLK];

    // a = (TMIN - old) / std, b = (TMAX - old) / std: the difference is
    // float32 arithmetic (a Python float against a float32 array)
    for (int i = 0; i < n; i++) {
        std_[i] = a->sd[si[i]];
        lsd[i] = c.log_sd[si[i]];
        lo[i] = (double)(c.tmin32 - old[i]) / std_[i];
        hi[i] = (double)(c.tmax32 - old[i]) / std_[i];
        if (!gauss_mass(k, lo[i], hi[i], &lgm[i])) return false;
        if (!(lu[i] > 0.0) || !(U[i] > 0.0)) return false;  // log(0) raises
    }

    // truncnorm._ppf: left form where a < 0, mirrored form elsewhere
    //   log_Phi = logsumexp([logcdf(a), log(q) + mass])       (a < 0)
    //   log_Phi = logsumexp([logcdf(-b), log1p(-q) + mass])   (a >= 0)
    uloop(k->np_log, k->np_log_data, U, t0, n);             // log(q)
    int nr = 0;
    int ridx[BLK];
    for (int i = 0; i < n; i++)
        if (!(lo[i] < 0.0)) {
            ridx[nr] = i;
            t1[nr++] = -U[i];
        }
    uloop(k->np_log1p, k->np_log1p_data, t1, t2, nr);       // log1p(-q)
    for (int j = 0; j < nr; j++) t0[ridx[j]] = t2[j];
    // p -> t1, q -> t0; top -> t2, min - top -> t3
    bool tie[BLK];
    for (int i = 0; i < n; i++) {
        const bool left = lo[i] < 0.0;
        const double pp = k->log_ndtr(left ? lo[i] : -hi[i], 0);
        const double qq = t0[i] + lgm[i];
        if (!isfinite(pp) || !isfinite(qq)) return false;
        const double top = pp > qq ? pp : qq;
        const double bot = pp > qq ? qq : pp;
        tie[i] = pp == qq;
        t2[i] = top;
        t3[i] = bot - top;
    }
    uloop(k->np_exp, k->np_exp_data, t3, t1, n);
    for (int i = 0; i < n; i++)
        if (tie[i]) t1[i] = 0.0;
    uloop(k->np_log1p, k->np_log1p_data, t1, t3, n);
    for (int i = 0; i < n; i++) {
        const double log_phi = t3[i] + (tie[i] ? c.log_m2 : c.log_m1) + t2[i];
        double x = k->ndtri_exp(log_phi, 0);
        if (!(lo[i] < 0.0)) x = -x;
        x = x * std_[i] + (double)old[i];
        nw[i] = (float)x;                                   // astype(float32)
    }
    if (accept_known) {
        float *out = a->new_theta + row;
        double *prior_out = a->prior_out ? a->prior_out + row : nullptr;
        for (int i = 0; i < n; i++) {
            out[at[i]] = nw[i];
            if (prior_out)
                prior_out[at[i]] = a->uniform_prior ? 0.0
                    : beta_logpdf1(k, nw[i], a->p, a->q, c.betaln_pq);
        }
        return true;
    }

    // forward and reverse proposal log-densities (truncnorm._logpdf around
    // old / around new), reverse bounds from the float32 proposal
    double fwd[BLK], rev[BLK];
    for (int i = 0; i < n; i++) {
        const double xs = (double)(nw[i] - old[i]) / std_[i];
        double v = -(xs * xs) / 2.0 - k->norm_pdf_logC;
        v = v - lgm[i] - lsd[i];
        fwd[i] = ((lo[i] <= xs) && (xs <= hi[i])) ? v : -INFINITY;

        const double ar = (double)(c.tmin32 - nw[i]) / std_[i];
        const double br = (double)(c.tmax32 - nw[i]) / std_[i];
        double mass_r;
        if (!gauss_mass(k, ar, br, &mass_r)) return false;
        const double xr = (double)(old[i] - nw[i]) / std_[i];
        double w = -(xr * xr) / 2.0 - k->norm_pdf_logC;
        w = w - mass_r - lsd[i];
        rev[i] = ((ar <= xr) && (xr <= br)) ? w : -INFINITY;
    }

    // likelihood of the cluster's cells under new / old: n1 * L1 + n0 * L0,
    // L1 = log(t * (1 - FN) + (1 - t) * FP), L0 = log(t * FN + (1 - t) *
    // (1 - FP)); (1 - t) in float32 (libs/CRP.py:198-200, 207-212)
    const double pFN1 = 1.0 - a->FN, pFP0 = 1.0 - a->FP;
    double ll_new[BLK], ll_old[BLK];
    for (int pass = 0; pass < 2; pass++) {
        const float *th = pass == 0 ? nw : old;
        for (int i = 0; i < n; i++) {
            const double t64 = (double)th[i];
            const double om64 = (double)(1.0f - th[i]);
            t0[i] = t64 * pFN1 + om64 * a->FP;
            t1[i] = t64 * a->FN + om64 * pFP0;
        }
        uloop(k->np_log, k->np_log_data, t0, t2, n);
        uloop(k->np_log, k->np_log_data, t1, t3, n);
        double *dst = pass == 0 ? ll_new : ll_old;
        for (int i = 0; i < n; i++)
            dst[i] = (double)n1[i] * t2[i] + (double)n0[i] * t3[i];
    }

    // A = new_ll + new_prior - old_ll - old_prior + rev - fwd
    double pr_new[BLK], pr_old[BLK];
    double A[BLK];
    for (int i = 0; i < n; i++) {
        if (a->uniform_prior) {
            pr_new[i] = pr_old[i] = 0.0;
        } else {
            pr_new[i] = beta_logpdf1(k, nw[i], a->p, a->q, c.betaln_pq);
            if (a->known_theta
                && !memcmp(a->known_theta + row + at[i], old + i,
                           sizeof(float)))
                pr_old[i] = a->known_prior[row + at[i]];
            else
                pr_old[i] = beta_logpdf1(k, old[i], a->p, a->q, c.betaln_pq);
        }
        double v = ll_new[i] + pr_new[i];
        v = v - ll_old[i];
        v = v - pr_old[i];
        v = v + rev[i];
        v = v - fwd[i];
        if (a->trans_prob && v > 0.0) v = 0.0;              // np.clip(max=0)
        A[i] = v;
    }

    // decline = log(u) >= A
    uloop(k->np_log, k->np_log_data, lu, t0, n);
    int nd = 0;
    int didx[BLK];
    float *out = a->new_theta + row;
    double *prior_out = a->prior_out ? a->prior_out + row : nullptr;
    for (int i = 0; i < n; i++) {
        const bool decline = t0[i] >= A[i];
        out[at[i]] = decline ? old[i] : nw[i];
        if (prior_out) prior_out[at[i]] = decline ? pr_old[i] : pr_new[i];
        if (decline) {
            didx[nd] = i;
            t1[nd++] = A[i];
        }
    }
    if (a->trans_prob && nd) {
        // A[declined] = log(-1 * expm1(A[declined]))
        uloop(k->np_expm1, k->np_expm1_data, t1, t2, nd);
        for (int j = 0; j < nd; j++) {
            t2[j] = -1.0 * t2[j];
            if (!(t2[j] > 0.0)) return false;               // log(<= 0) raises
        }
        uloop(k->np_log, k->np_log_data, t2, t3, nd);
        for (int j = 0; j < nd; j++) A[didx[j]] = t3[j];
    }
    double *A_out = a->A + row;
    for (int i = 0; i < n; i++) A_out[at[i]] = A[i];
    __atomic_fetch_add(&a->declined[g], (int64_t)nd, __ATOMIC_RELAXED);
    return true;
}

}  // namespace

// What rank 0 of the NEXT screened bnpc_mh_batch of this thread does before it
// joins the evaluation (bnpc_internal.h); consumed by that call.
static thread_local const std::function<void()> *g_rank0_hook = nullptr;

void bnpc_mh_rank0_hook(const std::function<void()> *hook)
{
    g_rank0_hook = hook;
}

// What every rank of the NEXT screened bnpc_mh_batch of this thread calls
// before it touches a row (bnpc_internal.h): it returns when the row's flags
// are there.  false: give the batch up.  Consumed by that call.
static thread_local const std::function<bool(int64_t)> *g_row_gate = nullptr;

void bnpc_mh_row_gate(const std::function<bool(int64_t)> *gate)
{
    g_row_gate = gate;
}

static int check_kernels(const bnpc_host_kernels *k)
{
    if (!k || !k->ndtr || !k->log_ndtr || !k->ndtri_exp || !k->sc_log1p
        || !k->xlogy || !k->xlog1py || !k->betaln || !k->np_log || !k->np_exp
        || !k->np_log1p || !k->np_expm1) {
        bnpc_set_error("bad argument: host kernel table incomplete");
        return 2;
    }
    return 0;
}

extern "C" int bnpc_mh_batch(const bnpc_host_kernels *k, bnpc_mt19937 *rng,
                             const bnpc_mh_args *a, int *status)
{
    if (check_kernels(k)) return 2;
    if (!a || !status || a->G < 0 || a->M < 1 || !a->old_theta || !a->n1
        || !a->n0 || !a->sd || a->n_sd < 1 || a->n_sd > 8 || !a->sd_idx
        || !a->U || !a->u || !a->new_theta || !a->A || !a->log_prob
        || !a->declined || (a->known_theta && !a->known_prior)) {
        bnpc_set_error("bad argument: mh_batch");
        return 2;
    }
    *status = 0;
    const int64_t G = a->G, M = a->M;
    if (G == 0) return 0;

    Consts c;
    {
        double one_two[2] = {1.0, 2.0}, lg[2];
        uloop(k->np_log, k->np_log_data, one_two, lg, 2);
        c.log_m1 = lg[0];
        c.log_m2 = lg[1];
        double sd[8], lsd[8];
        for (int i = 0; i < a->n_sd; i++) sd[i] = a->sd[i];
        uloop(k->np_log, k->np_log_data, sd, lsd, a->n_sd);
        for (int i = 0; i < a->n_sd; i++) c.log_sd[i] = lsd[i];
        c.betaln_pq = a->uniform_prior ? 0.0 : k->betaln(a->p, a->q, 0);
        c.tmin32 = (float)a->tmin;
        c.tmax32 = (float)a->tmax;
    }

    for (int64_t g = 0; g < G; g++) a->declined[g] = 0;
    // BNPC_TIMING=mh: phase times of every parameter batch on stderr
    static const bool trace = [] {
        const char *e = getenv("BNPC_TIMING");
        return e && strstr(e, "mh");
    }();
    timespec ts0;
    long t_draws = 0;
    if (trace) clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto since = [&]() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (ts.tv_sec - ts0.tv_sec) * 1000000000L
            + (ts.tv_nsec - ts0.tv_nsec);
    };
    std::atomic<int> bail(0);
    int threads = a->threads;

    if (a->screen && !a->trans_prob) {
        // The device has screened the batch (bnpc_mh_screen): screen[g, m]
        // == 0 means "declined for certain" - the approximate log acceptance
        // ratio lies below log(u) by more than every error it can carry - so
        // the element keeps its old value and none of its transcendentals is
        // evaluated; everything else (likely accepted, in doubt, exotic) goes
        // through the exact arithmetic below, a few per cent of the batch.
        if (rng) {
            bnpc_set_error("bad argument: a screened batch brings its draws");
            return 2;
        }
        const bool want_prior = a->prior_out && !a->uniform_prior;
        // One pass per row used to touch every element (copy, cache compare,
        // branch on its flag: 4 ns each); then the declined majority went by
        // block copies, the flagged minority was found eight flags at a time
        // and the cache misses by a compare that all but never fires - 2.4 ns
        // each, but still ONE thread's pass in front of the team's arithmetic:
        // 83 of the 135 us of a 7 x 5000 part of a config-5 batch.  Now the
        // team does both: a task is a SEGMENT of a row - its copies, its
        // compare, its flags, and the exact arithmetic of what it finds
        // flagged (lists on the stack, no shared list, no second phase).
        constexpr int64_t SEG = 256;
        // how many elements are flagged (what the thread count goes by): eight
        // flags at a time
        int64_t flagged_all = a->flagged_estimate;
        if (flagged_all <= 0) {
            flagged_all = 0;
            const uint8_t *sc = a->screen;
            const int64_t E = G * M;
            int64_t e = 0, light = 0;   // light: flag 3 with the device's bits
            const uint64_t lo7 = 0x7f7f7f7f7f7f7f7full;
            for (; e + 8 <= E; e += 8) {
                uint64_t w;
                memcpy(&w, sc + e, 8);
                if (!w) continue;
                // (the high bit of every non-zero byte; of every byte == 3)
                const uint64_t x = w ^ 0x0303030303030303ull;
                flagged_all += __builtin_popcountll(
                    (((w & lo7) + lo7) | w) & ~lo7);
                light += __builtin_popcountll(~(((x & lo7) + lo7) | x | lo7));
            }
            for (; e < E; e++) {
                flagged_all += sc[e] != 0;
                light += sc[e] == 3;
            }
            // (a value taken and one density - two special-function calls
            // against ten: measured ~60 against ~300 ns - counted as half an
            // entry: the segment's own pass, copies and compare come on top)
            if (a->screen_theta) flagged_all -= light / 2;
        }
        const int64_t segs = (M + SEG - 1) / SEG;
        const int64_t n_tasks = G * segs;
        // a thread per ~12 blocks of 64 flagged elements: waking a parked
        // team costs more than several hundred elements
        // (4 gave 130 against 180 us per update_parameters under the
        // profiler and nothing on the bench line at config 3, whose batches
        // leave a dozen blocks; the parts of a config-5 batch leave 100 each,
        // the scans of its moves 20, and are bound by this arithmetic: a rank
        // per 4 blocks from 16 blocks on - config 5 267 -> 282 steps/s, config
        // 4 1177 -> 1262; from 8 on, or a rank per 3: no further gain)
        const int64_t blocks = (flagged_all + BLK / 2 - 1) / (BLK / 2);
        // (32 blocks and more are the parts of a pipelined batch: the team
        // is awake, spinning between the parts, and rank 0 spends the part
        // issuing the next but one - the others should be done when it
        // joins: a rank per 2 blocks)
        // The small batches (config 3's parameter update: a dozen blocks; the
        // scans of a move: half a dozen) take a rank per 2 blocks as well
        // when this chain has the host's cores to itself - with the team
        // spinning 300 us between jobs, i.e. across a whole converged step
        // (Team::Team) - and a rank per 12 when it has fewer than
        // GREEDY_MIN_CPUS logical CPUs of its NUMA node to itself
        // (bnpc_amd._lib.host_settings: the short spin, Team::frugal).
        // Round 5, config 3, five interleaved runs on one box: 2054-2147
        // steps/s (median 2140) with a rank per 12 blocks and 50 us of
        // spinning, 2107-2275 (median 2234) with a rank per 2 and 300 us, at
        // 3.4 busy threads instead of 1.4; a rank per 3 and 150 us: 2128.
        // Three interleaved runs each on a loaded box (load average 34), new
        // against old: config 3 2145 / 2098, config 4 1518 / 1457, config 5
        // 314 / 299 (means).
        // (read where the team reads its spin: once per process, again in a
        // forked child - the drivers set both before the first batch)
        const int64_t per_small = team_for(1)->frugal() ? 12 : 2;
        const int64_t per = blocks >= 32 ? 2 : (blocks >= 16 ? 4 : per_small);
        if (threads > (blocks + per - 1) / per)
            threads = (int)((blocks + per - 1) / per);
        if (threads > n_tasks) threads = (int)n_tasks;
        if (threads < 1) threads = 1;
        std::atomic<int64_t> next(0), n_todo(0), n_sure(0), n_miss(0),
            n_given(0);
        const std::function<bool(int64_t)> *gate = g_row_gate;
        g_row_gate = nullptr;
        // the sum of the batch's prior densities in index order
        // (bnpc_mh_args.prior_seq_sum): segments still missing per row; the
        // LAST rank adds a row up as soon as it and all rows before it are
        // complete - one chain of dependent adds that trails the team instead
        // of following it
        const bool want_sum = a->prior_seq_sum && a->prior_out;
        std::unique_ptr<std::atomic<int>[]> row_left;
        if (want_sum) {
            row_left.reset(new std::atomic<int>[(size_t)G]);
            for (int64_t g = 0; g < G; g++)
                row_left[(size_t)g].store((int)segs, std::memory_order_relaxed);
        }
        double seq_sum = want_sum ? *a->prior_seq_sum : 0.0;
        bool seq_started = want_sum && seq_sum == seq_sum;
        int64_t seq_next = 0;           // (restartable: rows [0, seq_next) are in)
        auto add_rows_up = [&]() {
            // (uniform prior: prior_out holds zeros - the sum of zeros)
            for (int64_t g = seq_next; g < G; g++) {
                for (long spins = 0; row_left[(size_t)g].load(
                         std::memory_order_acquire) != 0; spins++) {
                    if (bail.load(std::memory_order_relaxed)) return;
                    if (spins < 2000) {
#if defined(__x86_64__)
                        __builtin_ia32_pause();
#endif
                    } else {
                        std::this_thread::yield();
                    }
                }
                const double *d = a->prior_out + (size_t)g * M;
                int64_t m = 0;
                if (!seq_started) {
                    seq_sum = d[0];
                    seq_started = true;
                    m = 1;
                }
                double sacc = seq_sum;
                for (; m < M; m++) sacc += d[m];
                seq_sum = sacc;
                seq_next = g + 1;
            }
        };
        // (the ranks there really are: the team keeps what it has when the
        // system refuses a thread)
        if (want_sum && threads > 2) threads = bnpc_team_ranks(threads);
        const int sum_rank = want_sum && threads > 2 ? threads - 1 : -1;
        auto work = [&](int rank) {
            if (rank == sum_rank) {
                add_rows_up();
                return;
            }
            int32_t todo[SEG], sure[SEG], given[SEG];
            const bool take_given = a->screen_theta != nullptr;
            for (;;) {
                const int64_t t = next.fetch_add(1, std::memory_order_relaxed);
                if (t >= n_tasks) break;
                if (bail.load(std::memory_order_relaxed)) continue;
                const int64_t g = t / segs, m0 = (t - g * segs) * SEG;
                const int64_t m1 = std::min(M, m0 + SEG);
                const size_t row = (size_t)g * M;
                // (a batch whose parts are still being screened: the row's
                // verdicts may not be there yet)
                if (gate && !(*gate)(g)) {
                    bail.store(1, std::memory_order_relaxed);
                    continue;
                }
                struct RowDone {        // this segment of row g is complete
                    std::atomic<int> *left;
                    ~RowDone()
                    {
                        if (left) left->fetch_sub(1, std::memory_order_release);
                    }
                } row_done = {want_sum ? &row_left[(size_t)g] : nullptr};
                const uint8_t *sc = a->screen + row;
                const float *old = a->old_theta + row;
                float *out = a->new_theta + row;
                memcpy(out + m0, old + m0, (size_t)(m1 - m0) * sizeof(float));
                int nt = 0, ns = 0, ng = 0;
                auto sort_out = [&](int64_t at) {
                    const uint8_t f = sc[at];
                    if (f == 3 && take_given) given[ng++] = (int32_t)at;
                    else if (f == 2 || f == 3) sure[ns++] = (int32_t)at;
                    else if (f) todo[nt++] = (int32_t)at;
                };
                int64_t m = m0;
                for (; m + 8 <= m1; m += 8) {
                    uint64_t w;
                    memcpy(&w, sc + m, 8);
                    if (!w) continue;
                    for (int b = 0; b < 8; b++) sort_out(m + b);
                }
                for (; m < m1; m++) sort_out(m);
                int64_t missed = 0;
                if (want_prior) {
                    double *po = a->prior_out + row;
                    if (a->known_theta) {
                        // entries whose cached parameter has the bits of the
                        // old one take the cached density; the others
                        // (declined ones only: the flagged get theirs from
                        // the evaluation) are evaluated.  64 bytes at a time:
                        // nearly every entry of a running chain is a hit.
                        memcpy(po + m0, a->known_prior + row + m0,
                               (size_t)(m1 - m0) * sizeof(double));
                        const float *kt = a->known_theta + row;
                        int64_t i = m0;
                        for (; i + 16 <= m1; i += 16) {
                            if (!memcmp(kt + i, old + i, 64)) continue;
                            for (int64_t j = i; j < i + 16; j++)
                                if (memcmp(kt + j, old + j, 4) && !sc[j]) {
                                    po[j] = beta_logpdf1(k, old[j], a->p,
                                                         a->q, c.betaln_pq);
                                    missed++;
                                }
                        }
                        for (; i < m1; i++)
                            if (memcmp(kt + i, old + i, 4) && !sc[i]) {
                                po[i] = beta_logpdf1(k, old[i], a->p, a->q,
                                                     c.betaln_pq);
                                missed++;
                            }
                    } else {
                        for (int64_t i = m0; i < m1; i++)
                            if (!sc[i]) {
                                po[i] = beta_logpdf1(k, old[i], a->p, a->q,
                                                     c.betaln_pq);
                                missed++;
                            }
                    }
                } else if (a->prior_out) {
                    // uniform prior: the density is 0 everywhere
                    memset(a->prior_out + row + m0, 0,
                           (size_t)(m1 - m0) * sizeof(double));
                }
                __atomic_fetch_add(&a->declined[g],
                                   (int64_t)(m1 - m0 - nt - ns - ng),
                                   __ATOMIC_RELAXED);
                // accepted, and the device vouches for the proposal's bits:
                // the value is taken, its prior density evaluated
                {
                    const float *dev_new = take_given ? a->screen_theta + row
                                                      : nullptr;
                    double *po = want_prior ? a->prior_out + row : nullptr;
                    for (int i = 0; i < ng; i++) {
                        const float v = dev_new[given[i]];
                        out[given[i]] = v;
                        if (po)
                            po[given[i]] = beta_logpdf1(k, v, a->p, a->q,
                                                        c.betaln_pq);
                    }
                }
                bool ok = true;
                for (int lo = 0; lo < nt && ok; lo += BLK / 2)
                    ok = mh_block(k, a, c, g, todo + lo, 0,
                                  std::min(BLK / 2, nt - lo), false);
                for (int lo = 0; lo < ns && ok; lo += BLK / 2)
                    ok = mh_block(k, a, c, g, sure + lo, 0,
                                  std::min(BLK / 2, ns - lo), true);
                if (!ok) bail.store(1, std::memory_order_relaxed);
                if (nt) n_todo.fetch_add(nt, std::memory_order_relaxed);
                if (ns) n_sure.fetch_add(ns, std::memory_order_relaxed);
                if (ng) n_given.fetch_add(ng, std::memory_order_relaxed);
                if (trace)
                    n_miss.fetch_add(missed, std::memory_order_relaxed);
            }
        };
        const long t_prep = trace ? since() : 0;
        const std::function<void()> *first = g_rank0_hook;
        g_rank0_hook = nullptr;
        if (threads > 1) {
            if (first) {
                const std::function<void(int)> both = [&](int rank) {
                    if (rank == 0) (*first)();
                    work(rank);
                };
                team_for(threads)->run(threads, both);
            } else {
                team_for(threads)->run(threads, work);
            }
        } else {
            if (first) (*first)();
            work(0);
        }
        if (want_sum && !bail.load()) {
            // (a team of one or two - or a sum rank that never came: after
            // the work, by the caller)
            add_rows_up();
            *a->prior_seq_sum = seq_sum;
        }
        if (trace)
            fprintf(stderr, "[mh_batch] G=%lld M=%lld screened: %lld in doubt, "
                    "%lld accepted (+ %lld with the device's bits) of %lld "
                    "elements, %lld prior misses, "
                    "threads=%d, %.1f us (counting the flags %.1f)\n",
                    (long long)G, (long long)M, (long long)n_todo.load(),
                    (long long)n_sure.load(), (long long)n_given.load(),
                    (long long)(G * M), (long long)n_miss.load(), threads,
                    since() / 1e3, t_prep / 1e3);
        if (a->flag_counts) {
            a->flag_counts[0] = n_todo.load();
            a->flag_counts[1] = n_sure.load();
            a->flag_counts[2] = n_given.load();
        }
        if (bail.load()) {
            *status = 1;
            return 0;
        }
        for (int64_t g = 0; g < G; g++) a->log_prob[g] = NAN;
        return 0;
    }

    // tasks of 128 elements; of 64 in a small batch, so that 16 ranks get
    // three even rounds out of 3 x 1000 elements instead of one and a half
    // (3 x 1000 on 16 ranks: 72-77 us with tasks of 128, 67-68 us with 64;
    // 1 x 1000: 45 against 33 us)
    const int64_t blk = G * M <= SMALL_BATCH ? BLK / 2 : BLK;
    const int64_t chunks = (M + blk - 1) / blk;
    const int64_t tasks = G * chunks;
    // a rank per 2 tasks (4 until a move became one native call: the batches
    // of a move now follow each other within the team's spin, and a rank per
    // 2 tasks is worth 10-25 us per move; waking a PARKED rank still costs as
    // much as 256-512 entries - 2000 entries on 31 ranks were no faster than
    // on one when the team was asleep between the batches of a step)
    const int64_t dense_per = 2;
    if (threads > (tasks + dense_per - 1) / dense_per)
        threads = (int)((tasks + dense_per - 1) / dense_per);
    if (threads < 1) threads = 1;

    std::atomic<int64_t> next(0), rows_ready(rng ? 0 : G);

    // the draws, cluster by cluster in the reference's order
    auto draw_rows = [&]() {
        for (int64_t g = 0; g < G; g++) {
            int32_t *si = a->sd_idx + g * M;
            double *Ug = a->U + g * M, *ug = a->u + g * M;
            mt_fill_interval32(rng, (uint32_t)(a->n_sd - 1), si, M);
            mt_fill_double(rng, Ug, M);         // uniform(0, 1) == sample
            mt_fill_double(rng, ug, M);
            rows_ready.store(g + 1, std::memory_order_release);
        }
        if (trace) t_draws = since();
    };
    // A small batch (the 2-3 rows of a restricted scan: 10 us of draws) is
    // drawn before the team is started: ranks that wait for rows while rank 0
    // draws them slow the drawing thread down by more than the overlap gains
    // (3 x 1000 on 16 ranks: 96 us against 60 us with the draws given).  In a
    // large batch the team starts on a cluster as soon as its draws are
    // published.
    const bool draw_first = rng && G * M <= SMALL_BATCH;
    if (draw_first) draw_rows();

    auto work = [&](int rank) {
        if (rank == 0 && rng && !draw_first) draw_rows();
        for (;;) {
            const int64_t t = next.fetch_add(1, std::memory_order_relaxed);
            if (t >= tasks) break;
            const int64_t g = t / chunks, ch = t - g * chunks;
            // the draws of this cluster are a few microseconds away at most
            // (a short spin; beyond it the waiters get out of the way of the
            // thread that draws - they may share its core)
            for (int spins = 0;
                 rows_ready.load(std::memory_order_acquire) <= g; spins++) {
                if (spins < 256) {
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                } else {
                    std::this_thread::yield();
                }
            }
            if (bail.load(std::memory_order_relaxed)) continue;
            const int64_t m0 = ch * blk;
            const int64_t m1 = m0 + blk < M ? m0 + blk : M;
            if (!mh_block(k, a, c, g, nullptr, m0, (int)(m1 - m0)))
                bail.store(1, std::memory_order_relaxed);
        }
    };
    if (threads > 1)
        team_for(threads)->run(threads, work);
    else
        work(0);

    if (trace)
        fprintf(stderr, "[mh_batch] G=%lld M=%lld threads=%d: draws done at "
                "%.1f us, all done at %.1f us\n", (long long)G, (long long)M,
                threads, t_draws / 1e3, since() / 1e3);
    if (bail.load()) {
        *status = 1;
        return 0;
    }
    for (int64_t g = 0; g < G; g++) {
        const double *A = a->A + g * M;
        double s = 0.0;
        if (a->trans_prob) {
            s = A[0];
            for (int64_t m = 1; m < M; m++) s += A[m];      // np.cumsum order
        }
        a->log_prob[g] = a->trans_prob ? s : NAN;
    }
    return 0;
}

extern "C" int bnpc_beta_logpdf_f32(const bnpc_host_kernels *k, const float *x,
                                    int64_t n, double p, double q,
                                    const float *known_theta,
                                    const double *known_prior, double *out,
                                    double *seq_sum, int threads)
{
    if (check_kernels(k)) return 2;
    if (n < 0 || (n > 0 && (!x || !out)) || (known_theta && !known_prior)) {
        bnpc_set_error("bad argument: beta_logpdf_f32");
        return 2;
    }
    const double bl = k->betaln(p, q, 0);
    const int64_t chunks = (n + 4 * BLK - 1) / (4 * BLK);
    if (threads > chunks) threads = (int)chunks;
    // cache hits are a compare and a copy: not worth waking anybody
    int64_t misses = 0;
    if (known_theta) {
        int64_t i = 0;
        for (; i + 16 <= n; i += 16) {      // 64 bytes at a time
            if (!memcmp(known_theta + i, x + i, 64)) continue;
            for (int64_t j = i; j < i + 16; j++)
                misses += memcmp(known_theta + j, x + j, sizeof(float)) != 0;
        }
        for (; i < n; i++)
            misses += memcmp(known_theta + i, x + i, sizeof(float)) != 0;
    } else {
        misses = n;
    }
    // (large arrays go to the team anyway: the pass is memory traffic)
    if (misses < 4096 && n < 65536) threads = 1;
    std::atomic<int64_t> next(0);
    auto work = [&](int) {
        for (;;) {
            const int64_t t = next.fetch_add(1, std::memory_order_relaxed);
            if (t >= chunks) break;
            const int64_t i1 = (t + 1) * 4 * BLK < n ? (t + 1) * 4 * BLK : n;
            int64_t i = t * 4 * BLK;
            if (known_theta)
                for (; i + 16 <= i1; i += 16) {
                    if (!memcmp(known_theta + i, x + i, 64)) {
                        memcpy(out + i, known_prior + i, 16 * sizeof(double));
                        continue;
                    }
                    for (int64_t j = i; j < i + 16; j++)
                        out[j] = !memcmp(known_theta + j, x + j, sizeof(float))
                            ? known_prior[j] : beta_logpdf1(k, x[j], p, q, bl);
                }
            for (; i < i1; i++) {
                if (known_theta
                    && !memcmp(known_theta + i, x + i, sizeof(float)))
                    out[i] = known_prior[i];
                else
                    out[i] = beta_logpdf1(k, x[i], p, q, bl);
            }
        }
    };
    if (threads > 1)
        team_for(threads)->run(threads, work);
    else
        work(0);
    if (seq_sum) {
        double s = 0.0;
        if (n > 0) {
            s = out[0];
            for (int64_t i = 1; i < n; i++) s += out[i];
        }
        *seq_sum = s;
    }
    return 0;
}

// truncnorm.logpdf(x, a, b, loc, scale) for scalars (the error-rate moves,
// libs/CRP_learning_errors.py:85-91, and their priors :47-49): SciPy's
// _norm_logpdf((x - loc) / scale) - _log_gauss_mass(a, b) - log(scale) with
// the support mask of the public wrapper.  *status = 1: an interval this
// library leaves to SciPy.
extern "C" int bnpc_tn_logpdf_scalar(const bnpc_host_kernels *k, double x,
                                     double a, double b, double loc,
                                     double scale, double *out, int *status)
{
    if (check_kernels(k)) return 2;
    if (!out || !status) {
        bnpc_set_error("bad argument: tn_logpdf_scalar");
        return 2;
    }
    *status = 0;
    double mass;
    if (a > 0.0 && k->left_ok) {
        // case_right: the mirrored interval left of zero
        mass = bnpc_log_diff_pi1(k->log_ndtr(-a, 0), k->log_ndtr(-b, 0));
    } else if (!gauss_mass(k, a, b, &mass)) {
        *status = 1;
        return 0;
    }
    if (!(scale > 0.0) || !isfinite(mass)) {
        *status = 1;
        return 0;
    }
    const double xs = (x - loc) / scale;
    double log_scale;
    uloop(k->np_log, k->np_log_data, &scale, &log_scale, 1);
    double v = -(xs * xs) / 2.0 - k->norm_pdf_logC;
    v = v - mass;
    v = v - log_scale;
    *out = ((a <= xs) && (xs <= b)) ? v : -INFINITY;
    return 0;
}

// ---------------------------------------------------------------------------
// CRP._get_log_A (libs/CRP.py:347-383) for GIVEN new / old parameter rows and
// proposal standard deviations - the transition-probability terms of the
// split / merge ratios (libs/CRP.py:674-681, 777-799), which score a move to
// parameters that already exist instead of proposing some.  Same arithmetic
// as the batch above minus the draws, the ppf and the accept step.
// ---------------------------------------------------------------------------
namespace {

bool logA_block(const bnpc_host_kernels *k, const bnpc_accept_args *a,
                double betaln_pq, int64_t g, int64_t m0, int64_t m1)
{
    const int n = (int)(m1 - m0);
    const size_t off = (size_t)g * a->M + m0;
    const float *nw = a->new_theta + off, *old = a->old_theta + off;
    const double *sd = a->std + off;
    const int32_t *n1 = a->n1 + off, *n0 = a->n0 + off;
    const float fmin32 = (float)a->fmin, fmax32 = (float)a->fmax;
    const float tmin32 = (float)a->tmin, tmax32 = (float)a->tmax;
    double lsd[BLK], t0[BLK], t1[BLK], t2[BLK], t3[BLK];
    double fwd[BLK], rev[BLK], ll_new[BLK], ll_old[BLK];
    uloop(k->np_log, k->np_log_data, sd, lsd, n);
    for (int i = 0; i < n; i++) {
        if (!(sd[i] > 0.0)) return false;
        const double lo = (double)(fmin32 - old[i]) / sd[i];
        const double hi = (double)(fmax32 - old[i]) / sd[i];
        double mass;
        if (!gauss_mass(k, lo, hi, &mass)) return false;
        const double xs = (double)(nw[i] - old[i]) / sd[i];
        double v = -(xs * xs) / 2.0 - k->norm_pdf_logC;
        v = v - mass - lsd[i];
        fwd[i] = ((lo <= xs) && (xs <= hi)) ? v : -INFINITY;

        const double ar = (double)(tmin32 - nw[i]) / sd[i];
        const double br = (double)(tmax32 - nw[i]) / sd[i];
        double mass_r;
        if (!gauss_mass(k, ar, br, &mass_r)) return false;
        const double xr = (double)(old[i] - nw[i]) / sd[i];
        double w = -(xr * xr) / 2.0 - k->norm_pdf_logC;
        w = w - mass_r - lsd[i];
        rev[i] = ((ar <= xr) && (xr <= br)) ? w : -INFINITY;
    }
    const double pFN1 = 1.0 - a->FN, pFP0 = 1.0 - a->FP;
    for (int pass = 0; pass < 2; pass++) {
        const float *th = pass == 0 ? nw : old;
        for (int i = 0; i < n; i++) {
            const double t64 = (double)th[i];
            const double om64 = (double)(1.0f - th[i]);
            t0[i] = t64 * pFN1 + om64 * a->FP;
            t1[i] = t64 * a->FN + om64 * pFP0;
        }
        uloop(k->np_log, k->np_log_data, t0, t2, n);
        uloop(k->np_log, k->np_log_data, t1, t3, n);
        double *dst = pass == 0 ? ll_new : ll_old;
        for (int i = 0; i < n; i++)
            dst[i] = (double)n1[i] * t2[i] + (double)n0[i] * t3[i];
    }
    double *A = a->A + off;
    for (int i = 0; i < n; i++) {
        double pn = 0.0, po = 0.0;
        if (!a->uniform_prior) {
            pn = beta_logpdf1(k, nw[i], a->p, a->q, betaln_pq);
            po = beta_logpdf1(k, old[i], a->p, a->q, betaln_pq);
        }
        double v = ll_new[i] + pn;
        v = v - ll_old[i];
        v = v - po;
        v = v + rev[i];
        v = v - fwd[i];
        if (a->clip && v > 0.0) v = 0.0;
        A[i] = v;
    }
    return true;
}

}  // namespace

extern "C" int bnpc_log_accept(const bnpc_host_kernels *k, const bnpc_accept_args *a,
                          int *status)
{
    if (check_kernels(k)) return 2;
    if (!a || !status || a->G < 0 || a->M < 1 || !a->new_theta
        || !a->old_theta || !a->std || !a->n1 || !a->n0 || !a->A || !a->sum) {
        bnpc_set_error("bad argument: log_A");
        return 2;
    }
    *status = 0;
    const double bl = a->uniform_prior ? 0.0 : k->betaln(a->p, a->q, 0);
    const int64_t chunks = (a->M + BLK - 1) / BLK, tasks = a->G * chunks;
    int threads = a->threads;
    // a rank per 2 tasks, the team from 1024 entries on
    const int64_t accept_per = 2, accept_min = 1024;
    if (threads > (tasks + accept_per - 1) / accept_per)
        threads = (int)((tasks + accept_per - 1) / accept_per);
    if (tasks * BLK < accept_min || threads < 1) threads = 1;
    std::atomic<int64_t> next(0);
    std::atomic<int> bail(0);
    auto work = [&](int) {
        for (;;) {
            const int64_t t = next.fetch_add(1, std::memory_order_relaxed);
            if (t >= tasks) break;
            if (bail.load(std::memory_order_relaxed)) continue;
            const int64_t g = t / chunks, m0 = (t - g * chunks) * BLK;
            const int64_t m1 = m0 + BLK < a->M ? m0 + BLK : a->M;
            if (!logA_block(k, a, bl, g, m0, m1))
                bail.store(1, std::memory_order_relaxed);
        }
    };
    if (threads > 1)
        team_for(threads)->run(threads, work);
    else
        work(0);
    if (bail.load()) {
        *status = 1;
        return 0;
    }
    for (int64_t g = 0; g < a->G; g++) {
        const double *A = a->A + g * a->M;
        double s = A[0];
        for (int64_t m = 1; m < a->M; m++) s += A[m];       // np.cumsum order
        a->sum[g] = s;
    }
    return 0;
}

// truncnorm.ppf(q, a, b, loc, scale) for scalars = truncnorm.rvs given its one
// uniform (the error-rate proposals, libs/CRP_learning_errors.py:81-84), as
// scipy's truncnorm_gen._ppf evaluates it (the per-element arithmetic of the
// batch above).  *status = 1: left to SciPy.
extern "C" int bnpc_tn_ppf_scalar(const bnpc_host_kernels *k, double q,
                                  double a, double b, double loc, double scale,
                                  double *out, int *status)
{
    if (check_kernels(k)) return 2;
    if (!out || !status) {
        bnpc_set_error("bad argument: tn_ppf_scalar");
        return 2;
    }
    *status = 1;
    double mass;
    if (!(q > 0.0 && q < 1.0) || !gauss_mass(k, a, b, &mass)) return 0;
    const bool left = a < 0.0;
    double lq, arg = left ? q : -q;
    if (left) uloop(k->np_log, k->np_log_data, &arg, &lq, 1);       // log(q)
    else uloop(k->np_log1p, k->np_log1p_data, &arg, &lq, 1);        // log1p(-q)
    const double pp = k->log_ndtr(left ? a : -b, 0);
    const double qq = lq + mass;
    if (!isfinite(pp) || !isfinite(qq)) return 0;
    const double top = pp > qq ? pp : qq, bot = pp > qq ? qq : pp;
    const bool tie = pp == qq;
    double d = bot - top, e, l1p, lm, m = tie ? 2.0 : 1.0;
    uloop(k->np_exp, k->np_exp_data, &d, &e, 1);
    if (tie) e = 0.0;
    uloop(k->np_log1p, k->np_log1p_data, &e, &l1p, 1);
    uloop(k->np_log, k->np_log_data, &m, &lm, 1);
    double x = k->ndtri_exp(l1p + lm + top, 0);
    if (!left) x = -x;
    *out = x * scale + loc;
    *status = 0;
    return 0;
}
