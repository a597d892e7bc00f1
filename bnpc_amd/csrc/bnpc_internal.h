// internal helpers shared by the translation units of libbnpc_hip.so
#ifndef BNPC_INTERNAL_H
#define BNPC_INTERNAL_H
#include <math.h>
#include <stdarg.h>
#include <stdint.h>

#include "bnpc_hip.h"

void bnpc_set_error(const char *fmt, ...);

#ifdef __cplusplus
#include <functional>
// fn(rank) for every rank 0..n-1, n = bnpc_team_ranks(threads), on this
// process's host thread team (the caller is rank 0; the team grows to n if it
// is smaller - bnpc_team_ranks does that and returns fewer when the system
// refuses threads); returns n when all ranks are done.  One job at a time:
// concurrent callers take turns (the team's job lock).
int bnpc_team_ranks(int threads);
int bnpc_team_run(int threads, const std::function<void(int)> &fn);
// fn() on the process's one aside thread, NEXT TO the caller (it returns at
// once; false: no thread to be had, nothing runs); bnpc_aside_wait returns
// when that job is over.  One job at a time, posted and awaited by the thread
// that drives the chain.
bool bnpc_aside_start(const std::function<void()> &fn);
void bnpc_aside_wait();
#endif

// bnpc_kernels.hip: the {ones, zeros} 64-bit words of one cell's row
// (W pairs), or NULL
const unsigned long long *bnpc_ctx_row(const bnpc_ctx *c, int64_t cell,
                                       int64_t *M, int *W);
// bnpc_mt.cpp: a profile row of legacy Beta draws from one cell's bit words
void bnpc_legacy_beta_row(bnpc_mt19937 *rng, bnpc_legacy_gauss *g, int64_t M,
                          const unsigned long long *row, double p, double q,
                          double tmin, double tmax, float *theta);

// bnpc_kernels.hip: bumped by every bnpc_colcounts_by_label on the context -
// whoever remembers "the resident per-cluster counts are those of state X"
// (bnpc_chain_step) remembers this number with it
uint64_t bnpc_ctx_label_counts_generation(const bnpc_ctx *c);
// bnpc_sweeps.cpp: a cluster for `cell` opened by the caller of
// bnpc_gibbs_sweep when the sweep returned the cell through st->new_cell (no
// spare column was left; the caller has made room): exactly what the sweep
// does itself while it has room (libs/CRP.py:281-282, 291-299, 183-188)
int bnpc_sweep_open_cluster(bnpc_gibbs_state *st, bnpc_mt19937 *rng,
                            int64_t cell, double *ll, int64_t *assignment,
                            int64_t *col_of_id, int64_t *col_id,
                            int64_t *col_size, int64_t *order);
// bnpc_mt.cpp: np.random.gamma(shape, scale) of the legacy stream
double bnpc_legacy_gamma(bnpc_mt19937 *rng, bnpc_legacy_gauss *g, double shape,
                         double scale);

// bnpc_kernels.hip: counts of the two launch clusters of a restricted scan +
// the (screened) parameter batch on one stream synchronisation
// The next screened bnpc_mh_batch of this thread has its rank 0 call *hook
// before it joins the evaluation of the batch (bnpc_mh_batch_dev: the draws
// and the screen launch of a later part of a large batch under the team's
// work on this one); NULL: nothing.  Consumed by that call.
void bnpc_mh_rank0_hook(const std::function<void()> *hook);
// ... and every rank of it calls (*gate)(row) before it touches a row of the
// batch: the call returns when that row's verdicts are there (a batch that is
// evaluated as ONE team job while its later parts are still being screened);
// false gives the batch up.  NULL: nothing.  Consumed by that call.
void bnpc_mh_row_gate(const std::function<bool(int64_t)> *gate);
#ifdef __cplusplus
// bnpc_kernels.hip: the draws of the NEXT screened parameter batch on this
// context taken ahead (MhAhead there): a walker on the aside thread takes a
// copy of (rng, g), runs `prelude` on it - the draws the stream goes through
// before the batch begins; false: it cannot tell - and then draws `rows` rows
// of M into the batch's pinned block.  The batch adopts them iff the live
// stream stands exactly where the walker's stood after the prelude.  *posted:
// whether a walker was started (not for small batches, without the screen,
// BNPC_MH_AHEAD=0 ...).  bnpc_mh_ahead_drop ends one that will not be used.
int bnpc_mh_ahead_begin(bnpc_ctx *c, const bnpc_mt19937 *rng,
                        const bnpc_legacy_gauss *g,
                        const std::function<bool(bnpc_mt19937 *,
                                                 bnpc_legacy_gauss *)> &prelude,
                        int64_t rows, int64_t M, int64_t n_sd, bool *posted,
                        int64_t min_entries = -1);
void bnpc_mh_ahead_drop(bnpc_ctx *c);
// ... and the rows of a restricted scan's batch: `uniforms` uniforms lie
// between the stream as it is and the batch's first draw
void bnpc_mh_ahead_scan(bnpc_ctx *c, const bnpc_mt19937 *rng, int64_t uniforms,
                        int64_t rows, int64_t M, int64_t n_sd);
// bnpc_sweeps.cpp: called by the NEXT bnpc_rg_scan (mode 0) of this thread
// right after it has drawn its visiting order; consumed by that call.
void bnpc_rg_scan_order_hook(const std::function<void()> *hook);
// bnpc_moves.cpp: called by the NEXT bnpc_sm_move of this thread right after
// the move's last draw of variable length, with the number of uniforms that
// still follow inside the move (its acceptance test: 0 or 1); consumed by
// that call.  NULL: nothing.
void bnpc_move_last_draw_hook(const std::function<void(int)> *hook);
#endif
// bnpc_ll_theta in two halves (bnpc_kernels.hip): the caller works between
// them - bnpc_sm_move draws the third Beta row under the first scan's sums
int bnpc_ll_theta_begin(bnpc_ctx *c, int view, const float *theta, int64_t K,
                        double FP, double FN, double *out, int64_t ldo);
int bnpc_ll_theta_end(bnpc_ctx *c);
// bnpc_rg_scan_step whose n x 2 log-likelihoods are already there (ll_ready,
// from bnpc_ll_theta_begin / _end); NULL: evaluated inside
int bnpc_rg_scan_step_with(bnpc_ctx *ctx, const bnpc_host_kernels *k,
                           bnpc_mt19937 *rng, int view, int64_t n,
                           int64_t *rg_assignment, double DP_a,
                           const bnpc_mh_args *mh, int32_t *n1, int32_t *n0,
                           double *scan_log_prob, int *status,
                           const double *ll_ready);
int bnpc_rg_counts_and_batch(bnpc_ctx *c, const bnpc_host_kernels *k,
                             bnpc_mt19937 *rng, int view,
                             const int64_t *labels, const bnpc_mh_args *a,
                             int32_t *n1, int32_t *n0, int *status);

// ---------------------------------------------------------------------------
// MT19937 (Matsumoto & Nishimura), state layout of np.random.get_state()
// ---------------------------------------------------------------------------
// bnpc_mt.cpp: the state refill (624 words), vectorised; picked at load time
#ifdef __cplusplus
void mt_refill_block(uint32_t *key);
#endif
static inline void mt_refill(bnpc_mt19937 *s)
{
    mt_refill_block(s->key);
    s->pos = 0;
}

static inline uint32_t mt_next32(bnpc_mt19937 *s)
{
    if (s->pos >= 624) mt_refill(s);
    uint32_t y = s->key[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static inline double mt_double(bnpc_mt19937 *s)
{
    const int32_t a = (int32_t)(mt_next32(s) >> 5);
    const int32_t b = (int32_t)(mt_next32(s) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

// legacy random_interval: uniform integer in [0, max], masked rejection
static inline uint64_t mt_interval(bnpc_mt19937 *s, uint64_t max)
{
    if (max == 0) return 0;
    uint64_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    mask |= mask >> 32;
    uint64_t v;
    if (max <= 0xffffffffull) {
        while ((v = (mt_next32(s) & mask)) > max) {}
    } else {
        for (;;) {
            const uint64_t hi = mt_next32(s), lo = mt_next32(s);
            v = ((hi << 32) | lo) & mask;
            if (v <= max) break;
        }
    }
    return v;
}

// ---- bulk draws (bnpc_mt.cpp): the same stream as mt_double / mt_interval
// one by one, tempered straight out of the state block in vectorised loops
// (AVX-512 / AVX2 / baseline clones, picked at load time) -------------------
void mt_fill_double(bnpc_mt19937 *s, double *out, int64_t n,
                    bool stream = false);
// bnpc_mt_mh_draws; stream: the uniforms go out with non-temporal stores (a
// large pinned destination the device reads next), fenced before return
int bnpc_mt_mh_draws_to(bnpc_mt19937 *rng, int64_t G, int64_t M, int64_t n_sd,
                        int32_t *sd_idx, double *U, double *u, bool stream);
void mt_fill_interval32(bnpc_mt19937 *s, uint32_t max, int32_t *out,
                        int64_t n);
void mt_fill_permutation(bnpc_mt19937 *s, int64_t n, int64_t *out);

// One element of bnpc_log_diff_pi (include/bnpc_hip.h): complex exp of
// (q - p, pi) = exp(q - p) * (cos pi, sin pi); the maximal term is split off
// and contributes 0 to the sum; complex log1p = log(hypot(re + 1, im));
// + log(1) + p.
static inline double bnpc_log_diff_pi1(double log_p, double log_q)
{
    double sin_pi, cos_pi;
    sincos(3.141592653589793, &sin_pi, &cos_pi);
    const double E = exp(log_q - log_p);
    const double re = E * cos_pi + 1.0;
    const double im = E * sin_pi;
    return (log(hypot(re, im)) + 0.0) + log_p;
}


#ifdef __cplusplus
#include <algorithm>
#include <vector>
// ---------------------------------------------------------------------------
// NumPy expressions restated for the native moves / steps (bnpc_moves.cpp,
// bnpc_step.cpp): NumPy's own float64 log loop, np.sum's pairwise order, the
// legacy np.random.choice(p=...)
// ---------------------------------------------------------------------------
static inline void np_loop(bnpc_uloop f, void *data, const double *in,
                           double *out, intptr_t n)
{
    if (n <= 0) return;
    char *args[2] = {(char *)in, (char *)out};
    intptr_t dims[1] = {n};
    intptr_t steps[2] = {(intptr_t)sizeof(double), (intptr_t)sizeof(double)};
    f(args, dims, steps, data);
}

static inline double np_log1(const bnpc_host_kernels *k, double x)
{
    double out;
    np_loop(k->np_log, k->np_log_data, &x, &out, 1);
    return out;
}

// NumPy's pairwise summation of a contiguous float64 run
// (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum): plain loop
// below 8 elements, 8 interleaved partial sums up to 128, halves (the first
// a multiple of 8) above.
static inline double np_pairwise(const double *a, int64_t n)
{
    if (n < 8) {
        double res = -0.0;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3]))
                     + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}

// np.sum(a) of a contiguous float64 vector: the reduction starts from the
// additive identity and adds the pairwise sum of every run of 8192 elements
// (the iterator's buffer size) in turn
static inline double np_sum(const double *a, int64_t n)
{
    double out = 0.0;
    for (int64_t at = 0; at < n; at += 8192)
        out = out + np_pairwise(a + at, n - at < 8192 ? n - at : 8192);
    return out;
}

// np.random.choice(K, p=p) of the legacy RandomState given its uniform u:
// cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(u, side='right')
static inline int64_t np_choice_p(const double *p, int64_t K,
                                  std::vector<double> &cdf, double u)
{
    cdf.resize((size_t)K);
    double s = 0.0;
    for (int64_t k = 0; k < K; k++) {
        s = k ? s + p[k] : p[0];
        cdf[k] = s;
    }
    const double last = cdf[K - 1];
    for (int64_t k = 0; k < K; k++) cdf[k] /= last;
    return std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin();
}
#endif

#endif
