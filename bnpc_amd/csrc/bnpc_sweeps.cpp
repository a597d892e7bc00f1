// bnpc_sweeps.cpp - the sequential halves of the sampler moves, native.
//
// After the cells x clusters x mutations op runs on the GPU, what remains of a
// Gibbs sweep is a loop over cells that is sequential by construction (cluster
// sizes and newly opened clusters feed forward): CRP.update_assignments_Gibbs,
// /root/reference/libs/CRP.py:260-288.  In Python that loop costs 100-200 us
// per cell ("Amdahl on the host"); here it is O(K) flops
// per cell.  To keep the assignment TRAJECTORY identical, the loop draws from
// an exact replica of NumPy's legacy global stream (MT19937 + the legacy
// random_sample / random_interval / choice(p=) algorithms), whose state is
// exchanged with np.random.get_state()/set_state().
//
// Caveat (exactness is "same stack", not a theorem): the normalisations below
// call the C library's exp / log1p, the reference NumPy's (SIMD) exp; the two
// agree to the last bit almost everywhere, and a 1-ulp difference only matters
// if a uniform draw falls within that ulp of a cumulative probability (never
// observed: profiles/r02/soak_*.log).  Non-finite posteriors, for which the
// reference has FloatingPointError branches (CRP.py:94-98, 110-114), are
// reported as errors (code 4), not guessed at.
//
// NumPy is a third-party dependency of the reference (unpinned, implied by
// requirements.txt:1-5); the algorithms restated here are those of
// numpy/random/_legacy (mt19937 genrand, legacy_double, random_interval,
// RandomState.choice), pinned by golden vectors captured from NumPy itself
// (tests/golden/rng.npz, tests/test_native_sweeps.py).

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <vector>

#include "bnpc_hip.h"
#include <functional>
#include "bnpc_internal.h"

extern "C" double bnpc_mt_random_sample(bnpc_mt19937 *rng)
{
    return mt_double(rng);
}

extern "C" int bnpc_mt_permutation(bnpc_mt19937 *rng, int64_t n, int64_t *out)
{
    if (!rng || (n > 0 && !out)) {
        bnpc_set_error("bad argument: NULL");
        return 2;
    }
    if (n > 0 && n <= 0xffffffffll) {
        mt_fill_permutation(rng, n, out);   // pre-tempered words (bnpc_mt.cpp)
        return 0;
    }
    for (int64_t i = 0; i < n; i++) out[i] = i;
    for (int64_t i = n - 1; i >= 1; i--) {
        const int64_t j = (int64_t)mt_interval(rng, (uint64_t)i);
        const int64_t t = out[i];
        out[i] = out[j];
        out[j] = t;
    }
    return 0;
}

// The three draws of MH_cluster_params (libs/CRP.py:328-335) for G clusters,
// in the reference's per-cluster order:
//   np.random.choice(sd, size=M)   = legacy randint(0, n_sd, M): masked
//                                    rejection on 32-bit draws
//   truncnorm.rvs(size=M)          : its M uniforms (random_state.uniform)
//   np.random.random(M)
extern "C" int bnpc_mt_mh_draws(bnpc_mt19937 *rng, int64_t G, int64_t M,
                                int64_t n_sd, int32_t *sd_idx, double *U,
                                double *u)
{
    return bnpc_mt_mh_draws_to(rng, G, M, n_sd, sd_idx, U, u, false);
}

int bnpc_mt_mh_draws_to(bnpc_mt19937 *rng, int64_t G, int64_t M, int64_t n_sd,
                        int32_t *sd_idx, double *U, double *u, bool stream)
{
    if (!rng || !sd_idx || !U || !u || n_sd < 1) {
        bnpc_set_error("bad argument: NULL");
        return 2;
    }
    for (int64_t g = 0; g < G; g++) {
        int32_t *si = sd_idx + g * M;
        double *Ug = U + g * M, *ug = u + g * M;
        mt_fill_interval32(rng, (uint32_t)(n_sd - 1), si, M);
        // uniform(0, 1): 0.0 + 1.0 * u == u
        mt_fill_double(rng, Ug, M, stream);
        mt_fill_double(rng, ug, M, stream);
    }
#if defined(__x86_64__)
    if (stream) __builtin_ia32_sfence();
#endif
    return 0;
}

// scipy.special.logsumexp([p, q + pi*1j], axis=0).real for q <= p, term by
// term as NumPy evaluates it: complex exp of (q - p, pi) = exp(q - p) *
// (cos pi, sin pi); the maximal term is split off and contributes 0 to the
// sum; complex log1p = log(hypot(re + 1, im)); + log(1) + p.
extern "C" int bnpc_log_diff_pi(const double *log_p, const double *log_q,
                                int64_t n, double *out)
{
    if ((!log_p || !log_q || !out) && n > 0) {
        bnpc_set_error("bad argument: NULL");
        return 2;
    }
    for (int64_t i = 0; i < n; i++) out[i] = bnpc_log_diff_pi1(log_p[i], log_q[i]);
    return 0;
}

// BNPC_LOOP_SHORTCUTS=0 makes the two loops below evaluate every exp() /
// log1p() the plain way (no "60 below the runner-up" cut in the sweep, no
// "pair 40 apart" short cut in the restricted scan): the old-vs-new
// comparison of tests/test_native_sweeps.py.  Read per call, not per cell.
static bool loop_shortcuts()
{
    const char *e = getenv("BNPC_LOOP_SHORTCUTS");
    return !(e && e[0] == '0');
}

// BNPC_SWEEP_LANE (tests; read per call): "0" = no lane for the cells decided
// from their records (sweep_window's general iteration for every cell: the
// old-vs-new comparison); a number x in (0, 0.5) = the lane hands every cell
// whose uniform lies within x of 0 or 1 over to the general iteration WITH
// that uniform - what it does for the 1e-11 slivers at either end, made
// frequent enough to test.  Returns the distance, < 0 for "no lane".
// BNPC_SWEEP_LANE=nostride: the lane without its stride (every cell by the
// lane's own steps: the A/B and the old-vs-new comparison of the stride)
static bool sweep_lane_stride()
{
    const char *e = getenv("BNPC_SWEEP_LANE");
    return !(e && strcmp(e, "nostride") == 0);
}

static double sweep_lane_sliver()
{
    // (1e-11: with at most 4096 live clusters the floor entries below the
    // winner end under 4.1e-12 and the ones above it start beyond 1 - 4.1e-12
    // - the rule sweep_window's dominated pick has for 512 and 1e-12)
    const char *e = getenv("BNPC_SWEEP_LANE");
    if (!e || !e[0] || strcmp(e, "nostride") == 0) return 1e-11;
    const double x = strtod(e, nullptr);
    if (!(x > 0.0)) return -1.0;
    return x < 0.5 ? (x > 1e-11 ? x : 1e-11) : 1e-11;
}

// ---------------------------------------------------------------------------
// Gibbs sweep
// ---------------------------------------------------------------------------
static const double LOG_EPS = -34.538776394910684;   // np.log(1e-15)
static const double EXP_LOG_EPS = exp(LOG_EPS);       // the probability floor

// floor_sums(k)[j] = the floor added to 0.0 j times, one by one (what a
// sequential cumsum over j clipped probabilities holds), for j <= k.
static const double *floor_sums(int64_t k)
{
    static thread_local std::vector<double> sums(1, 0.0);
    while ((int64_t)sums.size() <= k + 1)
        sums.push_back(sums.back() + EXP_LOG_EPS);
    return sums.data();
}

// For x in [1, 2): fl(x + floor) = x + FLOOR_STEP * 2^-52, because the floor
// is 4.5036 ulps of such x - never a tie.  Checked once against the adder.
static int64_t floor_step_ulps()
{
    const double x = 1.0 + 12345 * 0x1p-52;
    const double y = x + EXP_LOG_EPS;
    const int64_t step = (int64_t)((y - x) * 0x1p52);
    const double frac = EXP_LOG_EPS * 0x1p52 - (double)(step - 1);
    // the rounding must not be a near-tie and must be the same at both ends
    if (!(frac > 0.501 && frac < 0.999)) return -1;
    const double z = (2.0 - 64 * 0x1p-52) + EXP_LOG_EPS;
    if ((int64_t)((z - (2.0 - 64 * 0x1p-52)) * 0x1p52) != step) return -1;
    return step;
}
static const int64_t FLOOR_STEP = floor_step_ulps();

// The running sums the dominated fast path of bnpc_gibbs_sweep uses instead
// of walking np.cumsum: cdf[a], a = 0..A, of the probability vector that is
// 1.0 at `top` and the floor elsewhere (exported for the tests).
extern "C" int bnpc_dominated_cdf(int64_t A, int64_t top, double *cdf)
{
    if (!cdf || A < 0 || top < 0 || top > A) {
        bnpc_set_error("bad argument: dominated_cdf");
        return 2;
    }
    if (FLOOR_STEP <= 0) {
        bnpc_set_error("closed form not available on this host");
        return 1;
    }
    const double *fs = floor_sums(top);
    const double at_top = fs[top] + 1.0;
    for (int64_t a = 0; a <= A; a++)
        cdf[a] = a < top ? fs[a + 1]
            : at_top + (double)(FLOOR_STEP * (a - top)) * 0x1p-52;
    return 0;
}

// -39 - log(A + 1): below this gap to the runner-up the tail sum of A
// exponentials cannot reach 2^-55 (see the dominated case below)
static const double *dominated_bounds(int64_t A)
{
    static thread_local std::vector<double> bound;
    while ((int64_t)bound.size() <= A)
        bound.push_back(-39.0 - log((double)(bound.size() + 1)));
    return bound.data();
}

// ---- the per-cell scan over many live clusters, on a team -------------------
// In the first sweep of a large data set a cell is scored against tens of
// thousands of clusters.  Rank 0 walks the cells (draws, state updates: the
// sequential part, unchanged); for a cell with many live clusters it posts the
// scan to the other ranks, which spin on a phase word for the duration of the
// window: every rank scores a contiguous range of the live list and reports
// (maximum, first position of it, runner-up); the ranges are combined in order,
// so `top` is the first maximum and `second` the largest other entry exactly as
// in the one-thread scan.  Exponentials of the non-dominated case are taken in
// parallel, their SUMS stay sequential in index order (same bits).
namespace {

inline void spin_pause()
{
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

struct ParScan {
    enum { SCAN = 1, EXP_TAIL = 2, EXP_PROB = 3, EXIT = 4, MAXT = 64 };
    int T = 1;
    alignas(64) std::atomic<uint32_t> phase{0};
    alignas(64) std::atomic<int> done{0};
    int mode = 0;
    const double *row = nullptr, *cpr = nullptr;
    const int64_t *order = nullptr;
    double *post = nullptr, *buf = nullptr;
    int64_t A = 0, top = 0;
    double ptop = 0.0, lnorm = 0.0;
    struct alignas(64) Part {
        double best, second;
        int64_t top;
    } part[MAXT];

    void work(int rank)
    {
        const int64_t lo = A * rank / T, hi = A * (rank + 1) / T;
        if (mode == SCAN) {
            double best = -INFINITY, second = -INFINITY;
            int64_t t = lo;
            for (int64_t a = lo; a < hi; a++) {
                const int64_t c = order[a];
                const double v = row[c] + cpr[c];
                post[a] = v;
                if (v > best) {
                    second = best;
                    best = v;
                    t = a;
                } else if (v > second) {
                    second = v;
                }
            }
            part[rank].best = best;
            part[rank].second = second;
            part[rank].top = t;
        } else if (mode == EXP_TAIL) {
            // entries 0..A of post (the new-cluster entry included: ranges
            // over A + 1 here)
            const int64_t n = A + 1;
            const int64_t l2 = n * rank / T, h2 = n * (rank + 1) / T;
            for (int64_t a = l2; a < h2; a++) {
                const double d = post[a] - ptop;
                buf[a] = (a != top && d > -746.0) ? exp(d) : 0.0;
            }
        } else if (mode == EXP_PROB) {
            const int64_t n = A + 1;
            const int64_t l2 = n * rank / T, h2 = n * (rank + 1) / T;
            for (int64_t a = l2; a < h2; a++) {
                const double v = post[a] - ptop - lnorm;
                buf[a] = (v <= LOG_EPS) ? EXP_LOG_EPS : exp(v > 0.0 ? 0.0 : v);
            }
        }
    }
    // rank 0: run one phase on all ranks
    void run(int m)
    {
        mode = m;
        done.store(0, std::memory_order_relaxed);
        phase.fetch_add(1, std::memory_order_release);
        work(0);
        while (done.load(std::memory_order_acquire) != T - 1) spin_pause();
    }
    void finish()
    {
        mode = EXIT;
        phase.fetch_add(1, std::memory_order_release);
    }
    // ranks 1..T-1
    void serve(int rank)
    {
        uint32_t seen = 0;
        for (;;) {
            uint32_t p;
            while ((p = phase.load(std::memory_order_acquire)) == seen)
                spin_pause();
            seen = p;
            if (mode == EXIT) return;
            work(rank);
            done.fetch_add(1, std::memory_order_release);
        }
    }
};

}  // namespace

// CRP.init_new_cluster for one cell (libs/CRP.py:291-299 with :183-188): the
// lowest free id; its profile: per mutation Beta(p + [x = 1], q + [x = 0])
// from the cell's own observations (NaN adds nothing), NumPy's legacy sampler
// on the stream, clipped, float32; the new column of ll from the device.
static int open_cluster(bnpc_gibbs_state *st, bnpc_mt19937 *rng, int64_t cell,
                        double *ll, int64_t *assignment, int64_t *col_of_id,
                        int64_t *col_id, int64_t *col_size, int64_t *order,
                        int64_t *free_hint)
{
    int64_t M = 0;
    int W = 0;
    const unsigned long long *row = bnpc_ctx_row(st->birth_ctx, cell, &M, &W);
    if (!row || !st->theta_host || !st->gauss || !st->born) {
        bnpc_set_error("bad argument: native birth is not set up");
        return 2;
    }
    int64_t id = *free_hint;
    while (id < st->n_cells && col_of_id[id] >= 0) id++;
    if (id >= st->n_cells) {
        bnpc_set_error("no free cluster id");
        return 3;
    }
    *free_hint = id + 1;
    float *theta = st->theta_host + (size_t)id * M;
    bnpc_legacy_gauss *g = (bnpc_legacy_gauss *)st->gauss;
    bnpc_legacy_beta_row(rng, g, M, row, st->beta_p, st->beta_q, st->tmin,
                         st->tmax, theta);
    if (st->birth_put) {
        const int rc = bnpc_theta_put(st->birth_ctx, id, theta, 1);
        if (rc) return rc;
    }
    static thread_local std::vector<double> column;
    column.resize((size_t)st->birth_rows);
    const int rc = bnpc_ll_theta(st->birth_ctx, st->birth_view, theta, 1,
                                 st->FP, st->FN, column.data(), 1);
    if (rc) return rc;
    const int64_t col = st->n_cols, ld = st->ld;
    for (int64_t r = 0; r < st->birth_rows; r++)
        ll[(size_t)r * ld + col] = column[(size_t)r];
    col_id[col] = id;
    col_size[col] = 1;
    col_of_id[id] = col;
    order[st->n_active] = col;
    st->n_active++;
    st->n_cols++;
    assignment[cell] = id;
    st->born[st->n_born++] = id;
    return 0;
}

int bnpc_sweep_open_cluster(bnpc_gibbs_state *st, bnpc_mt19937 *rng,
                            int64_t cell, double *ll, int64_t *assignment,
                            int64_t *col_of_id, int64_t *col_id,
                            int64_t *col_size, int64_t *order)
{
    if (!st || !rng || !ll || !assignment || !col_of_id || !col_id
        || !col_size || !order || !st->birth_ctx || st->n_cols >= st->ld
        || st->n_born >= st->born_cap) {
        bnpc_set_error("bad argument: sweep_open_cluster");
        return 2;
    }
    int64_t free_hint = 0;
    return open_cluster(st, rng, cell, ll, assignment, col_of_id, col_id,
                        col_size, order, &free_hint);
}

// A cell torn between two live entries (`top`, the first maximum, and `sec`,
// d2 = runner-up - top <= 0 apart), every other of the A + 1 entries on the
// 1e-15 floor: which entry does the uniform u pick?  Only WHICH interval of
// the cumulative sums u falls into matters, not their bits.  The two live
// entries have probabilities 1 / (1 + x) and x / (1 + x), x = exp(d2): the
// interval ends are known to ~1e-14 (A floors of 1e-15, a few roundings of
// 1e-16, the exponent's own rounding 37 * 1e-16) from ONE exp() instead of four
// calls, A + 1 adds and a bisection with divisions.  A uniform further than
// 1e-11 from both ends of one of the two wide intervals picks what the full
// arithmetic picks; -1 = nearer than that (or inside a floor sliver): the
// caller evaluates the cell in full.
//
// pair_pick_weights: the same from the two entries' WEIGHTS (any common
// scale; known to a relative `band` / 4 or better): w = 1 and exp(d2) here;
// the sweep loop passes size1 and e2 * size2 - the likelihood ratio the hint
// kernel has already exponentiated (bnpc_top2.e2, a float32: band 1e-6) times
// the ratio of the CURRENT sizes, which IS the ratio of the priors
// (log(size) - common term, libs/CRP.py:83-85): no exp() in the loop.
static inline int64_t pair_pick_weights(double w_a, double w_b, int64_t a,
                                        int64_t b, int64_t A, double u,
                                        double band)
{
    const double inv = 1.0 / (w_a + w_b);
    const int64_t m1 = a < b ? a : b;
    const int64_t m2 = a < b ? b : a;
    const double e1 = (a < b ? w_a : w_b) * inv;
    const double e2 = (a < b ? w_b : w_a) * inv;
    const double un = u * (1.0 + (double)(A - 1) * EXP_LOG_EPS);
    const double lo1 = (double)m1 * EXP_LOG_EPS, hi1 = lo1 + e1;
    const double lo2 = hi1 + (double)(m2 - m1 - 1) * EXP_LOG_EPS;
    const double hi2 = lo2 + e2;
    if (un > lo1 + band && un < hi1 - band) return m1;
    if (un > lo2 + band && un < hi2 - band) return m2;
    return -1;
}

static inline int64_t pair_pick_quick(double d2, int64_t A, int64_t top,
                                      int64_t sec, double u)
{
    const double x = d2 > -746.0 ? exp(d2) : 0.0;
    return pair_pick_weights(1.0, x, top, sec, A, u, 1e-11);
}

// The same pick by the scan's own arithmetic (_normalize_log_probs,
// libs/CRP.py:88-100, with every entry but the two at the floor; cdf: A + 1
// doubles of scratch)
static inline int64_t pair_pick_full(double d2, int64_t A, int64_t top,
                                     int64_t sec, double u, double *cdf)
{
    double tail = 0.0, run = 0.0;
    if (d2 > -746.0) tail += exp(d2);
    const double lnorm = log1p(tail);
    const double v_top = 0.0 - lnorm;
    const double v_sec = d2 - lnorm;
    const double e_top = (v_top <= LOG_EPS) ? EXP_LOG_EPS
        : exp(v_top > 0.0 ? 0.0 : v_top);
    const double e_sec = (v_sec <= LOG_EPS) ? EXP_LOG_EPS
        : exp(v_sec > 0.0 ? 0.0 : v_sec);
    for (int64_t a = 0; a <= A; a++) {
        run += a == top ? e_top : (a == sec ? e_sec : EXP_LOG_EPS);
        cdf[a] = run;
    }
    const double total = cdf[A];
    int64_t lo = 0, hi = A + 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cdf[mid] / total > u) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

// One cell of a restricted 2-way scan that needs no log-probabilities:
// P(0) = 1 / (1 + x) or x / (1 + x), x = exp(-|p0 - p1|) (d <= 0, `big` = the
// larger entry), known to ~1e-15 from one exp(); a uniform further than 1e-11
// from it picks what two_way_pick_full picks, -1 = too close to tell.
static inline int two_way_pick_quick(double d, int big, double u)
{
    const double x = exp(d);
    const double q0 = big ? x / (1.0 + x) : 1.0 / (1.0 + x);
    if (fabs(u - q0) > 1e-11) return u < q0 ? 0 : 1;
    return -1;
}

// _normalize_log for two entries + np.random.choice([0, 1], p=exp(.)) given
// its uniform (libs/CRP.py:103-116, 625-628); l0 / l1 = the log-probabilities
static inline int two_way_pick_full(double d, int big, double u, double *l0,
                                    double *l1)
{
    const double z = log1p(exp(d));
    *l0 = big ? d - z : 0.0 - z;
    *l1 = big ? 0.0 - z : d - z;
    const double e0 = exp(*l0), e1 = exp(*l1);
    const double c0 = e0, c1 = e0 + e1;
    int pick = (c0 / c1 > u) ? 0 : 1;
    if (!(c1 / c1 > u)) pick = 1;
    return pick;
}

// The same for a cell with THREE live entries (the pieces of a cluster that
// was split twice, 9 % of the cells of a running config-3 chain): values q[i]
// at list positions a[i], every other of the A + 1 entries more than 61 below
// the runner-up (floor probability, no term in the tail sum).  Probabilities
// x_i / (1 + x_b + x_c) from two exp(); an entry the scan would clip to the
// floor differs from that by less than 1e-15.  -1 = u is within 1e-11 of an
// interval end: the caller scans the cell (with the same uniform).
//
// triple_pick_weights: from the three entries' weights (any common scale,
// relative error below band / 4), as pair_pick_weights.
static inline int64_t triple_pick_weights(const double x[3],
                                          const int64_t a[3], int64_t A,
                                          double u, double band)
{
    const double Z = x[0] + x[1] + x[2];
    int o[3] = {0, 1, 2};                       // by list position
    if (a[o[0]] > a[o[1]]) { const int w = o[0]; o[0] = o[1]; o[1] = w; }
    if (a[o[1]] > a[o[2]]) { const int w = o[1]; o[1] = o[2]; o[2] = w; }
    if (a[o[0]] > a[o[1]]) { const int w = o[0]; o[0] = o[1]; o[1] = w; }
    const double un = u * (1.0 + (double)(A - 2) * EXP_LOG_EPS);
    double edge = 0.0;
    int64_t prev = -1;
    for (int j = 0; j < 3; j++) {
        const int i = o[j];
        const double lo = edge + (double)(a[i] - prev - 1) * EXP_LOG_EPS;
        const double hi = lo + x[i] / Z;
        if (un > lo + band && un < hi - band) return a[i];
        edge = hi;
        prev = a[i];
    }
    return -1;
}

static inline int64_t triple_pick_quick(const double q[3], const int64_t a[3],
                                        int64_t A, double u)
{
    int t = 0;
    for (int i = 1; i < 3; i++)
        if (q[i] > q[t] || (q[i] == q[t] && a[i] < a[t])) t = i;
    double x[3];
    for (int i = 0; i < 3; i++) {
        const double d = q[i] - q[t];
        x[i] = i == t ? 1.0 : (d > -746.0 ? exp(d) : 0.0);
    }
    return triple_pick_weights(x, a, A, u, 1e-11);
}

// _normalize_log_probs + choice for a scanned cell whose maximum (`top`,
// value ptop) and runner-up value are known: the arithmetic of the plain scan
// (exp() only where its result is not known, see sweep_window)
static inline int64_t scan_pick_full(const double *post, int64_t A,
                                     int64_t top, double ptop, double second,
                                     double u, double *cdf, bool shortcuts)
{
    double run = 0.0, tail = 0.0;
    const double cut = shortcuts ? (second - ptop) - 60.0 : -INFINITY;
    for (int64_t a = 0; a <= A; a++) {
        if (a == top) continue;
        const double d = post[a] - ptop;
        if (d > -746.0 && d >= cut) tail += exp(d);
    }
    const double lnorm = log1p(tail);
    for (int64_t a = 0; a <= A; a++) {
        const double v = post[a] - ptop - lnorm;
        if (v <= LOG_EPS) run += EXP_LOG_EPS;
        else run += exp(v > 0.0 ? 0.0 : v);
        cdf[a] = run;
    }
    const double total = cdf[A];
    int64_t lo = 0, hi = A + 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cdf[mid] / total > u) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

// Checker hook: the pick of a cell with three live entries.  quick = 1: the
// one-exp-per-entry decision (-1 = not decided); quick = 0: the scan's own
// arithmetic over the full row (the three values at their positions, -1e4
// everywhere else).
extern "C" int bnpc_triple_pick(int quick, const double *q, const int64_t *a,
                                int64_t A, double u, int64_t *pick)
{
    if (!pick || !q || !a || A < 2 || A > 4096) {
        bnpc_set_error("bad argument: triple_pick");
        return 2;
    }
    for (int i = 0; i < 3; i++)
        if (a[i] < 0 || a[i] > A || !(q[i] == q[i])
            || a[i] == a[(i + 1) % 3]) {
            bnpc_set_error("bad argument: triple_pick");
            return 2;
        }
    if (quick == 2) {
        // as the sweep loop decides it from a hint record: the weights of
        // the second / third entry relative to the first as float32 (the
        // record's e2 / e3 - here the priors are inside them), band 1e-6
        const double w[3] = {1.0, (double)(float)exp(q[1] - q[0]),
                             (double)(float)exp(q[2] - q[0])};
        *pick = (w[1] < 1e30 && w[2] < 1e30)
            ? triple_pick_weights(w, a, A, u, 1e-6) : -1;
        return 0;
    }
    if (quick) {
        *pick = triple_pick_quick(q, a, A, u);
        return 0;
    }
    std::vector<double> post((size_t)A + 1, -1e4), cdf((size_t)A + 1);
    for (int i = 0; i < 3; i++) post[(size_t)a[i]] = q[i];
    int64_t top = 0;
    double best = -INFINITY, second = -INFINITY;
    for (int64_t i = 0; i <= A; i++) {
        const double v = post[i];
        if (v > best) {
            second = best;
            best = v;
            top = i;
        } else if (v > second) {
            second = v;
        }
    }
    *pick = scan_pick_full(post.data(), A, top, best, second, u, cdf.data(),
                           loop_shortcuts());
    return 0;
}

// Checker hooks for the four functions above (tests/test_native_sweeps.py):
// -1 from a quick variant = "not decided here".
extern "C" int bnpc_pair_pick(int quick, double d2, int64_t A, int64_t top,
                              int64_t sec, double u, int64_t *pick)
{
    if (!pick || A < 1 || A > 4096 || top < 0 || top > A || sec < 0
        || sec > A || top == sec || !(d2 <= 0.0)) {
        bnpc_set_error("bad argument: pair_pick");
        return 2;
    }
    std::vector<double> cdf((size_t)A + 1);
    if (quick == 2)     // the loop's form: float32 weight, band 1e-6
        *pick = pair_pick_weights(1.0, (double)(float)exp(d2), top, sec, A, u,
                                  1e-6);
    else
        *pick = quick ? pair_pick_quick(d2, A, top, sec, u)
                      : pair_pick_full(d2, A, top, sec, u, cdf.data());
    return 0;
}

extern "C" int bnpc_two_way_pick(int quick, double p0, double p1, double u,
                                 int64_t *pick)
{
    if (!pick || !(p0 == p0) || !(p1 == p1)) {
        bnpc_set_error("bad argument: two_way_pick");
        return 2;
    }
    const int big = p1 > p0 ? 1 : 0;
    const double d = big ? p0 - p1 : p1 - p0;
    double l0, l1;
    *pick = quick ? two_way_pick_quick(d, big, u)
                  : two_way_pick_full(d, big, u, &l0, &l1);
    return 0;
}

// live clusters up to which a cell is decided between its two / among its three
// candidate columns from the hint alone: the interval ends the quick picks
// compare the uniform with carry one rounding (~1e-16) per floor entry summed,
// 4096 of them stay a factor 20 inside the 1e-11 band the picks keep clear of
// (the checker hooks bnpc_pair_pick / bnpc_triple_pick take the same range)
static const int64_t QUICK_PICK_MAX = 4096;

// The LANE of the cells a hint record decides, in a whole-matrix sweep: in a
// converged chain nearly every cell is decided by its record - two in three
// (config 3) or 98 in 100 (configs 4, 5) by its first column beyond doubt, the
// rest picked between its two or among its three best columns - and what
// sweep_window's general iteration spends on them (10.9 ns per dominated cell,
// 40-50 per pick, measured on the GPU box's EPYC: tools/hinted_loop_bench.py)
// is its own generality: a hundred and sixty instructions per cell, half of
// its variables on the stack.  Here the same steps for exactly those cases,
// in the same order with the same arithmetic: the cell leaves its cluster
// (sizes, prior, drift), the dominance test of its record under the drift,
// else the pair test, else the triple test, the cell's uniform, the pick, the
// move - 7.2 ns per dominated cell, 33 / 40 per pick.  The lane stops in front
// of anything else - a cell whose cluster would die, odd input, the end of
// the window: nothing done, `removed` = false - or in the middle of a cell it
// cannot decide: the cell has left its cluster (`removed`) and the general
// iteration goes on from there, with the cell's uniform if that has been drawn
// (`have_u`: a dominated pick in the 1e-11 slivers at either end, a pick
// within the band of an interval end).  No cluster is born or dies in here,
// so the live list, its positions and the dominance margin are constants of
// a run of the lane.
struct RecordLane {
    // constants of a run
    const int64_t *perm;
    const bnpc_top2 *hint;      // narrow records
    bool in_order;              // hint[p] is the cell at position p
    const double *post_new, *crp_prior, *cpr0;
    int64_t hint_cols, N, n_cols, A, pos_end;
    double margin;              // dom_bound[A]
    double sliver;              // 1e-11 (sweep_lane_sliver)
    double band;                // 1e-6: of the two- / three-candidate picks
    const int64_t *pos64;       // position of a column in the live list, or
    const int32_t *pos32;       // the same as 32-bit entries (one is NULL)
    int64_t *assignment;
    const int64_t *col_of_id, *col_id, *order;
    int64_t *col_size;
    double *cpr;
    bool weights;               // the prior table is log(size) + const: the
                                // two- and three-candidate picks as well
    // the stride (below): folded[c] != 0 - column c's prior at its current
    // size and at one cell less is in `drift` already
    uint8_t *folded;
    bool stride;
    // state
    int64_t pos;
    double drift;
    int64_t decided;            // cells the lane has moved ...
    int64_t strided;            // ... of them whole runs at a time (the stride)
    int64_t pairs, triples;     // ... of them between two / among three
    // the cell it stopped in
    bool removed, have_u;
    int64_t cell;
    double u;
};

static void record_lane(RecordLane &L, bnpc_mt19937 *rng)
{
    const int64_t *__restrict__ perm = L.perm;
    const bnpc_top2 *__restrict__ hint = L.hint;
    const double *__restrict__ post_new = L.post_new;
    const double *__restrict__ crp_prior = L.crp_prior;
    const double *__restrict__ cpr0 = L.cpr0;
    int64_t *assignment = L.assignment;
    const int64_t *col_of_id = L.col_of_id, *col_id = L.col_id;
    const int64_t *order = L.order;
    int64_t *col_size = L.col_size;
    double *cpr = L.cpr;
    const int64_t hint_cols = L.hint_cols, N = L.N, n_cols = L.n_cols;
    const int64_t pos_end = L.pos_end;
    const double margin = L.margin, sliver = L.sliver, band = L.band;
    const bool in_order = L.in_order;
    const int64_t A = L.A;
    const bool weights = L.weights;
    int64_t pos = L.pos, decided = 0, pairs = 0, triples = 0;
    double drift = L.drift;
    L.removed = L.have_u = false;
    auto pos_of = [&](int64_t c) -> int64_t {
        return L.pos64 ? L.pos64[c] : (int64_t)L.pos32[c];
    };
    // THE STRIDE (round 6).  Nearly every cell of a converged sweep is
    // dominated by the cluster it sits in: it leaves, its record's first
    // column - its own - beats everything else by more than the dominance
    // margin under any drift, its uniform lies clear of 0 and 1, it rejoins.
    // Nothing of that depends on the cell before it, so a RUN of such cells
    // is taken at once: per cell two gathers (label -> column, the
    // new-cluster term), three compares against ONE threshold that holds for
    // every drift the run can see, its uniform peeked - tempered straight
    // out of the state block, not yet consumed - and compared with the
    // slivers; the run ends in front of the first cell that fails anything,
    // and that cell takes the lane's own steps below with nothing drawn.
    //   The threshold: the lane accepts when second + d - (best - d) < margin
    // and post_new - (best - d) < margin, d = the drift when the cell has
    // left.  While every cell of a run stays, sizes are what they are, so d
    // never exceeds D = max(drift, over the live hint columns c:
    // |prior[size_c] - prior0_c|, |prior[size_c - 1] - prior0_c|); best -
    // second > 2 D - margin and best - post_new > 2 D - margin (+ 1e-6: the
    // two forms round differently) imply both for any d <= D.  `dmax` keeps D
    // from above (it only grows while the lane runs: conservative).
    //   What a staying cell leaves behind - the two priors of its column
    // folded into the drift - is folded once per column and size (`folded`).
    // 7.2 -> ~2 ns per dominated cell (tools/hinted_loop_bench.py).
    double dmax = drift;
    const bool stride = L.stride && L.folded;
    uint8_t *folded = L.folded;
    if (stride) {
        // (the general iteration may have changed any size since the lane
        // last ran: nothing counts as folded)
        memset(folded, 0, (size_t)hint_cols);
        for (int64_t a = 0; a < A; a++) {
            const int64_t c = order[a];
            if (c >= hint_cols) continue;
            const int64_t sz = col_size[c];
            double d = fabs(crp_prior[sz] - cpr0[c]);
            if (d > dmax) dmax = d;
            if (sz >= 1) {
                d = fabs(crp_prior[sz - 1] - cpr0[c]);
                if (d > dmax) dmax = d;
            }
        }
    }
    int64_t strided = 0;
    constexpr int STRIDE = 32;
    // (a column whose size has changed: its priors are folded again)
#define LANE_RESIZED(c_)                                                      \
    if (stride && (c_) < hint_cols) {                                         \
        folded[c_] = 0;                                                       \
        const int64_t sz_ = col_size[c_];                                     \
        double d_ = fabs(crp_prior[sz_] - cpr0[c_]);                          \
        if (d_ > dmax) dmax = d_;                                             \
        if (sz_ >= 1) {                                                       \
            d_ = fabs(crp_prior[sz_ - 1] - cpr0[c_]);                         \
            if (d_ > dmax) dmax = d_;                                         \
        }                                                                     \
    }
    while (pos < pos_end) {
        if (stride) {
            if (rng->pos >= 624) mt_refill(rng);
            int64_t B = pos_end - pos;
            if (B > STRIDE) B = STRIDE;
            if (B > (624 - rng->pos) / 2) B = (624 - rng->pos) / 2;
            if (drift > dmax) dmax = drift;
            const double thr = 2.0 * dmax - margin + 1e-6;
            int32_t cols[STRIDE];
            int64_t n_ok = 0;
            for (; n_ok < B; n_ok++) {
                const int64_t p = pos + n_ok;
                const uint64_t cell = (uint64_t)perm[p];
                if (cell >= (uint64_t)N) break;
                const uint64_t id = (uint64_t)assignment[cell];
                if (id >= (uint64_t)N) break;
                const int64_t c = col_of_id[id];
                if ((uint64_t)c >= (uint64_t)hint_cols) break;
                const bnpc_top2 &h = hint[in_order ? p : (int64_t)cell];
                if ((int64_t)h.col != c || col_size[c] < 2
                    || !(h.best - h.second > thr)
                    || !(h.best - post_new[cell] > thr))
                    break;
                // its uniform, as mt_double will form it
                const uint32_t *k = rng->key + rng->pos + 2 * n_ok;
                uint32_t y0 = k[0], y1 = k[1];
                y0 ^= (y0 >> 11);
                y0 ^= (y0 << 7) & 0x9d2c5680u;
                y0 ^= (y0 << 15) & 0xefc60000u;
                y0 ^= (y0 >> 18);
                y1 ^= (y1 >> 11);
                y1 ^= (y1 << 7) & 0x9d2c5680u;
                y1 ^= (y1 << 15) & 0xefc60000u;
                y1 ^= (y1 >> 18);
                const double u = ((int32_t)(y0 >> 5) * 67108864.0
                                  + (int32_t)(y1 >> 6)) / 9007199254740992.0;
                if (!(u > sliver && u < 1.0 - sliver)) break;
                cols[n_ok] = (int32_t)c;
            }
            if (n_ok > 0) {
                rng->pos += (int32_t)(2 * n_ok);
                for (int64_t i = 0; i < n_ok; i++) {
                    const int64_t c = cols[i];
                    if (folded[c]) continue;
                    folded[c] = 1;
                    const int64_t sz = col_size[c];
                    double d = fabs(crp_prior[sz - 1] - cpr0[c]);
                    if (d > drift) drift = d;
                    d = fabs(crp_prior[sz] - cpr0[c]);
                    if (d > drift) drift = d;
                }
                pos += n_ok;
                decided += n_ok;
                strided += n_ok;
                if (n_ok == B) continue;    // (else: the cell that ended the run)
                if (pos >= pos_end) break;
            }
        }
        const int64_t cell = perm[pos];
        if (pos + 16 < pos_end) {
            const uint64_t ahead = (uint64_t)perm[pos + 16];
            if (ahead < (uint64_t)N) {
                if (!in_order) __builtin_prefetch(&hint[ahead], 0, 1);
                __builtin_prefetch(&assignment[ahead], 1, 1);
                __builtin_prefetch(&post_new[ahead], 0, 1);
            }
        }
        if ((uint64_t)cell >= (uint64_t)N) break;
        const int64_t old_id = assignment[cell];
        if ((uint64_t)old_id >= (uint64_t)N) break;
        const int64_t old_col = col_of_id[old_id];
        if ((uint64_t)old_col >= (uint64_t)n_cols) break;
        const int64_t old_size = col_size[old_col];
        if (old_size < 2) break;        // its cluster dies (or odd input)
        // the cell leaves its cluster (libs/CRP.py:262-266)
        col_size[old_col] = old_size - 1;
        const double left = crp_prior[old_size - 1];
        cpr[old_col] = left;
        if (old_col < hint_cols) {
            const double d = fabs(left - cpr0[old_col]);
            if (d > drift) drift = d;
        }
        // its record's first column against everything else, widened by how
        // far the priors have moved since the launch
        const bnpc_top2 &h = hint[in_order ? pos : cell];
        const int64_t hc = h.col;
        int64_t at = -1;
        if ((uint64_t)hc < (uint64_t)hint_cols && col_size[hc] > 0)
            at = pos_of(hc);
        double other = h.second + drift;
        const double pn = post_new[cell];
        if (pn > other) other = pn;
        int64_t pick = -1;
        bool drawn = false;
        double u = 0.0;
        if (at >= 0 && other - (h.best - drift) < margin) {
            u = mt_double(rng);
            drawn = true;
            if (u > sliver && u < 1.0 - sliver) pick = at;
        } else if (weights && at >= 0) {
            // Not dominated: torn between its record's two best columns,
            // re-scored under the current priors, everything else more than
            // 61 below the lower one (sweep_window's pair test), ...
            const int64_t c2 = h.col2, c3 = h.col3;
            int64_t a2 = -1, a3 = -1;
            if ((uint64_t)c2 < (uint64_t)hint_cols && c2 != hc
                && col_size[c2] > 0)
                a2 = pos_of(c2);
            if (a2 >= 0) {
                const double q1 = h.ll_best + cpr[hc];
                const double q2 = h.ll_second + cpr[c2];
                double rest = (double)h.third + drift;
                if (pn > rest) rest = pn;
                const double low = q1 < q2 ? q1 : q2;
                if (q1 > -INFINITY && q1 < INFINITY && q2 > -INFINITY
                    && q2 < INFINITY && rest < low - 61.0) {
                    if (h.e2 < 1e30f) {
                        const bool first = q1 > q2 || (q1 == q2 && at < a2);
                        const int64_t top = first ? at : a2;
                        const double gap = first ? q2 - q1 : q1 - q2;
                        u = mt_double(rng);
                        drawn = true;
                        if (gap < margin) {
                            // (the runner-up beyond the dominance margin)
                            if (u > sliver && u < 1.0 - sliver) pick = top;
                        } else {
                            pick = pair_pick_weights((double)col_size[hc],
                                (double)h.e2 * (double)col_size[c2], at, a2, A,
                                u, band);
                        }
                        if (pick >= 0) pairs++;
                    }
                } else if (A >= 2 && (uint64_t)c3 < (uint64_t)hint_cols
                           && c3 != hc && c3 != c2 && col_size[c3] > 0
                           && (a3 = pos_of(c3)) >= 0) {
                    // ... or among its three best, everything else more than
                    // 61 below the middle one (the triple test)
                    const double q3 = h.ll_third + cpr[c3];
                    double rest4 = (double)h.fourth + drift;
                    if (pn > rest4) rest4 = pn;
                    const double hi3 = q1 > q2 ? (q1 > q3 ? q1 : q3)
                                               : (q2 > q3 ? q2 : q3);
                    const double lo3 = q1 < q2 ? (q1 < q3 ? q1 : q3)
                                               : (q2 < q3 ? q2 : q3);
                    const double mid3 = q1 + q2 + q3 - hi3 - lo3;
                    if (lo3 > -INFINITY && hi3 < INFINITY
                        && rest4 < mid3 - 61.0 && h.e2 < 1e30f
                        && h.e3 < 1e30f) {
                        const double w[3] = {(double)col_size[hc],
                            (double)h.e2 * (double)col_size[c2],
                            (double)h.e3 * (double)col_size[c3]};
                        const int64_t a[3] = {at, a2, a3};
                        u = mt_double(rng);
                        drawn = true;
                        pick = triple_pick_weights(w, a, A, u, band);
                        if (pick >= 0) triples++;
                    }
                }
            }
        }
        if (pick < 0) {
            // not for this lane: the general iteration goes on with the cell
            // (and with its uniform, if that has been drawn)
            L.removed = true;
            L.have_u = drawn;
            L.cell = cell;
            L.u = u;
            break;
        }
        // it joins (or stays in) the cluster picked
        const int64_t c = order[pick];
        assignment[cell] = col_id[c];
        const int64_t grown = col_size[c] + 1;
        col_size[c] = grown;
        const double now = crp_prior[grown];
        cpr[c] = now;
        if (c < hint_cols) {
            const double d = fabs(now - cpr0[c]);
            if (d > drift) drift = d;
        }
        if (c != old_col) {         // two columns have another size now
            LANE_RESIZED(old_col)
            LANE_RESIZED(c)
        }
        pos++;
        decided++;
    }
#undef LANE_RESIZED
    L.pos = pos;
    L.drift = drift;
    L.decided = decided;
    L.strided = strided;
    L.pairs = pairs;
    L.triples = triples;
}

static int sweep_window(bnpc_gibbs_state *st, bnpc_mt19937 *rng,
                        const int64_t *perm, const double *ll,
                        const double *post_new, const double *crp_prior,
                        int64_t *assignment, int64_t *col_of_id,
                        int64_t *col_id, int64_t *col_size, int64_t *order,
                        double *scratch, double *cpr, ParScan *par,
                        int64_t par_min)
{
    const int64_t N = st->n_cells, ld = st->ld;
    double *post = scratch;             // ld + 1
    double *cdf = scratch + ld + 1;     // ld + 1
    st->new_cell = -1;
    if (st->pos_end > N || st->pos > st->pos_end ||
        (st->row_base >= 0 && st->row_base > st->pos)) {
        bnpc_set_error("bad sweep window [%lld, %lld) / row_base %lld",
                       (long long)st->pos, (long long)st->pos_end,
                       (long long)st->row_base);
        return 2;
    }

    // both tables are thread-local and grow on demand: fetched once, for the
    // largest cluster count this call can see (at most ld live columns)
    const double *fs = floor_sums(ld + 1);
    const double *dom_bound = dominated_bounds(ld + 1);
    const bool shortcuts = loop_shortcuts();
    const int64_t ahead_by = 16;    // cells the loop prefetches ahead
    for (int64_t c = 0; c < st->n_cols && c < ld; c++) {
        const int64_t sz = col_size[c];
        cpr[c] = (sz >= 0 && sz <= N + 1) ? crp_prior[sz] : 0.0;
    }

    // The hint (include/bnpc_hip.h: bnpc_top2): the largest entries of every
    // row under the priors at launch.  `drift` bounds how far any column's
    // prior has moved since; with it the hint decides a cell without scanning
    // it when best - drift beats everything else + drift (and the new-cluster
    // entry, and the columns born since, which are looked at) by more than
    // the dominance margin.  Two kinds: NARROW hints (a whole-matrix sweep,
    // up to 32767 columns - a converged chain with a dozen clusters or with
    // hundreds, the first sweep of a data set whose matrix fits the host
    // budget: rows indexed by cell, the pair / triple tests apply) and WIDE
    // hints (any number of columns, the tiles
    // of a first sweep: rows indexed by tile position, the column of the
    // largest entry as a 32-bit number, the dominance test only - from the
    // moment the true clusters have been born nearly every cell is dominated
    // by one of them and the 30 000 random ones need not be walked).
    const bool tile_rows = st->row_base >= 0;
    const bnpc_top2 *hint = (st->hint && st->hint_prior && st->hint_cols > 0
                             && (tile_rows || st->hint_cols <= 32767)
                             && st->hint_cols <= st->n_cols && FLOOR_STEP > 0)
        ? st->hint : nullptr;
    const int64_t hint_cols = hint ? st->hint_cols : 0;
    const bool narrow = hint && !tile_rows;
    // (the records of a whole-matrix sweep in visiting order: hint[p] is the
    // cell at position p, bnpc_hints_in_order_issue)
    const bool in_order = narrow && st->hint_in_order;
    const int64_t hint_rows = !hint ? 0
        : (tile_rows ? st->pos_end - st->row_base : N);
    // The hints sit in pinned memory the device has just written: every line
    // of them is a miss all the way to DRAM, and the loop visits them in
    // permutation order - one exposed miss per cell, which software prefetch
    // hides only in part (measured: 140-230 cycles per decided cell, best at
    // a prefetch distance of 4, against ~40 from a private copy).  One
    // sequential pass at the start of the sweep / tile (the hardware
    // prefetcher streams it: 240 KB in 7 us) into a block that stays in L2
    // costs a twentieth of that.
    static thread_local std::vector<bnpc_top2> hint_local;
    static thread_local const bnpc_top2 *hint_local_of = nullptr;
    // (up to 1 MiB of hints, 16 384 cells: at config 5's 3.2 MB the pass
    // itself runs at 2 GB/s and costs more than the misses it saves - Gibbs
    // step 3.6-4.5 against 2.05 ms - while 640 KB at config 4 still gain)
    // (records in visiting order - 3.2 MB of them at config 5 - are read
    // front to back by the loop itself; a small set is copied all the same:
    // the copy streams at 30 GB/s, the loop's reads wait for every line)
    if (hint && (size_t)hint_rows * sizeof(bnpc_top2) <= ((size_t)1 << 20)) {
        // (a sweep resumed after a birth in the caller finds its copy; the
        // pinned buffers of tiles are re-used, so a tile copies at its start)
        const int64_t start = tile_rows ? st->row_base : 0;
        if (st->pos == start || hint_local_of != hint
            || (int64_t)hint_local.size() != hint_rows) {
            hint_local.resize((size_t)hint_rows);
            memcpy(hint_local.data(), hint,
                   (size_t)hint_rows * sizeof(bnpc_top2));
            hint_local_of = hint;
        }
        hint = hint_local.data();
    }
    // The quick picks among two / three candidates weigh them by e * size
    // (bnpc_top2.e2 / e3 times the current size of the column's cluster), which
    // is right when the prior table is log(size) + a common term, as the
    // reference's is (libs/CRP.py:83-85).  A table of another shape (a caller
    // of the C-ABI is free to pass one) is told by a few samples and gets the
    // exponentials of the re-scored entries instead.
    bool size_weights = narrow && shortcuts;
    if (size_weights) {
        const int64_t probe[6] = {2, 3, 5, 64, N / 2, N};
        for (int i = 0; i < 6 && size_weights; i++) {
            const int64_t m = probe[i] < 1 ? 1 : (probe[i] > N ? N : probe[i]);
            const double d = crp_prior[m] - crp_prior[1];
            size_weights = fabs(d - log((double)m)) < 1e-12 * (1.0 + fabs(d));
        }
    }
    const double *cpr0 = st->hint_prior;
    double drift = 0.0;
    int64_t pos_of_col[64];
    // (a table re-made at every death of a cluster: for a few columns only)
    const bool pos_table = narrow && hint_cols <= 64;
    auto index_live = [&]() {
        if (!pos_table) return;
        for (int64_t c = 0; c < hint_cols; c++) pos_of_col[c] = -1;
        for (int64_t a = 0; a < st->n_active; a++)
            if (order[a] < hint_cols) pos_of_col[order[a]] = a;
    };
    // position of a live column in the live list, or -1: a table for hints of
    // up to 64 columns, re-made at every death; for more columns a bisection
    // (the list is ascending in column index; verified by the comparison at
    // the end) - and, while no cluster dies, a table as well: a bisection over
    // 200 live clusters is eight unpredictable branches per cell (a sweep of a
    // converged chain with 200 clusters: 37 ns per cell against 10 with 14),
    // but re-making a table of thousands of entries at every death of a first
    // sweep would cost more than it saves - so the table is made after a
    // stretch of look-ups without a death and dropped at the next one.
    static thread_local std::vector<int32_t> pos_big;
    bool big_valid = false;
    int64_t bisected = 0;
    auto pos_of = [&](int64_t c) -> int64_t {
        if (pos_table) return pos_of_col[c];
        if (narrow) {
            if (big_valid) return pos_big[(size_t)c];
            if (++bisected > st->n_active / 16 + 8) {
                pos_big.assign((size_t)hint_cols, -1);
                for (int64_t a = 0; a < st->n_active; a++)
                    if (order[a] < hint_cols)
                        pos_big[(size_t)order[a]] = (int32_t)a;
                big_valid = true;
                bisected = 0;
                return pos_big[(size_t)c];
            }
        }
        int64_t a = 0, b = st->n_active;
        while (a < b) {
            const int64_t mid = (a + b) >> 1;
            if (order[mid] < c) a = mid + 1;
            else b = mid;
        }
        return (a < st->n_active && order[a] == c) ? a : -1;
    };
    if (hint) {
        for (int64_t c = 0; c < hint_cols; c++)
            if (col_size[c] > 0) {
                const double d = fabs(cpr[c] - cpr0[c]);
                if (d > drift) drift = d;
            }
        index_live();
    }
    // the matrix may still be on its way from the device (it is only read
    // where a hint is in doubt): wait for it before the first read
    bool matrix_ready = st->matrix_wait == nullptr;
#define NEED_MATRIX()                                                         \
    if (!matrix_ready) {                                                      \
        if (st->matrix_wait(st->matrix_wait_arg)) {                           \
            bnpc_set_error("waiting for the log-likelihood matrix failed");   \
            return 5;                                                         \
        }                                                                     \
        matrix_ready = true;                                                  \
        st->matrix_wait = nullptr;                                            \
    }
#define NOTE_PRIOR(c_)                                                        \
    if ((c_) < hint_cols) {                                                   \
        const double d_ = fabs(cpr[c_] - cpr0[c_]);                           \
        if (d_ > drift) drift = d_;                                           \
    }

    int64_t free_hint = 0;      // no id below it is free (native births)
    // the lane of the dominated cells (record_lane): whole-matrix hints,
    // the shortcuts on; entered whenever no column has been born since the
    // launch, the live list has its position table and at most 4096 entries
    // (the quick picks' range; sweep_lane_sliver for the dominated pick)
    RecordLane lane;
    const double lane_sliver = sweep_lane_sliver();
    const bool lane_usable = narrow && shortcuts && lane_sliver > 0.0;
    if (lane_usable) {
        lane.perm = perm;
        lane.hint = hint;
        lane.in_order = in_order;
        lane.post_new = post_new;
        lane.crp_prior = crp_prior;
        lane.cpr0 = cpr0;
        lane.hint_cols = hint_cols;
        lane.N = N;
        lane.pos_end = st->pos_end;
        lane.assignment = assignment;
        lane.col_of_id = col_of_id;
        lane.col_id = col_id;
        lane.order = order;
        lane.col_size = col_size;
        lane.cpr = cpr;
        lane.weights = size_weights;
        lane.sliver = lane_sliver;
        // (the test switch widens the band of the picks as well)
        lane.band = lane_sliver > 1e-6 ? lane_sliver : 1e-6;
        static thread_local std::vector<uint8_t> folded;
        folded.assign((size_t)hint_cols + 1, 0);
        lane.folded = folded.data();
        lane.stride = sweep_lane_stride();
    }
    while (st->pos < st->pos_end) {
        bool lane_removed = false, lane_have_u = false;
        double lane_u = 0.0;
        if (lane_usable && st->n_active >= 1
            && st->n_active <= QUICK_PICK_MAX
            && order[st->n_active - 1] < hint_cols
            && (pos_table || big_valid)) {
            lane.n_cols = st->n_cols;
            lane.A = st->n_active;
            lane.margin = dom_bound[st->n_active];
            lane.pos64 = pos_table ? pos_of_col : nullptr;
            lane.pos32 = pos_table ? nullptr : pos_big.data();
            lane.pos = st->pos;
            lane.drift = drift;
            record_lane(lane, rng);
            st->pos = lane.pos;
            drift = lane.drift;
            st->hint_used += lane.decided;
            st->lane_used += lane.decided;
            st->stride_used += lane.strided;
            st->pair_used += lane.pairs;
            st->triple_used += lane.triples;
            if (!lane.removed) {
                if (st->pos >= st->pos_end) break;
            } else {
                lane_removed = true;
                lane_have_u = lane.have_u;
                lane_u = lane.u;
            }
        }
        const int64_t cell = perm[st->pos];
        // rows are visited in permutation order: pull the row (and the
        // per-cell scalars) of a cell a few positions ahead into the cache
        if (ahead_by > 0 && st->pos + ahead_by < st->pos_end) {
            const int64_t ahead = perm[st->pos + ahead_by];
            if (ahead >= 0 && ahead < N) {
                const char *r = (const char *)(ll + (size_t)(
                    st->row_base >= 0 ? st->pos + ahead_by - st->row_base
                                      : ahead) * ld);
                if (hint) {
                    // the row itself is only read if the hint is in doubt
                    __builtin_prefetch(&hint[tile_rows
                        ? st->pos + ahead_by - st->row_base
                        : (in_order ? st->pos + ahead_by : ahead)], 0, 1);
                } else {
                    const size_t bytes = (size_t)st->n_cols * sizeof(double);
                    const char *end = r + (bytes < 512 ? bytes : 512);
                    for (const char *q =
                             (const char *)((uintptr_t)r & ~(uintptr_t)63);
                         q < end; q += 64)
                        __builtin_prefetch(q, 0, 1);
                }
                __builtin_prefetch(&assignment[ahead], 1, 1);
                __builtin_prefetch(&post_new[ahead], 0, 1);
            }
        }
        if (cell < 0 || cell >= N) {
            bnpc_set_error("perm[%lld] out of range", (long long)st->pos);
            return 2;
        }
        // remove the cell from its cluster (CRP.py:262-266) - unless the
        // lane has (it stopped at this cell because it is not dominated)
        const int64_t old_id = assignment[cell];
        const int64_t old_col = (old_id >= 0 && old_id < N) ?
            col_of_id[old_id] : -1;
        if (lane_removed) {
            // (the cell has left its cluster: nothing to do here)
        } else if (old_col < 0 || old_col >= st->n_cols
                   || col_size[old_col] < 1) {
            bnpc_set_error("cell %lld sits in unknown cluster %lld",
                           (long long)cell, (long long)old_id);
            return 3;
        }
        if (lane_removed) {
            // (sizes, prior and drift are the lane's)
        } else if (col_size[old_col] == 1) {
            // the live list is ascending in column index (columns are handed
            // out in increasing order and deletions keep the order): bisect,
            // and fall back to a walk should that ever not hold
            int64_t a = 0, b = st->n_active;
            while (a < b) {
                const int64_t mid = (a + b) >> 1;
                if (order[mid] < old_col) a = mid + 1;
                else b = mid;
            }
            if (a >= st->n_active || order[a] != old_col) {
                a = 0;
                while (a < st->n_active && order[a] != old_col) a++;
            }
            memmove(order + a, order + a + 1,
                    (size_t)(st->n_active - a - 1) * sizeof(int64_t));
            st->n_active--;
            col_size[old_col] = 0;
            col_of_id[old_id] = -1;
            if (old_id < free_hint) free_hint = old_id;
            if (hint) index_live();
            big_valid = false;
            bisected = 0;
        } else {
            col_size[old_col]--;
            cpr[old_col] = crp_prior[col_size[old_col]];
            NOTE_PRIOR(old_col)
        }

        // log posterior of joining each live cluster / a new one (:268-274)
        const int64_t A = st->n_active;
        const double *row = ll + (size_t)(st->row_base >= 0 ?
            st->pos - st->row_base : cell) * ld;
        int64_t top = 0;                    // first maximum
        double best = -INFINITY;
        double second = -INFINITY;          // largest entry that is not `top`
        bool hinted = false;
        const int64_t hrow = tile_rows ? st->pos - st->row_base
            : (in_order ? st->pos : cell);
        if (hint) {
            const bnpc_top2 &h = hint[hrow];
            // (a wide record carries its column as 32 bits in col | col2)
            const int64_t hc = narrow ? (int64_t)h.col
                : (int64_t)((uint32_t)(uint16_t)h.col
                            | ((uint32_t)(uint16_t)h.col2 << 16));
            const double pn = post_new[cell];
            // the columns born since the launch: looked at one by one
            double late1 = -INFINITY, late2 = -INFINITY;
            int64_t late_at = -1;
            for (int64_t a = A - 1; a >= 0 && order[a] >= hint_cols; a--) {
                NEED_MATRIX()
                const double v = row[order[a]] + cpr[order[a]];
                if (v >= late1) {       // (>=: the first in list order wins)
                    late2 = late1;
                    late1 = v;
                    late_at = a;
                } else if (v > late2) {
                    late2 = v;
                }
            }
            if (hc >= 0 && hc < hint_cols && col_size[hc] > 0) {
                const int64_t at = pos_of(hc);
                double other = h.second + drift;
                if (pn > other) other = pn;
                if (late1 > other) other = late1;
                if (at >= 0 && other - (h.best - drift) < dom_bound[A]) {
                    top = at;
                    hinted = true;
                    st->hint_used++;
                }
            }
            if (!hinted && late_at >= 0) {
                // ... or one of the columns born since dominates everything
                // the launch knew (its largest entry, widened by the drift)
                double other = h.best + drift;
                if (pn > other) other = pn;
                if (late2 > other) other = late2;
                if (other - late1 < dom_bound[A]) {
                    top = late_at;
                    hinted = true;
                    st->hint_used++;
                }
            }
        }
        // Not dominated - typically a cell torn between the two halves of a
        // freshly split cluster, which are close to each other and far above
        // everything else.  If the row's two best columns, re-scored under
        // the CURRENT priors exactly as the scan would (ll + prior), leave
        // everything else - the third entry at launch widened by the drift,
        // the new-cluster entry, columns born since - more than 61 below the
        // lower of the two, the scan below would give every other entry no
        // exp() (the 60-below-the-runner-up rule) and the floor probability:
        // the cell is decided from the two entries, bit for bit as the scan
        // decides it, without reading its row.
        bool pair = false, pair_w = false;
        int64_t pair_second = 0;
        double pair_wtop = 0.0, pair_wsec = 0.0;
        if (narrow && !hinted && A <= QUICK_PICK_MAX && shortcuts) {
            const bnpc_top2 &h = hint[hrow];
            const int64_t c1 = h.col, c2 = h.col2;
            int64_t a1 = -1, a2 = -1;
            if (c1 >= 0 && c1 < hint_cols && c2 >= 0 && c2 < hint_cols
                && c1 != c2 && col_size[c1] > 0 && col_size[c2] > 0
                && (a1 = pos_of(c1)) >= 0 && (a2 = pos_of(c2)) >= 0) {
                const double q1 = h.ll_best + cpr[c1];
                const double q2 = h.ll_second + cpr[c2];
                double other = h.third + drift;
                const double pn = post_new[cell];
                if (pn > other) other = pn;
                for (int64_t a = A - 1; a >= 0 && order[a] >= hint_cols; a--) {
                    NEED_MATRIX()
                    const double v = row[order[a]] + cpr[order[a]];
                    if (v > other) other = v;
                }
                const double low = q1 < q2 ? q1 : q2;
                if (q1 > -INFINITY && q1 < INFINITY && q2 > -INFINITY
                    && q2 < INFINITY && other < low - 61.0) {
                    // first maximum in list order wins a tie, as in the scan
                    const bool first = q1 > q2 || (q1 == q2 && a1 < a2);
                    top = first ? a1 : a2;
                    pair_second = first ? a2 : a1;
                    best = first ? q1 : q2;
                    second = first ? q2 : q1;
                    pair = true;
                    // (weights of the entries at a1 / a2 for the quick pick; a
                    // float that overflowed or a table of another shape: none)
                    pair_w = size_weights && h.e2 < 1e30f;
                    if (pair_w) {
                        pair_wtop = first ? (double)col_size[c1]
                            : (double)h.e2 * (double)col_size[c2];
                        pair_wsec = first ? (double)h.e2 * (double)col_size[c2]
                            : (double)col_size[c1];
                    }
                    st->hint_used++;
                    st->pair_used++;
                }
            }
        }
        // Three candidates (a cluster split twice): the same reasoning with
        // the hint's third column - everything else (the fourth entry at
        // launch widened by the drift, the new-cluster entry, columns born
        // since) more than 61 below the runner-up.  The cell's uniform is
        // drawn now; if it is clear of the interval ends the three entries
        // imply, the pick is known, else the cell is scanned with that same
        // uniform.
        // (have_u: the lane has drawn this cell's uniform already - one of
        // the 1e-11 slivers at either end, for the dominated pick below)
        bool triple = false, have_u = lane_have_u;
        int64_t triple_pick = 0;
        double u_saved = lane_u;
        if (narrow && !hinted && !pair && A <= QUICK_PICK_MAX && A >= 2
            && shortcuts) {
            const bnpc_top2 &h = hint[hrow];
            const int64_t c[3] = {h.col, h.col2, h.col3};
            int64_t a3[3] = {-1, -1, -1};
            bool ok = c[0] != c[1] && c[0] != c[2] && c[1] != c[2];
            for (int i = 0; i < 3 && ok; i++)
                ok = c[i] >= 0 && c[i] < hint_cols && col_size[c[i]] > 0
                    && (a3[i] = pos_of(c[i])) >= 0;
            if (ok) {
                const double q[3] = {h.ll_best + cpr[c[0]],
                                     h.ll_second + cpr[c[1]],
                                     h.ll_third + cpr[c[2]]};
                double other = h.fourth + drift;
                const double pn = post_new[cell];
                if (pn > other) other = pn;
                for (int64_t a = A - 1; a >= 0 && order[a] >= hint_cols; a--) {
                    NEED_MATRIX()
                    const double v = row[order[a]] + cpr[order[a]];
                    if (v > other) other = v;
                }
                // the runner-up: the middle one of the three
                const double hi3 = q[0] > q[1] ? (q[0] > q[2] ? q[0] : q[2])
                                               : (q[1] > q[2] ? q[1] : q[2]);
                const double lo3 = q[0] < q[1] ? (q[0] < q[2] ? q[0] : q[2])
                                               : (q[1] < q[2] ? q[1] : q[2]);
                const double mid3 = q[0] + q[1] + q[2] - hi3 - lo3;
                // (mid3 carries a rounding error of a few ulps: the margin
                // below is 61 against the scan's 60)
                if (lo3 > -INFINITY && hi3 < INFINITY
                    && other < mid3 - 61.0) {
                    if (!have_u) u_saved = mt_double(rng);
                    have_u = true;
                    if (size_weights && h.e2 < 1e30f && h.e3 < 1e30f) {
                        const double w[3] = {(double)col_size[c[0]],
                            (double)h.e2 * (double)col_size[c[1]],
                            (double)h.e3 * (double)col_size[c[2]]};
                        triple_pick = triple_pick_weights(w, a3, A, u_saved,
                                                          1e-6);
                    } else {
                        triple_pick = triple_pick_quick(q, a3, A, u_saved);
                    }
                    if (triple_pick >= 0) {
                        triple = true;
                        st->hint_used++;
                        st->triple_used++;
                    }
                }
            }
        }
        if (!hinted && !pair && !triple) {
            // the scan reads the row: columns of the launch from the matrix
            // - which the hint kernel has written through for the rows it
            // could tell would be scanned - and columns born since, which
            // were written here after the matrix had arrived
            if (!(narrow && hint[hrow].row_here == 1
                  && (A == 0 || order[A - 1] < hint_cols)))
                NEED_MATRIX()
        }
        if (hinted || pair || triple) {
            // `top` is known: nothing else is needed from the row
        } else if (par && A >= par_min) {
            par->row = row;
            par->A = A;
            par->run(ParScan::SCAN);
            for (int r = 0; r < par->T; r++) {
                const ParScan::Part &q = par->part[r];
                if (q.best > best) {
                    second = best > q.second ? best : q.second;
                    best = q.best;
                    top = q.top;
                } else if (q.best > second) {
                    second = q.best;
                }
            }
        } else if (A <= 64) {
            // few clusters (the converged regime): where the maximum sits is
            // data, so selects instead of branches
            for (int64_t a = 0; a < A; a++) {
                const int64_t c = order[a];
                const double v = row[c] + cpr[c];
                post[a] = v;
                const bool gt = v > best;
                const double runner = v > second ? v : second;
                second = gt ? best : runner;
                top = gt ? a : top;
                best = gt ? v : best;
            }
        } else {
            // many clusters: a new maximum is rare, branches predict well
            // and there is no select chain from one entry to the next
            for (int64_t a = 0; a < A; a++) {
                const int64_t c = order[a];
                const double v = row[c] + cpr[c];
                post[a] = v;
                if (v > best) {
                    second = best;
                    best = v;
                    top = a;
                } else if (v > second) {
                    second = v;
                }
            }
        }
        if (!hinted && !pair && !triple) {
            const double v = post_new[cell];
            post[A] = v;
            const bool gt = v > best;
            const double runner = v > second ? v : second;
            second = gt ? best : runner;
            top = gt ? A : top;
            best = gt ? v : best;
        }

        // A posterior without a finite maximum (NaN likelihoods, every entry
        // -inf) has no counterpart here: the reference would go through its
        // FloatingPointError branches (CRP.py:94-98).  Nothing has been drawn
        // for this cell yet; fail loudly instead of opening a cluster.
        if (!hinted && !pair && !triple
            && !(best > -INFINITY && best < INFINITY)) {
            bnpc_set_error("non-finite log posterior for cell %lld "
                           "(maximum %g over %lld clusters)",
                           (long long)cell, best, (long long)A);
            return 4;
        }

        // _normalize_log_probs (CRP.py:88-100) + choice(p=): cdf = cumsum(p);
        // cdf /= cdf[-1]; searchsorted(u, right).
        const double ptop = best;
        const double u_dominated = dom_bound[A];
        int64_t lo = 0, hi = A + 1;       // first a with cdf[a]/total > u
        if (triple) {
            lo = triple_pick;
        } else if (hinted || (FLOOR_STEP > 0 && second - ptop < u_dominated)) {
            // One cluster dominates: the tail sum of exponentials is below
            // 2^-55, so log1p(tail) == tail < half an ulp of 1: the winner's
            // probability is exp(-tail) == 1.0 exactly and every other entry
            // sits below the floor log(1e-15) and is clipped to it.  No
            // exponential has to be evaluated (the usual case once the
            // clusters have separated), with bit-identical probabilities -
            // and the running sums np.cumsum would produce are known without
            // walking them: `top` copies of the floor added one by one
            // (floor_sums), then + 1.0, then every further floor moves the
            // sum, which now lies in [1, 2), by exactly FLOOR_STEP ulps.
            const double u = have_u ? u_saved : mt_double(rng);
            const double at_top = fs[top] + 1.0;
            auto cdf_at = [&](int64_t a) {
                return a < top ? fs[a + 1]
                    : at_top + (double)(FLOOR_STEP * (a - top)) * 0x1p-52;
            };
            const double total = cdf_at(A);
            // the answer is `top` unless u falls into one of the 1e-15
            // slivers (the predicate cdf[a]/total > u is monotone in a).
            // With at most 512 entries the slivers below `top` end under
            // 512e-15 and the ones above it start beyond 1 - 513e-15: a
            // uniform between 1e-12 and 1 - 1e-12 needs no division to know.
            if (shortcuts && A <= 512 && u > 1e-12 && u < 1.0 - 1e-12) {
                lo = top;
            } else if (cdf_at(top) / total > u
                && (top == 0 || !(cdf_at(top - 1) / total > u))) {
                lo = top;
            } else {
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (cdf_at(mid) / total > u) hi = mid;
                    else lo = mid + 1;
                }
            }
        } else {
            // exp() is still skipped where its result is known exactly:
            // exp(d) == 0.0 for d < -746 (below the smallest subnormal), and
            // every entry clipped at log(1e-15) contributes exp(LOG_EPS).
            double run = 0.0;
            double tail = 0.0;
            // (the uniform is drawn here - unless the three-candidate test
            // above has drawn it; nothing below touches the stream)
            const double u = have_u ? u_saved : mt_double(rng);
            bool decided = false;
            if (pair && shortcuts) {
                const int64_t quick = pair_w
                    ? pair_pick_weights(pair_wtop, pair_wsec, top, pair_second,
                                        A, u, 1e-6)
                    : pair_pick_quick(second - ptop, A, top, pair_second, u);
                if (quick >= 0) {
                    lo = quick;
                    decided = true;
                }
            }
            if (decided) {
                // lo holds the pick
            } else if (pair) {
                lo = pair_pick_full(second - ptop, A, top, pair_second, u,
                                    cdf);
                decided = true;
            } else if (par && A >= par_min) {
                // the exponentials on the team, the sums here in index order
                par->top = top;
                par->ptop = ptop;
                par->run(ParScan::EXP_TAIL);
                for (int64_t a = 0; a <= A; a++) tail += cdf[a];
                const double lnorm = log1p(tail);
                par->lnorm = lnorm;
                par->run(ParScan::EXP_PROB);
                for (int64_t a = 0; a <= A; a++) {
                    run += cdf[a];
                    cdf[a] = run;
                }
            } else {
                // ... and an entry more than 60 below the runner-up is not
                // worth its exp() either: the runner-up's term alone is in
                // the sum, up to 64 such entries together stay below 1e-24
                // of it, i.e. 1e-8 ulp of any partial sum that contains it -
                // far inside what libm's exp() and NumPy's differ by (see
                // the note at the top of this file); before the runner-up
                // is added they cannot be told from zeros afterwards.
                lo = scan_pick_full(post, A, top, ptop, second, u, cdf,
                                    shortcuts);
                decided = true;
            }
            if (!decided) {
                const double total = cdf[A];
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (cdf[mid] / total > u) hi = mid;
                    else lo = mid + 1;
                }
            }
        }
        int64_t pick = lo;
        if (pick > A) pick = A;

        st->pos++;
        if (pick == A) {
            // a new cluster: opened here when the call is equipped for it
            // and has room, else in the caller
            if (!st->birth_ctx || st->n_cols >= ld
                || st->n_born >= st->born_cap) {
                st->new_cell = cell;
                return 0;
            }
            NEED_MATRIX()           // its column is about to be written
            const int rc = open_cluster(st, rng, cell, (double *)ll,
                                        assignment, col_of_id, col_id,
                                        col_size, order, &free_hint);
            if (rc) return rc;
            cpr[st->n_cols - 1] = crp_prior[1];
            continue;
        }
        const int64_t c = order[pick];
        assignment[cell] = col_id[c];
        col_size[c]++;
        cpr[c] = crp_prior[col_size[c]];
        NOTE_PRIOR(c)
    }
    return 0;
#undef NOTE_PRIOR
#undef NEED_MATRIX
}

extern "C" int bnpc_gibbs_sweep(bnpc_gibbs_state *st, bnpc_mt19937 *rng,
                                const int64_t *perm, const double *ll,
                                const double *post_new,
                                const double *crp_prior, int64_t *assignment,
                                int64_t *col_of_id, int64_t *col_id,
                                int64_t *col_size, int64_t *order,
                                double *scratch)
{
    if (!st || !rng || !perm || !ll || !post_new || !crp_prior ||
        !assignment || !col_of_id || !col_id || !col_size || !order ||
        !scratch) {
        bnpc_set_error("bad argument: NULL");
        return 2;
    }
    // log prior of joining a column's cluster at its current size
    static thread_local std::vector<double> cpr_store;
    if ((int64_t)cpr_store.size() < st->ld) cpr_store.resize(st->ld);
    double *cpr = cpr_store.data();

    // live clusters from which the scan of a cell goes to the team (read per
    // window, not per cell: the tests lower it)
    const char *pm = getenv("BNPC_SWEEP_PAR_MIN");
    int64_t par_min = pm ? atol(pm) : 2048;
    if (par_min < 2) par_min = 2;
    int T = (int)st->threads;
    if (T > ParScan::MAXT) T = ParScan::MAXT;
    // (the ranks the team really has: every rank of the scan has a range)
    if (T >= 2 && st->n_active >= par_min) T = bnpc_team_ranks(T);
    if (T < 2 || st->n_active < par_min)
        return sweep_window(st, rng, perm, ll, post_new, crp_prior,
                            assignment, col_of_id, col_id, col_size, order,
                            scratch, cpr, nullptr, par_min);

    ParScan par;
    par.T = T;
    par.cpr = cpr;
    par.order = order;
    par.post = scratch;
    par.buf = scratch + st->ld + 1;
    int rc = 0;
    // the error text is thread-local: rank 0 is the calling thread
    bnpc_team_run(T, [&](int rank) {
        if (rank == 0) {
            rc = sweep_window(st, rng, perm, ll, post_new, crp_prior,
                              assignment, col_of_id, col_id, col_size, order,
                              scratch, cpr, &par, par_min);
            par.finish();
        } else {
            par.serve(rank);
        }
    });
    return rc;
}

// ---------------------------------------------------------------------------
// restricted Gibbs 2-way scans (CRP.py:609-632 and :800-820)
// ---------------------------------------------------------------------------
// What the NEXT bnpc_rg_scan (mode 0) of this thread calls right after its
// visiting order is drawn (bnpc_internal.h); consumed by that call.
static thread_local const std::function<void()> *g_scan_order_hook = nullptr;

void bnpc_rg_scan_order_hook(const std::function<void()> *hook)
{
    g_scan_order_hook = hook;
}

extern "C" int bnpc_rg_scan(bnpc_mt19937 *rng, int mode, int64_t S,
                            const double *ll, double DP_a,
                            int64_t *rg_assignment, const int64_t *target,
                            double *log_prob)
{
    if ((mode == 0 && !rng) || (S > 0 && (!ll || !rg_assignment)) ||
        (mode == 1 && S > 0 && !target) || (mode == 1 && !log_prob)) {
        bnpc_set_error("bad argument: NULL");
        return 2;
    }
    const int64_t n = S + 2;
    const double lden = log((double)(n - 1) + DP_a);
    std::vector<int64_t> perm(S);
    std::vector<double> prob(S, 0.0);
    if (mode == 0) {
        bnpc_mt_permutation(rng, S, perm.data());
        // (from here the scan takes exactly one uniform per cell)
        const std::function<void()> *hook = g_scan_order_hook;
        g_scan_order_hook = nullptr;
        if (hook) (*hook)();
    } else {
        for (int64_t s = 0; s < S; s++) perm[s] = s;
    }
    int64_t ones = 0;
    for (int64_t s = 0; s < S; s++) ones += (rg_assignment[s] == 1);
    const bool shortcuts = loop_shortcuts();
    // log(k) of the cluster sizes that can occur: the same libm call, made
    // once per size instead of twice per cell
    static thread_local std::vector<double> log_int;
    if ((int64_t)log_int.size() < n + 1) {
        const int64_t from = (int64_t)log_int.size();
        log_int.resize((size_t)n + 1);
        for (int64_t k = from; k <= n; k++) log_int[k] = log((double)k);
    }

    for (int64_t t = 0; t < S; t++) {
        const int64_t cell = perm[t];
        // take the cell out: n_j = sum(rg) + 2 with rg[cell] = -1
        ones -= (rg_assignment[cell] == 1);
        const int64_t n_j = ones + 1;
        const int64_t n_i = n - n_j - 1;
        const double p0 = ll[2 * cell] + (log_int[n_i] - lden);
        const double p1 = ll[2 * cell + 1] + (log_int[n_j] - lden);
        if (!(p0 == p0) || !(p1 == p1) || (p0 == -INFINITY && p1 == -INFINITY)
            || p0 == INFINITY || p1 == INFINITY) {
            // (the reference's 2-entry fallback, CRP.py:110-114, is for
            // overflow in exp, which max-shifting cannot produce)
            bnpc_set_error("non-finite log posterior in the restricted scan "
                           "(cell %lld: %g, %g)", (long long)cell, p0, p1);
            return 4;
        }
        // _normalize_log for two entries; first maximum wins ties.  Where
        // the smaller entry is more than 40 below the larger one (most cells
        // once the two clusters have parted), x = exp(small - big) < 2^-54
        // and the remaining calls are known without being made: log1p(x)
        // returns x, the larger probability exp(-x) is 1.0, the smaller one
        // stays below 2^-57, so the cumulative sums are (1, 1) or (tiny, 1)
        // and every uniform draw but an exact 0.0 picks the larger entry.
        const int big = p1 > p0 ? 1 : 0;
        const double d = big ? p0 - p1 : p1 - p0;
        if (mode == 0 && !log_prob && shortcuts && d >= -40.0) {
            // An unscored scan needs the pick, not the log-probabilities
            const double u = mt_double(rng);
            int pick = two_way_pick_quick(d, big, u);
            if (pick < 0) {     // in doubt: the full arithmetic, same uniform
                double l0, l1;
                pick = two_way_pick_full(d, big, u, &l0, &l1);
            }
            rg_assignment[cell] = pick;
            ones += (pick == 1);
            continue;
        }
        const bool far = shortcuts && d < -40.0;
        const double z = far ? exp(d) : log1p(exp(d));
        double l0, l1;
        if (big) {
            l0 = p0 - p1 - z;
            l1 = p1 - p1 - z;
        } else {
            l0 = p0 - p0 - z;
            l1 = p1 - p0 - z;
        }
        int64_t pick;
        if (mode == 0) {
            // np.random.choice([0, 1], p=np.exp(log_probs))
            const double u = mt_double(rng);
            if (far && u != 0.0) {
                pick = big;
            } else {
                const double e0 = exp(l0), e1 = exp(l1);
                const double c0 = e0, c1 = e0 + e1;
                pick = (c0 / c1 > u) ? 0 : 1;
                if (!(c1 / c1 > u)) pick = 1;
            }
        } else {
            pick = target[cell] ? 1 : 0;
        }
        rg_assignment[cell] = pick;
        ones += (pick == 1);
        prob[cell] = pick ? l1 : l0;
    }
    if (log_prob) {
        double sum = 0.0;
        for (int64_t s = 0; s < S; s++) sum += prob[s];
        *log_prob = sum;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// One intermediate restricted-Gibbs scan of a split/merge move as ONE call
// (CRP._rg_scan_split followed by CRP._rg_scan_merge, libs/CRP.py:570-606,
// as called from run_rg_nc :535-537): the log-likelihoods of the move's cells
// under the two launch clusters (device), the sequential 2-way assignment
// scan (here, on the stream), the column counts of the two launch clusters for
// the NEW assignment (device; the merged cluster's are their sum), and the MH
// update of the three parameter rows (bnpc_mh_batch).  Nothing here is new
// arithmetic - it is the four calls the binding used to make one by one,
// without the interpreter in between.
// ---------------------------------------------------------------------------
extern "C" int bnpc_rg_scan_step(bnpc_ctx *ctx, const bnpc_host_kernels *k,
                                 bnpc_mt19937 *rng, int view, int64_t n,
                                 int64_t *rg_assignment, double DP_a,
                                 const bnpc_mh_args *mh, int32_t *n1,
                                 int32_t *n0, double *scan_log_prob,
                                 int *status)
{
    return bnpc_rg_scan_step_with(ctx, k, rng, view, n, rg_assignment, DP_a,
                                  mh, n1, n0, scan_log_prob, status, nullptr);
}

int bnpc_rg_scan_step_with(bnpc_ctx *ctx, const bnpc_host_kernels *k,
                           bnpc_mt19937 *rng, int view, int64_t n,
                           int64_t *rg_assignment, double DP_a,
                           const bnpc_mh_args *mh, int32_t *n1, int32_t *n0,
                           double *scan_log_prob, int *status,
                           const double *ll_ready)
{
    if (!ctx || !k || !rng || !rg_assignment || !mh || !n1 || !n0 || !status
        || n < 3 || (mh->G != 3 && mh->G != 2) || mh->n1 != n1
        || mh->n0 != n0) {
        bnpc_set_error("bad argument: rg_scan_step");
        return 2;
    }
    *status = 0;
    const int64_t S = n - 2;
    static thread_local std::vector<double> ll_own;
    static thread_local std::vector<int64_t> labels;
    labels.resize((size_t)n);

    // rows 0 and 1 of the parameter block are the launch clusters
    int rc = 0;
    const double *ll_at = ll_ready;
    if (!ll_at) {
        ll_own.resize((size_t)n * 2);
        rc = bnpc_ll_theta(ctx, view, mh->old_theta, 2, mh->FP, mh->FN,
                           ll_own.data(), 0);
        if (rc) return rc;
        ll_at = ll_own.data();
    }
    // (an unscored scan: no log-probabilities, the loop may take its picks
    // from one exp() per cell)
    double log_prob = 0.0;
    // The draws of this scan's parameter batch depend on nothing but the
    // position of the stream, and once the visiting order is drawn exactly S
    // uniforms lie between here and there: a walker takes them under the
    // scan's loop (and the counts launch); the fused batch adopts them iff
    // the live stream arrives where the walker's stood.  (A scored scan's
    // batch draws into the caller's arrays on the host: no walker.)
    const std::function<void()> at_order = [&]() {
        bnpc_mh_ahead_scan(ctx, rng, S, mh->G, mh->M, mh->n_sd);
    };
    if (!mh->trans_prob) bnpc_rg_scan_order_hook(&at_order);
    rc = bnpc_rg_scan(rng, 0, S, ll_at + 2, DP_a, rg_assignment, nullptr,
                      mh->trans_prob ? &log_prob : nullptr);
    bnpc_rg_scan_order_hook(nullptr);
    if (rc) return rc;

    // anchors: first slot -> cluster i, last slot -> cluster j
    labels[0] = 0;
    for (int64_t s = 0; s < S; s++) labels[s + 1] = rg_assignment[s] ? 1 : 0;
    labels[n - 1] = 1;
    if (scan_log_prob) *scan_log_prob = log_prob;
    // counts of the two launch clusters (the merged cluster's are their sum)
    // and the parameter batch: the device counts, then screens the rows of an
    // unscored scan against those counts, one wait for both; the host
    // evaluates what the screen leaves
    return bnpc_rg_counts_and_batch(ctx, k, rng, view, labels.data(), mh, n1,
                                    n0, status);
}
