// A whole split / merge move of the sampler as ONE call
// (CRP.do_split_move / do_merge_move, libs/CRP.py:434-524, with run_rg_nc
// :527-544, the restricted-Gibbs scans :547-638, the two acceptance tests
// :641-665 and their four ratios :668-820).
//
// Nothing here is new arithmetic: it is the sequence of calls the binding
// (bnpc_amd/model.py) makes one by one - bnpc_view_set, bnpc_ll_tables,
// bnpc_view_counts, bnpc_mt_beta_theta, bnpc_rg_scan_step, bnpc_log_accept,
// bnpc_mh_batch, bnpc_ll_theta, bnpc_rg_scan, bnpc_beta_logpdf_f32 - with the
// NumPy expressions in between (the proposal's np.random.choice calls, the
// element tables, np.sum's pairwise order, the scalar logs) restated on
// NumPy's own log loop and SciPy's gammaln, without the interpreter.
//
// Whatever this file does not model - moves of at most 2 cells, a parameter
// batch with an element the kernel table leaves to SciPy - is handed back:
// the stream and the cached Gaussian are put back where they were when the
// call started, nothing else has been modified, *status = 1, and the binding
// runs the move through its own step-by-step path.
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <algorithm>
#include <cstring>
#include <vector>

#include "bnpc_internal.h"

namespace {

struct Scratch {
    std::vector<int64_t> cells, others, rg, labels, perm, target;
    std::vector<double> probs, cdf, work, L, ll, ll_first, tmp, U, u, A, std2;
    std::vector<int32_t> n1, n0, sd_idx;
    std::vector<float> rows, fresh, gather;
};

struct Restore {
    bnpc_mt19937 *rng;
    bnpc_legacy_gauss *g;
    bnpc_mt19937 rng0;
    bnpc_legacy_gauss g0;
    bool armed;
    Restore(bnpc_mt19937 *r, bnpc_legacy_gauss *gg)
        : rng(r), g(gg), rng0(*r), g0(*gg), armed(true) {}
    ~Restore()
    {
        if (armed) {
            *rng = rng0;
            *g = g0;
        }
    }
};

void gather_cells(const int64_t *assignment, int64_t N, int64_t cl,
                  std::vector<int64_t> &out)
{
    for (int64_t c = 0; c < N; c++)
        if (assignment[c] == cl) out.push_back(c);
}

// the proposal of a split (libs/CRP.py:434-457): false = nothing to split
// natively.  cells = [i, S..., j]; size_data[0] = ltrans, others = the sizes
// of the other clusters
bool propose_split(const bnpc_host_kernels *k, bnpc_mt19937 *rng,
                   const bnpc_move_state *st, Scratch &s, int64_t *pos_out,
                   double *ltrans)
{
    const int64_t K = st->K;
    int64_t tot = 0, largest = 0;
    for (int64_t i = 0; i < K; i++) {
        if (st->sizes[i] < 1) return false;
        tot += st->sizes[i];
        if (st->sizes[i] > largest) largest = st->sizes[i];
    }
    // nothing but singletons: the reference's loop (libs/CRP.py:441-447)
    // never ends; that is left to the caller, where it can be interrupted
    if (largest < 2) return false;
    s.probs.resize((size_t)K);
    for (int64_t i = 0; i < K; i++)
        s.probs[i] = (double)st->sizes[i] / (double)tot;
    int64_t pos;
    for (;;) {
        pos = np_choice_p(s.probs.data(), K, s.cdf, mt_double(rng));
        if (pos >= K) return false;
        s.cells.clear();
        gather_cells(st->assignment, st->N, st->ids[pos], s.cells);
        if (s.cells.size() != 1) break;
    }
    const int64_t n = (int64_t)s.cells.size();
    if (n < 2 || n != st->sizes[pos]) return false;
    // np.random.choice(n, size=2, replace=False) = permutation(n)[:2]
    s.perm.resize((size_t)n);
    mt_fill_permutation(rng, n, s.perm.data());
    std::swap(s.cells[0], s.cells[(size_t)s.perm[0]]);
    std::swap(s.cells[(size_t)n - 1], s.cells[(size_t)s.perm[1]]);
    const double size = (double)st->sizes[pos];
    *ltrans = np_log1(k, s.probs[pos]) - np_log1(k, size)
              - np_log1(k, size - 1.0);
    s.others.clear();
    for (int64_t i = 0; i < K; i++)
        if (i != pos) s.others.push_back(st->sizes[i]);
    *pos_out = pos;
    return true;
}

// the proposal of a merge (libs/CRP.py:484-510)
bool propose_merge(const bnpc_host_kernels *k, bnpc_mt19937 *rng,
                   const bnpc_move_state *st, Scratch &s, int64_t *pos_i,
                   int64_t *pos_j, int64_t *n_i, double *size_data)
{
    const int64_t K = st->K;
    if (K < 2) return false;
    s.work.resize((size_t)K);
    for (int64_t i = 0; i < K; i++) {
        if (st->sizes[i] < 1) return false;
        s.work[i] = 1.0 / (double)st->sizes[i];
    }
    const double tot = np_sum(s.work.data(), K);
    s.probs.resize((size_t)K);
    for (int64_t i = 0; i < K; i++) s.probs[i] = s.work[i] / tot;
    // np.random.choice(ids, p=probs, size=2, replace=False): two uniforms,
    // both looked up in the same cdf; a repeated pick is redrawn with the
    // first one's probability zeroed
    const double u0 = mt_double(rng), u1 = mt_double(rng);
    int64_t a = np_choice_p(s.probs.data(), K, s.cdf, u0);
    int64_t b = std::upper_bound(s.cdf.begin(), s.cdf.end(), u1)
                - s.cdf.begin();
    if (a >= K || b >= K) return false;
    while (a == b) {
        std::vector<double> p(s.probs);
        p[(size_t)a] = 0.0;
        b = np_choice_p(p.data(), K, s.cdf, mt_double(rng));
        if (b >= K) return false;
    }
    s.cells.clear();
    gather_cells(st->assignment, st->N, st->ids[a], s.cells);
    const int64_t ni = (int64_t)s.cells.size();
    if (ni < 1 || ni != st->sizes[a]) return false;
    // np.random.choice(n) = legacy randint(0, n)
    const int64_t ai = (int64_t)mt_interval(rng, (uint64_t)(ni - 1));
    std::swap(s.cells[0], s.cells[(size_t)ai]);
    gather_cells(st->assignment, st->N, st->ids[b], s.cells);
    const int64_t nj = (int64_t)s.cells.size() - ni;
    if (nj < 1 || nj != st->sizes[b]) return false;
    const int64_t aj = (int64_t)mt_interval(rng, (uint64_t)(nj - 1));
    std::swap(s.cells[(size_t)(ni + nj - 1)], s.cells[(size_t)(ni + aj)]);
    const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
    *size_data = (np_log1(k, s.probs[lo]) + np_log1(k, s.probs[hi]))
                 - (np_log1(k, (double)st->sizes[lo])
                    + np_log1(k, (double)st->sizes[hi]));
    *pos_i = a;
    *pos_j = b;
    *n_i = ni;
    return true;
}

// CRP._tables for a float32 profile row: L1 / L0 into out[0..M) / out[M..2M)
void tables_f32(const bnpc_host_kernels *k, const float *theta, int64_t M,
                double FP, double FN, double *out, std::vector<double> &tmp)
{
    tmp.resize((size_t)2 * M);
    const double one_fn = 1 - FN, one_fp = 1 - FP;
    for (int64_t m = 0; m < M; m++) {
        const double t = (double)theta[m];
        const double om = (double)(1.0f - theta[m]);
        tmp[m] = t * one_fn + om * FP;
        tmp[M + m] = t * FN + om * one_fp;
    }
    np_loop(k->np_log, k->np_log_data, tmp.data(), out, 2 * M);
}

// CRP._subset_ll(theta, counts).sum(): np.sum of n1 * L1 + n0 * L0
double subset_ll_sum(const bnpc_host_kernels *k, const float *theta,
                     const int32_t *n1, const int32_t *n0, const int32_t *n1b,
                     const int32_t *n0b, int64_t M, double FP, double FN,
                     Scratch &s)
{
    s.L.resize((size_t)2 * M);
    tables_f32(k, theta, M, FP, FN, s.L.data(), s.tmp);
    s.work.resize((size_t)M);
    for (int64_t m = 0; m < M; m++) {
        const double c1 = (double)(n1[m] + (n1b ? n1b[m] : 0));
        const double c0 = (double)(n0[m] + (n0b ? n0b[m] : 0));
        s.work[m] = c1 * s.L[m] + c0 * s.L[M + m];
    }
    return np_sum(s.work.data(), M);
}

}   // namespace

// Checker hooks (tests/test_native_sweeps.py): np.sum of a float64 vector,
// and the proposals alone on the caller's stream
extern "C" int bnpc_np_sum(const double *a, int64_t n, double *out)
{
    if ((!a && n > 0) || !out || n < 0) {
        bnpc_set_error("bad argument: np_sum");
        return 2;
    }
    *out = np_sum(a, n);
    return 0;
}

extern "C" int bnpc_move_propose(const bnpc_host_kernels *k, bnpc_mt19937 *rng,
                                 const bnpc_move_state *st, int64_t *cells,
                                 int64_t *n_cells, int64_t *n_first,
                                 int64_t *picked, double *size_data,
                                 int64_t *others, int *status)
{
    if (!k || !rng || !st || !cells || !n_cells || !n_first || !picked
        || !size_data || !others || !status) {
        bnpc_set_error("bad argument: move_propose");
        return 2;
    }
    static thread_local Scratch s;
    *status = 0;
    bool ok;
    picked[0] = picked[1] = -1;
    *n_first = 0;
    if (st->move == 0) {
        ok = propose_split(k, rng, st, s, &picked[0], size_data);
        if (ok) std::copy(s.others.begin(), s.others.end(), others);
    } else {
        ok = propose_merge(k, rng, st, s, &picked[0], &picked[1], n_first,
                           size_data);
    }
    if (!ok) {
        *status = 1;
        return 0;
    }
    *n_cells = (int64_t)s.cells.size();
    std::copy(s.cells.begin(), s.cells.end(), cells);
    return 0;
}

// What the NEXT bnpc_sm_move of this thread calls right after its last draw
// of variable length (bnpc_internal.h); consumed by that call.
static thread_local const std::function<void(int)> *g_last_draw_hook = nullptr;

void bnpc_move_last_draw_hook(const std::function<void(int)> *hook)
{
    g_last_draw_hook = hook;
}

extern "C" int bnpc_sm_move(bnpc_ctx *ctx, const bnpc_host_kernels *k,
                            bnpc_mt19937 *rng, bnpc_move_state *st,
                            int *status)
{
    if (!ctx || !k || !rng || !st || !status || !st->gauss || !k->gammaln
        || !st->ids || !st->sizes || !st->assignment || !st->parameters
        || !st->sd || st->K < 1 || st->M < 1 || st->n_sd < 1
        || st->param_stride < st->M || st->view < 1) {
        bnpc_set_error("bad argument: sm_move");
        return 2;
    }
    *status = 1;
    st->accepted = 0;
    const std::function<void(int)> *last_draw = g_last_draw_hook;
    g_last_draw_hook = nullptr;
    bnpc_legacy_gauss *gauss = (bnpc_legacy_gauss *)st->gauss;
    Restore restore(rng, gauss);    // undone below when the move completes
    static thread_local Scratch s;
    const int64_t M = st->M, N = st->N;
    const double FP = st->FP, FN = st->FN;
    const bool split = st->move == 0;
    // BNPC_TIMING=move: where a move's time goes, on stderr
    static const bool trace = [] {
        const char *e = getenv("BNPC_TIMING");
        return e && strstr(e, "move");
    }();
    timespec tr0;
    double tr_at[12];
    int tr_n = 0;
    if (trace) clock_gettime(CLOCK_MONOTONIC, &tr0);
    auto mark = [&]() {
        if (!trace || tr_n >= 12) return;
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        tr_at[tr_n++] = (t.tv_sec - tr0.tv_sec) * 1e6
            + (t.tv_nsec - tr0.tv_nsec) / 1e3;
    };
    auto report = [&](const char *what) {
        if (!trace) return;
        mark();
        fprintf(stderr, "[move] %s of %lld cells, M=%lld:", what,
                (long long)st->n_cells, (long long)st->M);
        static const char *names[] = {"proposal", "view + launch sums",
            "counts", "3 Beta rows", "scans", "scored scan / batch",
            "reverse proposal", "priors", "likelihood ratio", "end"};
        for (int i = 0; i < tr_n; i++)
            fprintf(stderr, " %s %.1f", i < 10 ? names[i] : "?",
                    tr_at[i] - (i ? tr_at[i - 1] : 0.0));
        fprintf(stderr, " us\n");
    };

    // ---- the proposal ----------------------------------------------------
    int64_t pos_i = -1, pos_j = -1, n_first = 0;
    double size_data = 0.0;
    if (split) {
        if (!propose_split(k, rng, st, s, &pos_i, &size_data)) return 0;
    } else {
        if (!propose_merge(k, rng, st, s, &pos_i, &pos_j, &n_first,
                           &size_data))
            return 0;
    }
    const int64_t n = (int64_t)s.cells.size(), S = n - 2;
    // (a move of the two anchors alone has no scan to make: the binding's
    // path.  Moves of 3 and 4 cells - tiny clusters come and go in a long
    // chain: 1.5 % of config 3's steps - were left to the binding too until
    // round 6: 1.3 ms each by its methods against 0.5 here)
    if (n <= 2) return 0;
    const int64_t *cells = s.cells.data();
    st->n_cells = n;
    mark();

    // ---- run_rg_nc: the launch state (libs/CRP.py:527-567) -----------------
    int rc = bnpc_view_set(ctx, st->view, cells, n);
    if (rc) return rc;
    // element tables of the two anchors' own rows (missing entries hold the
    // prior's mean): three possible values per mutation
    {
        int W = 0;
        int64_t Mr = 0;
        const unsigned long long *row[2] = {
            bnpc_ctx_row(ctx, cells[0], &Mr, &W),
            bnpc_ctx_row(ctx, cells[n - 1], &Mr, &W)};
        if (!row[0] || !row[1] || Mr != M) return 0;
        const double vals[3] = {1.0, 0.0, st->fill};
        double arg[6], lg[6];
        for (int v = 0; v < 3; v++) {
            const double t = vals[v], om = 1 - vals[v];
            arg[v] = t * (1 - FN) + om * FP;
            arg[3 + v] = t * FN + om * (1 - FP);
        }
        np_loop(k->np_log, k->np_log_data, arg, lg, 6);
        s.L.resize((size_t)4 * M);      // L1 (2 x M), then L0 (2 x M)
        for (int r = 0; r < 2; r++) {
            for (int64_t m = 0; m < M; m++) {
                const unsigned long long bit = 1ull << (m & 63);
                const int v = (row[r][2 * (m >> 6)] & bit) ? 0
                    : (row[r][2 * (m >> 6) + 1] & bit) ? 1 : 2;
                s.L[(size_t)r * M + m] = lg[v];
                s.L[(size_t)(2 + r) * M + m] = lg[3 + v];
            }
        }
    }
    s.ll.resize((size_t)n * 2);
    rc = bnpc_ll_tables(ctx, st->view, s.L.data(), s.L.data() + 2 * M, 2,
                        s.ll.data(), 0);
    if (rc) return rc;
    s.rg.resize((size_t)S);
    s.labels.resize((size_t)n);
    for (int64_t t = 0; t < S; t++)
        s.rg[t] = s.ll[2 * (t + 1) + 1] > s.ll[2 * (t + 1)] ? 1 : 0;
    auto set_labels = [&]() {
        s.labels[0] = 0;
        for (int64_t t = 0; t < S; t++) s.labels[t + 1] = s.rg[t] ? 1 : 0;
        s.labels[n - 1] = 1;
    };
    set_labels();
    mark();
    s.n1.resize((size_t)3 * M);
    s.n0.resize((size_t)3 * M);
    int32_t *n1 = s.n1.data(), *n0 = s.n0.data();
    rc = bnpc_view_counts(ctx, st->view, s.labels.data(), 2, n1, n0);
    if (rc) return rc;
    auto sum_rows = [&]() {
        for (int64_t m = 0; m < M; m++) {
            n1[2 * M + m] = n1[m] + n1[M + m];
            n0[2 * M + m] = n0[m] + n0[M + m];
        }
    };
    sum_rows();
    mark();
    s.rows.resize((size_t)3 * M);
    s.fresh.resize((size_t)3 * M);
    float *rows = s.rows.data(), *fresh = s.fresh.data();
    // The next call is a restricted scan over the two launch rows (an
    // intermediate one, or the scored scan of a split): its sums are queued
    // as soon as those two rows exist and run under the draws of the third
    // (the merged cluster's: ~M gamma pairs on the host).
    const bool ahead = st->scan_no > 0 || split;
    bool begun = false;
    for (int g = 0; g < 3; g++) {
        if (g == 2 && ahead) {
            s.ll_first.resize((size_t)n * 2);
            rc = bnpc_ll_theta_begin(ctx, st->view, rows, 2, FP, FN,
                                     s.ll_first.data(), 0);
            if (rc) return rc;
            begun = true;
        }
        rc = bnpc_mt_beta_theta(rng, gauss, M, st->p, st->q, n1 + g * M,
                                n0 + g * M, 1.0, st->tmin, st->tmax,
                                rows + g * M);
        if (rc) {
            if (begun) (void)bnpc_ll_theta_end(ctx);
            return rc;
        }
    }
    if (begun) {
        rc = bnpc_ll_theta_end(ctx);
        if (rc) return rc;
    }
    const double *ll_first = begun ? s.ll_first.data() : nullptr;
    mark();

    // ---- the intermediate scans (libs/CRP.py:535-537) ----------------------
    s.sd_idx.resize((size_t)3 * M);
    s.U.resize((size_t)3 * M);
    s.u.resize((size_t)3 * M);
    s.A.resize((size_t)3 * M);
    double log_prob[3] = {0, 0, 0};
    int64_t declined[3];
    bnpc_mh_args mh;
    memset(&mh, 0, sizeof mh);
    mh.M = M;
    mh.n1 = n1;
    mh.n0 = n0;
    mh.sd = st->sd;
    mh.n_sd = st->n_sd;
    mh.tmin = st->tmin;
    mh.tmax = st->tmax;
    mh.FP = FP;
    mh.FN = FN;
    mh.p = st->p;
    mh.q = st->q;
    mh.uniform_prior = st->uniform_prior;
    mh.sd_idx = s.sd_idx.data();
    mh.U = s.U.data();
    mh.u = s.u.data();
    mh.A = s.A.data();
    mh.log_prob = log_prob;
    mh.declined = declined;
    mh.threads = st->threads;
    int sub = 0;
    double scan_prob = 0.0;
    for (int it = 0; it < st->scan_no; it++) {
        mh.G = 3;
        mh.trans_prob = 0;
        mh.old_theta = rows;
        mh.new_theta = fresh;
        rc = bnpc_rg_scan_step_with(ctx, k, rng, st->view, n, s.rg.data(),
                                    st->DP_a, &mh, n1, n0, &scan_prob, &sub,
                                    ll_first);
        ll_first = nullptr;
        if (rc) return rc;
        if (sub) return 0;
        std::swap(rows, fresh);
    }

    mark();
    const double log_a = np_log1(k, st->DP_a);
    double A;
    auto rg_ones = [&]() {
        int64_t ones = 0;
        for (int64_t t = 0; t < S; t++) ones += s.rg[t];
        return ones;
    };
    s.gather.resize((size_t)2 * M);
    if (split) {
        // ---- _do_rg_split_MH (libs/CRP.py:641-653) -------------------------
        mh.G = 2;
        mh.trans_prob = 1;
        mh.old_theta = rows;
        mh.new_theta = fresh;
        // (the merged cluster's row is not part of the scored scan)
        memcpy(fresh + 2 * M, rows + 2 * M, (size_t)M * sizeof(float));
        rc = bnpc_rg_scan_step_with(ctx, k, rng, st->view, n, s.rg.data(),
                                    st->DP_a, &mh, n1, n0, &scan_prob, &sub,
                                    ll_first);
        ll_first = nullptr;
        if (rc) return rc;
        if (sub) return 0;
        std::swap(rows, fresh);
        sum_rows();
        mark();
        const double gs_split = scan_prob + (0.0 + log_prob[0] + log_prob[1]);
        // np.random.choice(sd, size=M); the reverse move's parameter
        // proposal: the merged launch row -> the cluster's own row
        mt_fill_interval32(rng, (uint32_t)(st->n_sd - 1), s.sd_idx.data(), M);
        // (all that the stream still gives inside this move: the uniform of
        // its acceptance test, unless the scan left a side empty)
        if (last_draw) {
            const int64_t ones_now = rg_ones();
            (*last_draw)(ones_now != 0 && ones_now != S ? 1 : 0);
        }
        s.std2.resize((size_t)2 * M);
        for (int64_t m = 0; m < M; m++) s.std2[m] = st->sd[s.sd_idx[m]];
        const int64_t cl = st->ids[pos_i];
        const float *own = st->parameters + cl * st->param_stride;
        double gs_merge = 0.0;
        bnpc_accept_args la;
        memset(&la, 0, sizeof la);
        la.G = 1;
        la.M = M;
        la.new_theta = own;
        la.old_theta = rows + 2 * M;
        la.std = s.std2.data();
        la.n1 = n1 + 2 * M;
        la.n0 = n0 + 2 * M;
        la.fmin = st->tmin;
        la.fmax = st->tmax;
        la.tmin = st->tmin;
        la.tmax = st->tmax;
        la.FP = FP;
        la.FN = FN;
        la.p = st->p;
        la.q = st->q;
        la.uniform_prior = st->uniform_prior;
        la.clip = 1;
        la.A = s.A.data();
        la.sum = &gs_merge;
        la.threads = st->threads_wide;
        rc = bnpc_log_accept(k, &la, &sub);
        if (rc) return rc;
        if (sub) return 0;
        const double trans = gs_merge - gs_split;
        mark();

        // _get_lprior_ratio_split (libs/CRP.py:695-713)
        const int64_t n_j = rg_ones() + 1, n_i = n - n_j;
        double lprior = log_a - k->gammaln((double)n, 0);
        if (n_i > 0) lprior += k->gammaln((double)n_j, 0);
        if (n_j > 0) lprior += k->gammaln((double)n_i, 0);
        if (!st->uniform_prior) {
            double sum_split = 0.0, sum_own = 0.0;
            s.work.resize((size_t)2 * M);
            rc = bnpc_beta_logpdf_f32(k, rows, 2 * M, st->p, st->q, nullptr,
                                      nullptr, s.work.data(), &sum_split,
                                      st->threads_wide);
            if (rc) return rc;
            rc = bnpc_beta_logpdf_f32(k, own, M, st->p, st->q, nullptr,
                                      nullptr, s.work.data(), &sum_own,
                                      st->threads_wide);
            if (rc) return rc;
            lprior += sum_split - sum_own;
        }
        // _get_ll_ratio (libs/CRP.py:716-733)
        mark();
        const double ll_i = subset_ll_sum(k, rows, n1, n0, nullptr, nullptr,
                                          M, FP, FN, s);
        const double ll_j = subset_ll_sum(k, rows + M, n1 + M, n0 + M,
                                          nullptr, nullptr, M, FP, FN, s);
        const double ll_all = subset_ll_sum(k, rows + 2 * M, n1, n0, n1 + M,
                                            n0 + M, M, FP, FN, s);
        const double llr = ll_i + ll_j - ll_all;
        mark();
        // _get_ltrans_prob_size_ratio_split (libs/CRP.py:757-764)
        double norm = 0.0;
        {
            bool first = true;
            auto add = [&](int64_t size) {
                const double v = 1.0 / (double)size;
                norm = first ? v : norm + v;
                first = false;
            };
            for (int64_t o : s.others) add(o);
            add(n_i);
            add(n_j);
        }
        const double rev = np_log1(k, 1.0 / (double)n_i / norm)
                           + np_log1(k, 1.0 / (double)n_j / norm);
        const double size_ratio = rev - size_data;
        A = trans + lprior + llr + size_ratio;
        const int64_t ones = rg_ones();
        bool accept = false;
        if (ones != 0 && ones != S)
            accept = np_log1(k, mt_double(rng)) < A;
        if (accept) {
            // the smallest unused id (libs/CRP.py:297-299)
            std::vector<char> used((size_t)st->K + 1, 0);
            for (int64_t i = 0; i < st->K; i++)
                if (st->ids[i] >= 0 && st->ids[i] <= st->K)
                    used[(size_t)st->ids[i]] = 1;
            int64_t new_cl = 0;
            while (used[(size_t)new_cl]) new_cl++;
            if (new_cl >= N) return 0;
            memcpy(st->parameters + cl * st->param_stride, rows,
                   (size_t)M * sizeof(float));
            memcpy(st->parameters + new_cl * st->param_stride, rows + M,
                   (size_t)M * sizeof(float));
            int64_t moved = 1;
            for (int64_t t = 0; t < S; t++) {
                if (s.rg[t] == 1) {
                    st->assignment[cells[t + 1]] = new_cl;
                    moved++;
                }
            }
            st->assignment[cells[n - 1]] = new_cl;
            st->accepted = 1;
            st->cl_i = cl;
            st->cl_j = new_cl;
            st->moved = moved;
        }
    } else {
        // ---- _do_rg_merge_MH (libs/CRP.py:656-665) -------------------------
        // the scored update of the merged launch row (libs/CRP.py:581-587)
        const int64_t cl_i = st->ids[pos_i], cl_j = st->ids[pos_j];
        double gs_merge = 0.0;
        {
            bnpc_mh_args one = mh;
            one.G = 1;
            one.trans_prob = 1;
            one.old_theta = rows + 2 * M;
            one.new_theta = fresh + 2 * M;
            one.n1 = n1 + 2 * M;
            one.n0 = n0 + 2 * M;
            one.log_prob = &gs_merge;
            rc = bnpc_mh_batch(k, rng, &one, &sub);
            if (rc) return rc;
            if (sub) return 0;
        }
        const float *merged = fresh + 2 * M;
        mark();
        // _rg_get_split_prob (libs/CRP.py:777-820)
        mt_fill_interval32(rng, (uint32_t)(st->n_sd - 1), s.sd_idx.data(),
                           2 * M);
        if (last_draw) (*last_draw)(1);     // (the acceptance test's uniform)
        s.std2.resize((size_t)2 * M);
        for (int64_t m = 0; m < 2 * M; m++) s.std2[m] = st->sd[s.sd_idx[m]];
        memcpy(s.gather.data(), st->parameters + cl_i * st->param_stride,
               (size_t)M * sizeof(float));
        memcpy(s.gather.data() + M, st->parameters + cl_j * st->param_stride,
               (size_t)M * sizeof(float));
        double prob2[2] = {0, 0};
        bnpc_accept_args la;
        memset(&la, 0, sizeof la);
        la.G = 2;
        la.M = M;
        la.new_theta = s.gather.data();
        la.old_theta = rows;
        la.std = s.std2.data();
        la.n1 = n1;
        la.n0 = n0;
        la.fmin = 0.0;
        la.fmax = 1.0;
        la.tmin = st->tmin;
        la.tmax = st->tmax;
        la.FP = FP;
        la.FN = FN;
        la.p = st->p;
        la.q = st->q;
        la.uniform_prior = st->uniform_prior;
        la.clip = 1;
        la.A = s.A.data();
        la.sum = prob2;
        la.threads = st->threads_wide;
        rc = bnpc_log_accept(k, &la, &sub);
        if (rc) return rc;
        if (sub) return 0;
        rc = bnpc_ll_theta(ctx, st->view, s.gather.data(), 2, FP, FN,
                           s.ll.data(), 0);
        if (rc) return rc;
        s.target.resize((size_t)S);
        for (int64_t t = 0; t < S; t++)
            s.target[t] = st->assignment[cells[t + 1]] == cl_i ? 0 : 1;
        double prob_assign = 0.0;
        rc = bnpc_rg_scan(nullptr, 1, S, s.ll.data() + 2, st->DP_a,
                          s.rg.data(), s.target.data(), &prob_assign);
        if (rc) return rc;
        const double gs_split = prob2[0] + prob2[1] + prob_assign;
        const double trans = gs_split - gs_merge;
        mark();

        // _get_lprior_ratio_merge (libs/CRP.py:736-754); the launch
        // assignment is now the clusters' own
        const int64_t n_j = rg_ones() + 1, n_i = n - n_j;
        double lprior = k->gammaln((double)n, 0) - log_a;
        if (n_i > 0) lprior -= k->gammaln((double)n_i, 0);
        if (n_j > 0) lprior -= k->gammaln((double)n_j, 0);
        if (!st->uniform_prior) {
            double sum_merged = 0.0, sum_own = 0.0;
            s.work.resize((size_t)2 * M);
            rc = bnpc_beta_logpdf_f32(k, merged, M, st->p, st->q, nullptr,
                                      nullptr, s.work.data(), &sum_merged,
                                      st->threads_wide);
            if (rc) return rc;
            rc = bnpc_beta_logpdf_f32(k, s.gather.data(), 2 * M, st->p, st->q,
                                      nullptr, nullptr, s.work.data(),
                                      &sum_own, st->threads_wide);
            if (rc) return rc;
            lprior += sum_merged - sum_own;
        }
        // _get_ll_ratio: the counts of the clusters' own halves
        mark();
        set_labels();
        rc = bnpc_view_counts(ctx, st->view, s.labels.data(), 2, n1, n0);
        if (rc) return rc;
        const double ll_i = subset_ll_sum(k, rows, n1, n0, nullptr, nullptr,
                                          M, FP, FN, s);
        const double ll_j = subset_ll_sum(k, rows + M, n1 + M, n0 + M,
                                          nullptr, nullptr, M, FP, FN, s);
        const double ll_all = subset_ll_sum(k, merged, n1, n0, n1 + M, n0 + M,
                                            M, FP, FN, s);
        const double llr = ll_all - ll_i - ll_j;
        mark();
        // _get_ltrans_prob_size_ratio_merge (libs/CRP.py:767-774)
        if (S - 1 <= 0) return 0;
        const double rev = -np_log1(k, (double)N) - np_log1(k, (double)(S - 1));
        const double size_ratio = rev - size_data;
        A = trans + lprior + llr + size_ratio;
        if (np_log1(k, mt_double(rng)) < A) {
            memcpy(st->parameters + cl_i * st->param_stride, merged,
                   (size_t)M * sizeof(float));
            for (int64_t t = n_first; t < n; t++)
                st->assignment[cells[t]] = cl_i;
            st->accepted = 1;
            st->cl_i = cl_i;
            st->cl_j = cl_j;
            st->moved = n - n_first;
        }
    }
    st->log_A = A;
    restore.armed = false;
    *status = 0;
    report(split ? "split" : "merge");
    return 0;
}
