// One whole MCMC step as ONE call: Chain_steps.do_step + Chain.update_results
// of the reference (libs/MCMC.py:320-342, 242-282), see include/bnpc_hip.h.
//
// Like bnpc_moves.cpp this file holds no new arithmetic: it is the sequence
// of library calls the binding makes per step (bnpc_amd/mcmc.py: advance,
// TraceStore.put_state; bnpc_amd/model.py: update_assignments_Gibbs,
// _gibbs_window, update_assignments_split_merge, update_DP_alpha,
// update_parameters, update_error_rates / MH_error_rates, get_ll_full,
// get_lprior_full) with the NumPy / SciPy scalar expressions between them
// restated on NumPy's own log loop and SciPy's own special functions (the
// kernel table).  ~150 interpreter-level calls per step were a fifth of a
// converged config-3 step.
//
// A phase this file does not model ends the call BEFORE that phase with the
// stream where the reference has it there (ch->need); the binding runs the
// phase through its methods and calls again (ch->phase).
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <cstring>
#include <vector>

#include "bnpc_internal.h"

namespace {

const double EPSILON = 1e-15;       // np.finfo(np.float64).resolution

struct Memo {
    double key = 0.0, val = 0.0;
    bool set = false;
};

struct Work {
    int64_t N = 0, M = 0;
    // get_lpost_single_new_cluster: the per-cell sums, per (FP, FN)
    std::vector<double> newcl, post_new;
    double newcl_FP = 0.0, newcl_FN = 0.0;
    bool newcl_set = false;
    // the sweep
    std::vector<int64_t> perm, assign, col_of_id, col_id, col_size, order, born;
    std::vector<double> scratch, col_prior, heap_ll, tabs;
    // parameter rows of the live clusters, dict order
    std::vector<float> rows, fresh;
    bool rows_current = false;
    // resident per-cluster counts: the state they were made for
    std::vector<int32_t> n1, n0;
    std::vector<int64_t> lab_ids, lab_assign;
    uint64_t lab_gen = 0;
    bool lab_set = false;
    // Beta prior log-density cache: rows of theta with their densities
    std::vector<int64_t> pc_ids;
    std::vector<float> pc_theta, kt;
    std::vector<double> pc_prior, kp, prior_out, dens;
    bool pc_set = false;
    bool pc_is_rows = false;    // ... and they are w.rows as gathered now
    double pc_sum = 0.0;        // their sum in index order, from the batch
    bool pc_sum_set = false;
    // batch scratch
    std::vector<int32_t> sd_idx;
    std::vector<double> U, u, A, log_prob, cdf;
    std::vector<int64_t> declined, sorted;
    // scalar densities by argument
    Memo alpha_prior, err_prior[2][8];
    int err_next[2] = {0, 0};
};

using Clock = std::chrono::steady_clock;

struct Lap {
    bnpc_chain *ch;
    int slot;
    Clock::time_point t0;
    Lap(bnpc_chain *c, int s) : ch(c), slot(s), t0(Clock::now()) {}
    void stop()
    {
        if (slot < 0) return;
        ch->clock_ns[slot] += std::chrono::duration_cast<
            std::chrono::nanoseconds>(Clock::now() - t0).count();
        ch->clock_calls[slot]++;
        slot = -1;
    }
};

struct Snapshot {
    bnpc_mt19937 rng;
    bnpc_legacy_gauss g;
    Snapshot(const bnpc_mt19937 *r, const bnpc_legacy_gauss *gg)
        : rng(*r), g(*gg) {}
    void put_back(bnpc_mt19937 *r, bnpc_legacy_gauss *gg) const
    {
        *r = rng;
        *gg = g;
    }
};

inline int team_for(const bnpc_chain *ch, int64_t elements)
{
    return elements >= ch->wide_from && ch->threads_wide > ch->threads
        ? ch->threads_wide : ch->threads;
}

void gather_rows(const bnpc_chain *ch, Work &w)
{
    if (w.rows_current) return;
    w.pc_is_rows = false;
    const int64_t M = ch->M;
    w.rows.resize((size_t)ch->K * M);
    for (int64_t g = 0; g < ch->K; g++)
        memcpy(w.rows.data() + (size_t)g * M,
               ch->parameters + (size_t)ch->ids[g] * ch->param_stride,
               (size_t)M * sizeof(float));
    w.rows_current = true;
}

bool counts_current(const bnpc_ctx *ctx, const bnpc_chain *ch, const Work &w)
{
    return w.lab_set && w.lab_gen == bnpc_ctx_label_counts_generation(ctx)
        && (int64_t)w.lab_ids.size() == ch->K
        && memcmp(w.lab_ids.data(), ch->ids, (size_t)ch->K * 8) == 0
        && memcmp(w.lab_assign.data(), ch->assignment, (size_t)ch->N * 8) == 0;
}

void counts_made(const bnpc_ctx *ctx, const bnpc_chain *ch, Work &w)
{
    w.lab_ids.assign(ch->ids, ch->ids + ch->K);
    w.lab_assign.assign(ch->assignment, ch->assignment + ch->N);
    w.lab_gen = bnpc_ctx_label_counts_generation(ctx);
    w.lab_set = true;
}

// CRP._label_counts: the per-cluster column counts of the current state,
// resident on the device (and in w.n1 / w.n0)
int ensure_counts(bnpc_ctx *ctx, const bnpc_chain *ch, Work &w)
{
    if (counts_current(ctx, ch, w)) return 0;
    w.lab_set = false;
    const size_t E = (size_t)ch->K * ch->M;
    w.n1.resize(E);
    w.n0.resize(E);
    const int rc = bnpc_colcounts_by_label(ctx, ch->assignment, ch->ids,
                                           ch->K, w.n1.data(), w.n0.data());
    if (rc) return rc;
    counts_made(ctx, ch, w);
    return 0;
}

// CRP._known_prior: the cache's rows aligned with the live clusters (rows of
// ids the cache does not hold carry NaN parameters, which match nothing)
void known_prior(const bnpc_chain *ch, Work &w, const float **kt,
                 const double **kp)
{
    *kt = nullptr;
    *kp = nullptr;
    if (ch->uniform_prior || !w.pc_set) return;
    const int64_t K = ch->K, M = ch->M;
    if ((int64_t)w.pc_ids.size() == K
        && memcmp(w.pc_ids.data(), ch->ids, (size_t)K * 8) == 0) {
        *kt = w.pc_theta.data();
        *kp = w.pc_prior.data();
        return;
    }
    w.kt.resize((size_t)K * M);
    w.kp.resize((size_t)K * M);
    const float nan32 = nanf("");
    for (int64_t g = 0; g < K; g++) {
        int64_t at = -1;
        for (size_t c = 0; c < w.pc_ids.size(); c++)
            if (w.pc_ids[c] == ch->ids[g]) {
                at = (int64_t)c;
                break;
            }
        float *t = w.kt.data() + (size_t)g * M;
        double *d = w.kp.data() + (size_t)g * M;
        if (at < 0) {
            for (int64_t m = 0; m < M; m++) {
                t[m] = nan32;
                d[m] = 0.0;
            }
        } else {
            memcpy(t, w.pc_theta.data() + (size_t)at * M, (size_t)M * 4);
            memcpy(d, w.pc_prior.data() + (size_t)at * M, (size_t)M * 8);
        }
    }
    *kt = w.kt.data();
    *kp = w.kp.data();
}

// scipy.stats.gamma.logpdf(x, a, loc) with scale 1
// (bnpc_amd/fastdist.py: _gamma_logpdf_direct): xs = (x - loc) / 1,
// _logpdf = xlogy(a - 1, xs) - xs - gammaln(a), minus np.log(1)
bool gamma_logpdf(const bnpc_host_kernels *k, double x, double a, double loc,
                  double *out)
{
    if (!k->gammaln || !k->xlogy) return false;
    const double xs = (x - loc) / 1.0;
    if (!(xs > 0.0)) return false;
    *out = (k->xlogy(a - 1.0, xs, 0) - xs - k->gammaln(a, 0)) - 0.0;
    return true;
}

// CRP_errors_learning._error_prior_logpdf: the truncated-normal prior density
// of an error rate, remembered by argument
bool error_prior(const bnpc_host_kernels *k, const bnpc_chain *ch, Work &w,
                 int which, double x, double *out)
{
    for (Memo &m : w.err_prior[which])
        if (m.set && m.key == x) {
            *out = m.val;
            return true;
        }
    const double *pr = which == 0 ? ch->FP_prior : ch->FN_prior;
    int st = 0;
    double val = 0.0;
    if (bnpc_tn_logpdf_scalar(k, x, pr[0], pr[1], pr[2], pr[3], &val, &st)
        || st)
        return false;
    Memo &m = w.err_prior[which][w.err_next[which]++ & 7];
    m.key = x;
    m.val = val;
    m.set = true;
    *out = val;
    return true;
}

// the draws of CRP.update_DP_alpha (libs/CRP.py:386-410) and the value they
// give - a function of the stream and of (DP_a, N, K, the prior) alone: the
// walker that takes a parameter batch's draws ahead goes through them on its
// copy of the stream
struct AlphaArgs {
    double DP_a, shape, rate;
    int64_t N, K;
};

int alpha_draws(const bnpc_host_kernels *k, bnpc_mt19937 *rng,
                bnpc_legacy_gauss *g, const AlphaArgs &p, double *alpha)
{
    const double kk = (double)p.K;
    double eta = 0.0;
    const double a = p.DP_a + 1, b = (double)p.N;
    int rc = bnpc_mt_beta(rng, g, 1, &a, &b, &eta);
    if (rc) return rc;
    const double rate = p.rate - np_log1(k, eta);
    const double wgt = (p.shape + kk - 1) / ((double)p.N * rate);
    const double pi_eta = wgt / (1 + wgt);
    if (mt_double(rng) < pi_eta)
        *alpha = bnpc_legacy_gamma(rng, g, p.shape + kk, rate);
    else
        *alpha = bnpc_legacy_gamma(rng, g, p.shape + kk - 1, rate);
    return 0;
}

// The draws of this step's parameter batch taken ahead (bnpc_mh_ahead_begin):
// from a point of the step after which the stream's way to the batch is known
// - `doubles` uniforms (a sweep's picks, one per cell, if no cluster is born;
// a move's acceptance test), the alpha test and, if it fires, update_DP_alpha
// with the clusters there are now - a walker goes that way on a copy of the
// stream and draws the rows of K + 1 clusters.  Whatever else happens (a
// birth, a move that changes K before an alpha update) leaves the live stream
// somewhere else, and the batch draws for itself.
void params_ahead(bnpc_ctx *ctx, const bnpc_host_kernels *k,
                  const bnpc_mt19937 *rng, const bnpc_chain *ch,
                  int64_t doubles)
{
    if (ch->fix_assign) return;
    const double dpa_prob = ch->dpa_prob;
    const AlphaArgs args = {ch->DP_a, ch->dpa_shape, ch->dpa_rate, ch->N,
                            ch->K};
    const auto way = [k, doubles, dpa_prob, args](bnpc_mt19937 *r,
                                                  bnpc_legacy_gauss *g) {
        // (2 words per uniform: whole state blocks are skipped untempered)
        for (int64_t left = 2 * doubles; left > 0;) {
            if (r->pos >= 624) mt_refill(r);
            const int64_t take = std::min<int64_t>(624 - r->pos, left);
            r->pos += (int32_t)take;
            left -= take;
        }
        if (mt_double(r) < dpa_prob) {
            double alpha;
            if (alpha_draws(k, r, g, args, &alpha)) return false;
        }
        return true;
    };
    bool posted = false;
    (void)bnpc_mh_ahead_begin(ctx, rng, (const bnpc_legacy_gauss *)ch->gauss,
                              way, ch->K + 1, ch->M, ch->n_sd, &posted);
}

// ---------------------------------------------------------------- the phases
// CRP.update_assignments_Gibbs (bnpc_amd/model.py; libs/CRP.py:254-288) for a
// sweep whose whole matrix is one hinted launch.  *done = false: not this
// kind of sweep - nothing was drawn.
int gibbs_phase(bnpc_ctx *ctx, const bnpc_host_kernels *k, bnpc_mt19937 *rng,
                bnpc_chain *ch, Work &w, bool *done)
{
    *done = false;
    const Clock::time_point tg_in = Clock::now();
    const int64_t N = ch->N, M = ch->M, K = ch->K;
    // (births write their rows into `parameters` M floats apart)
    // (any number of clusters whose whole matrix fits the host budget: the
    // hint record holds its columns as 16-bit numbers)
    // spare columns for the clusters opened during the sweep (running out
    // means a copy of the whole matrix into a wider one): a few for a
    // converged chain, an eighth of the columns when there are hundreds or
    // thousands (a first sweep opens about as many clusters as the data hold)
    const int64_t spare = K <= 64
        ? std::max<int64_t>(4, std::min<int64_t>(16, K / 4))
        : std::min<int64_t>(512, std::max<int64_t>(16, K / 8));
    // (the budget is tested on the matrix as it is allocated, spare columns
    // included - the same expression as CRP.update_assignments_Gibbs)
    if (K < 1 || K + 512 > 32767 || !ch->sweep_hint || ch->param_stride != M
        || ch->sweep_bytes / (8 * (K + spare)) < N)
        return 0;
    const double FP = ch->FP, FN = ch->FN;
    // get_lpost_single_new_cluster (libs/CRP.py:230-234): an m-sequential
    // device sum over constant tables, cached per (FP, FN)
    if (!w.newcl_set || w.newcl_FP != FP || w.newcl_FN != FN) {
        const double c1 = np_log1(k, ch->mix1 * (1 - FN) + ch->mix0 * FP);
        const double c0 = np_log1(k, ch->mix1 * FN + ch->mix0 * (1 - FP));
        w.tabs.resize((size_t)2 * M);
        for (int64_t m = 0; m < M; m++) {
            w.tabs[m] = c1;
            w.tabs[M + m] = c0;
        }
        w.newcl.resize((size_t)N);
        const int rc = bnpc_ll_tables(ctx, 0, w.tabs.data(), w.tabs.data() + M,
                                      1, w.newcl.data(), 0);
        if (rc) return rc;
        w.newcl_FP = FP;
        w.newcl_FN = FN;
        w.newcl_set = true;
    }
    w.post_new.resize((size_t)N);
    const double prior_new = ch->crp_prior[N + 1];
    for (int64_t i = 0; i < N; i++) w.post_new[i] = w.newcl[i] + prior_new;

    int64_t ld = K + spare;
    w.col_prior.resize((size_t)K);
    for (int64_t g = 0; g < K; g++) {
        if (ch->sizes[g] < 1 || ch->sizes[g] > N) {
            bnpc_set_error("bad argument: cluster size out of range");
            return 2;
        }
        w.col_prior[g] = ch->crp_prior[ch->sizes[g]];
    }
    static const bool trace_g = [] {
        const char *e = getenv("BNPC_TIMING");
        return e && strstr(e, "gibbs");
    }();
    const Clock::time_point tg0 = Clock::now();
    gather_rows(ch, w);
    const Clock::time_point tg1 = Clock::now();
    double *ll = nullptr;
    bnpc_top2 *top2 = nullptr;
    // The sums are queued first; the visiting order - the sweep's first draw,
    // which consumes no result of the device - is drawn under them, and the
    // hint kernel, queued behind, makes its records IN that order: the loop
    // below reads them front to back.
    int rc = bnpc_ll_theta_pinned_sums_issue(ctx, 0, w.rows.data(), K, FP, FN,
                                             ld, w.col_prior.data(), &ll);
    if (rc) return rc;
    const Clock::time_point tg2 = Clock::now();
    w.perm.resize((size_t)N);
    mt_fill_permutation(rng, N, w.perm.data());
    rc = bnpc_hints_in_order_issue(ctx, w.perm.data(), &top2);
    if (rc) return rc;
    // from here to the parameter batch the stream gives one uniform per cell
    // (and whatever a birth draws: then the walker's work is dropped)
    if (ch->phase == BNPC_PHASE_ASSIGN) params_ahead(ctx, k, rng, ch, N);
    // under the launches: the sweep's private state
    w.assign.assign(ch->assignment, ch->assignment + N);
    w.col_of_id.assign((size_t)N, -1);
    w.col_id.assign((size_t)ld, -1);
    w.col_size.assign((size_t)ld, 0);
    w.order.assign((size_t)ld, 0);
    for (int64_t g = 0; g < K; g++) {
        const int64_t id = ch->ids[g];
        if (id < 0 || id >= N || w.col_of_id[id] >= 0) {
            (void)bnpc_hints_wait(ctx);
            bnpc_set_error("bad argument: cluster ids");
            return 2;
        }
        w.col_of_id[id] = g;
        w.col_id[g] = id;
        w.col_size[g] = ch->sizes[g];
        w.order[g] = g;
    }
    w.scratch.resize((size_t)2 * (ld + 1));
    w.born.resize((size_t)N);
    const Clock::time_point tg3 = Clock::now();
    {
        // how long the sweep waits for its evaluation (tables, sums, combine,
        // hint: four launches) once the visiting order and its state are
        // ready - the part of the device's work that is NOT hidden
        const Clock::time_point t0 = Clock::now();
        rc = bnpc_hints_wait(ctx);
        ch->clock_ns[9] += std::chrono::duration_cast<
            std::chrono::nanoseconds>(Clock::now() - t0).count();
        ch->clock_calls[9]++;
    }
    if (rc) return rc;
    const Clock::time_point tg4 = Clock::now();

    bnpc_gibbs_state st;
    memset(&st, 0, sizeof st);
    st.n_cells = N;
    st.ld = ld;
    st.n_cols = K;
    st.n_active = K;
    st.pos = 0;
    st.new_cell = -1;
    st.pos_end = N;
    st.row_base = -1;
    st.threads = ch->threads;
    if (top2) {
        st.hint = top2;
        st.hint_prior = w.col_prior.data();
        st.hint_cols = K;
        st.matrix_wait = (int (*)(void *))bnpc_matrix_wait;
        st.matrix_wait_arg = ctx;
        st.hint_in_order = 1;
    }
    st.birth_ctx = ctx;
    st.birth_view = 0;
    st.birth_put = 0;
    st.birth_rows = N;
    st.theta_host = ch->parameters;
    st.beta_p = ch->p;
    st.beta_q = ch->q;
    st.tmin = ch->tmin;
    st.tmax = ch->tmax;
    st.FP = FP;
    st.FN = FN;
    st.gauss = ch->gauss;
    st.born = w.born.data();
    st.born_cap = N;
    bool waited = false;
    for (;;) {
        st.n_born = 0;
        rc = bnpc_gibbs_sweep(&st, rng, w.perm.data(), ll, w.post_new.data(),
                              ch->crp_prior, w.assign.data(),
                              w.col_of_id.data(), w.col_id.data(),
                              w.col_size.data(), w.order.data(),
                              w.scratch.data());
        if (rc) return rc;
        if (st.n_born) w.rows_current = false;
        if (st.new_cell < 0) break;
        // a birth the sweep had no column for: widen the matrix (a copy on
        // the heap: the pinned one cannot grow) and open the cluster here
        if (top2 && !waited) {
            rc = bnpc_matrix_wait(ctx);
            if (rc) return rc;
            waited = true;
        }
        if (st.n_cols == ld) {
            const int64_t wider = ld + std::max<int64_t>(16, ld / 4);
            std::vector<double> grown((size_t)N * wider);
            for (int64_t r = 0; r < N; r++)
                memcpy(grown.data() + (size_t)r * wider, ll + (size_t)r * ld,
                       (size_t)st.n_cols * sizeof(double));
            w.heap_ll.swap(grown);
            ll = w.heap_ll.data();
            w.col_id.resize((size_t)wider, -1);
            w.col_size.resize((size_t)wider, 0);
            w.order.resize((size_t)wider, 0);
            w.scratch.resize((size_t)2 * (wider + 1));
            ld = wider;
            st.ld = ld;
        }
        rc = bnpc_sweep_open_cluster(&st, rng, st.new_cell, ll,
                                     w.assign.data(), w.col_of_id.data(),
                                     w.col_id.data(), w.col_size.data(),
                                     w.order.data());
        if (rc) return rc;
        w.rows_current = false;
        st.new_cell = -1;
    }
    const Clock::time_point tg5 = Clock::now();
    // commit: the live clusters in dict order, the new labels
    for (int64_t a = 0; a < st.n_active; a++) {
        const int64_t col = w.order[a];
        ch->ids[a] = w.col_id[col];
        ch->sizes[a] = w.col_size[col];
    }
    // (the gathered parameter rows stay valid if no cluster died or was born)
    if (st.n_active != K) w.rows_current = false;
    for (int64_t a = 0; a < st.n_active && w.rows_current; a++)
        if (w.order[a] != a) w.rows_current = false;
    ch->K = st.n_active;
    memcpy(ch->assignment, w.assign.data(), (size_t)N * 8);
    ch->swept += N;
    ch->hint_used += st.hint_used;
    ch->pair_used += st.pair_used;
    ch->triple_used += st.triple_used;
    ch->lane_used += st.lane_used;
    ch->stride_used += st.stride_used;
    std::vector<double>().swap(w.heap_ll);
    if (trace_g) {
        auto us = [](Clock::time_point a, Clock::time_point b) {
            return std::chrono::duration_cast<std::chrono::nanoseconds>(
                b - a).count() / 1e3;
        };
        fprintf(stderr, "[gibbs] N=%lld K=%lld: new-cluster term + priors "
                "%.1f, rows %.1f, issue %.1f, order + hints issue + state "
                "%.1f, wait %.1f, loop %.1f, commit %.1f us\n", (long long)N,
                (long long)K, us(tg_in, tg0), us(tg0, tg1), us(tg1, tg2),
                us(tg2, tg3), us(tg3, tg4), us(tg4, tg5),
                us(tg5, Clock::now()));
    }
    *done = true;
    return 0;
}

// CRP.do_split_move / do_merge_move through bnpc_sm_move.  *done = false: not
// done natively, the stream is where it was.
int move_phase(bnpc_ctx *ctx, const bnpc_host_kernels *k, bnpc_mt19937 *rng,
               bnpc_chain *ch, Work &w, int move, bool *done)
{
    *done = false;
    if (!k->gammaln) return 0;
    bnpc_move_state st;
    memset(&st, 0, sizeof st);
    st.move = move;
    st.scan_no = ch->sm_steps;
    st.view = ch->view_move;
    st.uniform_prior = ch->uniform_prior;
    st.threads = team_for(ch, 3 * ch->M);
    st.threads_wide = ch->threads;
    st.K = ch->K;
    st.ids = ch->ids;
    st.sizes = ch->sizes;
    st.N = ch->N;
    st.M = ch->M;
    st.assignment = ch->assignment;
    st.parameters = ch->parameters;
    st.param_stride = ch->param_stride;
    st.DP_a = ch->DP_a;
    st.sd = ch->sd;
    st.n_sd = ch->n_sd;
    st.FP = ch->FP;
    st.FN = ch->FN;
    st.p = ch->p;
    st.q = ch->q;
    st.tmin = ch->tmin;
    st.tmax = ch->tmax;
    st.fill = ch->mix0;
    st.gauss = ch->gauss;
    int sub = 1;
    // (after the move's last draw of variable length: its acceptance test,
    // then the way to the parameter batch as after a sweep)
    const std::function<void(int)> last_draw = [&](int uniforms_left) {
        if (ch->phase == BNPC_PHASE_ASSIGN)
            params_ahead(ctx, k, rng, ch, uniforms_left);
    };
    bnpc_move_last_draw_hook(&last_draw);
    const int rc = bnpc_sm_move(ctx, k, rng, &st, &sub);
    bnpc_move_last_draw_hook(nullptr);
    if (rc || sub) bnpc_mh_ahead_drop(ctx);
    if (rc) return rc;
    if (sub) return 0;
    *done = true;
    ch->native_moves++;
    ch->sm_accepted = st.accepted;
    ch->sm_cells = st.n_cells;
    if (!st.accepted) return 0;
    w.rows_current = false;
    int64_t at_i = -1, at_j = -1;
    for (int64_t g = 0; g < ch->K; g++) {
        if (ch->ids[g] == st.cl_i) at_i = g;
        if (ch->ids[g] == st.cl_j) at_j = g;
    }
    if (at_i < 0 || (move == 1 && at_j < 0) || (move == 0 && at_j >= 0)
        || (move == 0 && ch->K >= ch->N)) {
        bnpc_set_error("sm_move returned clusters the chain does not hold");
        return 3;
    }
    if (move == 0) {            // dict: cl_i shrinks, cl_j is appended
        ch->sizes[at_i] -= st.moved;
        ch->ids[ch->K] = st.cl_j;
        ch->sizes[ch->K] = st.moved;
        ch->K++;
    } else {                    // dict: cl_i grows, cl_j is deleted
        ch->sizes[at_i] += st.moved;
        for (int64_t g = at_j; g + 1 < ch->K; g++) {
            ch->ids[g] = ch->ids[g + 1];
            ch->sizes[g] = ch->sizes[g + 1];
        }
        ch->K--;
    }
    return 0;
}

// CRP.update_DP_alpha (libs/CRP.py:386-410) + init_DP_prior (:191-194)
int alpha_phase(const bnpc_host_kernels *k, bnpc_mt19937 *rng, bnpc_chain *ch,
                Work &w)
{
    bnpc_legacy_gauss *g = (bnpc_legacy_gauss *)ch->gauss;
    const int64_t N = ch->N;
    double alpha;
    const AlphaArgs args = {ch->DP_a, ch->dpa_shape, ch->dpa_rate, N, ch->K};
    int rc = alpha_draws(k, rng, g, args, &alpha);
    if (rc) return rc;
    ch->DP_a = (1 + EPSILON) < alpha ? alpha : (1 + EPSILON);
    // CRP_prior = [0, log(1..N, DP_a) - log(N - 1 + DP_a)]: NumPy's log loop
    // over the same N + 1 element vector the reference hands it
    w.scratch.resize((size_t)2 * (N + 1));
    double *arg = w.scratch.data(), *lg = arg + (N + 1);
    for (int64_t i = 0; i < N; i++) arg[i] = (double)(i + 1);
    arg[N] = ch->DP_a;
    np_loop(k->np_log, k->np_log_data, arg, lg, N + 1);
    const double denom = np_log1(k, (double)(N - 1) + ch->DP_a);
    ch->crp_prior[0] = 0.0;
    for (int64_t i = 0; i <= N; i++) ch->crp_prior[i + 1] = lg[i] - denom;
    ch->alpha_updated = 1;
    return 0;
}

// CRP.update_parameters (libs/CRP.py:302-311).  *done = false: left to the
// binding, the stream is where it was.
int params_phase(bnpc_ctx *ctx, const bnpc_host_kernels *k, bnpc_mt19937 *rng,
                 bnpc_chain *ch, Work &w, bool *done)
{
    *done = false;
    const int64_t K = ch->K, M = ch->M;
    if (K < 1) return 0;
    const size_t E = (size_t)K * M;
    static const bool trace_p = [] {
        const char *e = getenv("BNPC_TIMING");
        return e && strstr(e, "params");
    }();
    const Clock::time_point tp0 = Clock::now();
    gather_rows(ch, w);
    const Clock::time_point tp1 = Clock::now();
    const bool stale = !counts_current(ctx, ch, w);
    const Clock::time_point tp2 = Clock::now();
    w.n1.resize(E);
    w.n0.resize(E);
    w.fresh.resize(E);
    w.sd_idx.resize(E);
    w.U.resize(E);
    w.u.resize(E);
    w.A.resize(E);
    w.log_prob.resize((size_t)K);
    w.declined.resize((size_t)K);
    const bool want_prior = !ch->uniform_prior;
    if (want_prior) w.prior_out.resize(E);
    const float *kt;
    const double *kp;
    known_prior(ch, w, &kt, &kp);
    bnpc_mh_args a;
    memset(&a, 0, sizeof a);
    a.G = K;
    a.M = M;
    a.old_theta = w.rows.data();
    a.n1 = w.n1.data();
    a.n0 = w.n0.data();
    a.sd = ch->sd;
    a.n_sd = ch->n_sd;
    a.tmin = ch->tmin;
    a.tmax = ch->tmax;
    a.FP = ch->FP;
    a.FN = ch->FN;
    a.p = ch->p;
    a.q = ch->q;
    a.uniform_prior = ch->uniform_prior;
    a.trans_prob = 0;
    a.known_theta = kt;
    a.known_prior = kp;
    a.sd_idx = w.sd_idx.data();
    a.U = w.U.data();
    a.u = w.u.data();
    a.new_theta = w.fresh.data();
    a.prior_out = want_prior ? w.prior_out.data() : nullptr;
    a.A = w.A.data();
    a.log_prob = w.log_prob.data();
    a.declined = w.declined.data();
    a.threads = team_for(ch, (int64_t)E);
    // (the densities' sum in index order - what recording the step needs of
    // them - is made by a rank of the batch's team behind the others)
    double prior_sum = NAN;
    a.prior_seq_sum = want_prior ? &prior_sum : nullptr;
    w.pc_sum_set = false;
    const Snapshot before(rng, (bnpc_legacy_gauss *)ch->gauss);
    const Clock::time_point tp3 = Clock::now();
    int sub = 0, rc;
    if (stale) {
        w.lab_set = false;
        rc = bnpc_label_counts_and_batch(ctx, k, rng, ch->assignment, ch->ids,
                                         &a, &sub);
        if (rc) return rc;
        counts_made(ctx, ch, w);
    } else {
        rc = bnpc_mh_batch_dev(ctx, k, rng, &a, 0, &sub);
        if (rc) return rc;
    }
    if (sub) {      // an element the kernel table leaves to SciPy
        before.put_back(rng, (bnpc_legacy_gauss *)ch->gauss);
        return 0;
    }
    const Clock::time_point tp4 = Clock::now();
    int64_t declined = 0;
    for (int64_t g = 0; g < K; g++) {
        memcpy(ch->parameters + (size_t)ch->ids[g] * ch->param_stride,
               w.fresh.data() + (size_t)g * M, (size_t)M * sizeof(float));
        declined += w.declined[g];
    }
    w.rows.swap(w.fresh);           // = parameters[ids] again
    if (want_prior && E <= ((size_t)1 << 22)) {
        w.pc_ids.assign(ch->ids, ch->ids + K);
        w.pc_theta.assign(w.rows.begin(), w.rows.begin() + E);
        w.pc_prior.swap(w.prior_out);
        w.pc_set = true;
        w.pc_is_rows = true;        // until the rows are gathered afresh
        w.pc_sum = prior_sum;
        w.pc_sum_set = prior_sum == prior_sum;
    } else if (want_prior) {
        w.pc_set = false;
    }
    ch->par_declined = declined;
    ch->par_accepted = (int64_t)E - declined;
    // the batch of a first step (K0 rows: 50 bytes of scratch per entry, 8 GB
    // at config 5) must not stay allocated for the chain's lifetime
    if (w.U.capacity() > 4 * E && w.U.capacity() > ((size_t)1 << 22)) {
        auto trim = [E](auto &v) {
            v.resize(std::min(v.size(), E));
            v.shrink_to_fit();
        };
        trim(w.n1);
        trim(w.n0);
        trim(w.fresh);
        trim(w.sd_idx);
        trim(w.U);
        trim(w.u);
        trim(w.A);
        trim(w.prior_out);
        trim(w.kt);
        trim(w.kp);
        trim(w.dens);
    }
    if (trace_p) {
        auto us = [](Clock::time_point a, Clock::time_point b) {
            return std::chrono::duration_cast<std::chrono::nanoseconds>(
                b - a).count() / 1e3;
        };
        fprintf(stderr, "[params] K=%lld stale=%d: rows %.1f, counts current? "
                "%.1f, buffers + cached prior %.1f, batch %.1f, results %.1f "
                "us\n", (long long)K, (int)stale, us(tp0, tp1), us(tp1, tp2),
                us(tp2, tp3), us(tp3, tp4), us(tp4, Clock::now()));
    }
    *done = true;
    return 0;
}

// CRP_errors_learning.MH_error_rates (libs/CRP_learning_errors.py:66-111) for
// one rate.  *done = false: a scalar the kernel table leaves to SciPy.
int error_rate(bnpc_ctx *ctx, const bnpc_host_kernels *k, bnpc_mt19937 *rng,
               bnpc_chain *ch, Work &w, int which, bool *done)
{
    *done = false;
    const double old = which == 0 ? ch->FP : ch->FN;
    const double *sds = which == 0 ? ch->FP_sd : ch->FN_sd;
    const double sd = sds[mt_interval(rng, 2)];         // np.random.choice
    const double a = (0 - old) / sd, b = (1 - old) / sd;
    const double q = mt_double(rng);                    // np.random.uniform()
    double fresh = 0.0, fwd = 0.0, rev = 0.0, p_new = 0.0, p_old = 0.0;
    int st = 0;
    if (bnpc_tn_ppf_scalar(k, q, a, b, old, sd, &fresh, &st) || st) return 0;
    if (bnpc_tn_logpdf_scalar(k, fresh, a, b, old, sd, &fwd, &st) || st)
        return 0;
    if (bnpc_tn_logpdf_scalar(k, old, (0 - fresh) / sd, (1 - fresh) / sd,
                              fresh, sd, &rev, &st) || st)
        return 0;
    if (!(fresh > 0.0 && fresh < 1.0)) return 0;
    if (!error_prior(k, ch, w, which, fresh, &p_new)
        || !error_prior(k, ch, w, which, old, &p_old))
        return 0;
    double FP[2], FN[2], ll[2];
    if (which == 0) {
        FP[0] = fresh;
        FP[1] = old;
        FN[0] = FN[1] = ch->FN;
    } else {
        FP[0] = FP[1] = ch->FP;
        FN[0] = fresh;
        FN[1] = old;
    }
    const int rc = bnpc_ll_total(ctx, w.rows.data(), ch->K, FP, FN, 2, ll);
    if (rc) return rc;
    const double A = ll[0] + p_new - ll[1] - p_old + rev - fwd;
    const bool accept = np_log1(k, mt_double(rng)) < A;
    if (accept) (which == 0 ? ch->FP : ch->FN) = fresh;
    (which == 0 ? ch->FP_accepted : ch->FN_accepted) = accept ? 1 : 0;
    *done = true;
    return 0;
}

// TraceStore.put_state / put_params (bnpc_amd/mcmc.py; libs/MCMC.py:242-282)
int record_phase(bnpc_ctx *ctx, const bnpc_host_kernels *k, bnpc_chain *ch,
                 Work &w, bool *done)
{
    *done = false;
    const int64_t K = ch->K, M = ch->M;
    // the scalar densities first: if one is left to SciPy nothing is pending
    double lprior;
    if (!(w.alpha_prior.set && w.alpha_prior.key == ch->DP_a)) {
        double val;
        if (!gamma_logpdf(k, ch->DP_a, ch->dpa_shape, ch->dpa_rate, &val))
            return 0;
        w.alpha_prior.key = ch->DP_a;
        w.alpha_prior.val = val;
        w.alpha_prior.set = true;
    }
    double pFP = 0.0, pFN = 0.0;
    if (ch->learning && (!error_prior(k, ch, w, 0, ch->FP, &pFP)
                         || !error_prior(k, ch, w, 1, ch->FN, &pFN)))
        return 0;
    int rc = ensure_counts(ctx, ch, w);
    if (rc) return rc;
    gather_rows(ch, w);
    rc = bnpc_ll_total_issue(ctx, w.rows.data(), K, &ch->FP, &ch->FN, 1);
    if (rc) return rc;
    // get_lprior_full under the launch (libs/CRP.py:241-251)
    double csum = 0.0;
    for (int64_t g = 0; g < K; g++) {
        const double v = ch->crp_prior[ch->sizes[g]];
        csum = g ? csum + v : v;
    }
    lprior = w.alpha_prior.val + csum;
    if (!ch->uniform_prior) {
        double seq = 0.0;
        if (w.pc_set && w.pc_is_rows
                && w.pc_prior.size() == (size_t)K * M) {
            // the rows are the ones the parameter batch of this step has just
            // produced, with their densities: the sum in index order is all
            // that is left (a compare and a copy per element otherwise: the
            // largest part of recording a config-5 step)
            if (w.pc_sum_set) {
                seq = w.pc_sum;
            } else {
                const double *d = w.pc_prior.data();
                seq = d[0];
                for (int64_t i = 1; i < K * M; i++) seq += d[i];
            }
        } else {
            const float *kt;
            const double *kp;
            known_prior(ch, w, &kt, &kp);
            w.dens.resize((size_t)K * M);
            rc = bnpc_beta_logpdf_f32(k, w.rows.data(), K * M, ch->p, ch->q,
                                      kt, kp, w.dens.data(), &seq,
                                      team_for(ch, K * M));
            if (rc) {
                double drop;
                (void)bnpc_ll_total_wait(ctx, &drop);
                return rc;
            }
        }
        lprior += seq;
    }
    if (ch->learning) lprior = lprior + pFP + pFN;
    // what does not depend on the total: the labels, the parameter rows
    if (ch->rec_assignment)
        memcpy(ch->rec_assignment, ch->assignment, (size_t)ch->N * 8);
    ch->rec_params_done = 0;
    if (ch->rec_params && K <= ch->rec_params_cap) {
        w.sorted.assign(ch->ids, ch->ids + K);
        std::sort(w.sorted.begin(), w.sorted.end());
        for (int64_t g = 0; g < K; g++)
            memcpy(ch->rec_params + (size_t)g * M,
                   ch->parameters + (size_t)w.sorted[g] * ch->param_stride,
                   (size_t)M * sizeof(float));
        ch->rec_params_done = 1;
    }
    double ML = 0.0;
    rc = bnpc_ll_total_wait(ctx, &ML);
    if (rc) return rc;
    ch->ML = ML;
    ch->lprior = lprior;
    if (ch->rec_scalars[0]) *ch->rec_scalars[0] = ML;
    if (ch->rec_scalars[1]) *ch->rec_scalars[1] = ML + lprior;
    if (ch->rec_scalars[2]) *ch->rec_scalars[2] = ch->DP_a;
    if (ch->rec_scalars[3]) *ch->rec_scalars[3] = ch->FN;
    if (ch->rec_scalars[4]) *ch->rec_scalars[4] = ch->FP;
    *done = true;
    return 0;
}

}   // namespace

extern "C" int bnpc_gamma_logpdf_scalar(const bnpc_host_kernels *k, double x,
                                        double a, double loc, double *out,
                                        int *status)
{
    if (!k || !out || !status) {
        bnpc_set_error("bad argument: gamma_logpdf_scalar");
        return 2;
    }
    *status = gamma_logpdf(k, x, a, loc, out) ? 0 : 1;
    return 0;
}

// Checker hook (CPU tests): CRP.update_DP_alpha alone on the caller's stream
extern "C" int bnpc_chain_update_alpha(const bnpc_host_kernels *k,
                                       bnpc_mt19937 *rng, bnpc_chain *ch)
{
    if (!k || !rng || !ch || !ch->work || !ch->crp_prior || !ch->gauss
        || ch->K < 1) {
        bnpc_set_error("bad argument: chain_update_alpha");
        return 2;
    }
    return alpha_phase(k, rng, ch, *(Work *)ch->work);
}

extern "C" int bnpc_chain_open(bnpc_chain *ch)
{
    if (!ch || ch->N < 1 || ch->M < 1) {
        bnpc_set_error("bad argument: chain_open");
        return 2;
    }
    Work *w = new Work();
    w->N = ch->N;
    w->M = ch->M;
    ch->work = w;
    return 0;
}

extern "C" int bnpc_chain_close(bnpc_chain *ch)
{
    if (ch && ch->work) {
        delete (Work *)ch->work;
        ch->work = nullptr;
    }
    return 0;
}

extern "C" int bnpc_chain_step(bnpc_ctx *ctx, const bnpc_host_kernels *k,
                               bnpc_mt19937 *rng, bnpc_chain *ch)
{
    if (!ctx || !k || !rng || !ch || !ch->work || !ch->assignment
        || !ch->parameters || !ch->ids || !ch->sizes || !ch->crp_prior
        || !ch->sd || !ch->gauss || ch->n_sd < 1 || ch->K < 1
        || ch->K > ch->N || ch->param_stride < ch->M
        || ch->phase < BNPC_PHASE_ASSIGN || ch->phase > BNPC_PHASE_RECORD) {
        bnpc_set_error("bad argument: chain_step");
        return 2;
    }
    Work &w = *(Work *)ch->work;
    if (w.N != ch->N || w.M != ch->M) {
        bnpc_set_error("bad argument: the chain was opened for another shape");
        return 2;
    }
    // BNPC_TIMING=step: the call's own wall time next to what its phase
    // clocks hold (what is outside them: between the phases, in here)
    static const bool trace_s = [] {
        const char *e = getenv("BNPC_TIMING");
        return e && strstr(e, "step");
    }();
    struct StepTrace {
        bnpc_chain *ch;
        bool on;
        Clock::time_point t0;
        int64_t ns0;
        int64_t laps() const
        {
            int64_t s = 0;
            for (int i = 0; i < 9; i++) s += ch->clock_ns[i];
            return s;
        }
        StepTrace(bnpc_chain *c, bool o)
            : ch(c), on(o), t0(Clock::now()), ns0(o ? laps() : 0) {}
        ~StepTrace()
        {
            if (!on) return;
            const double total = std::chrono::duration_cast<
                std::chrono::nanoseconds>(Clock::now() - t0).count() / 1e3;
            fprintf(stderr, "[step] move %d: %.1f us in the call, %.1f in "
                    "its phases\n", (int)ch->move, total,
                    (laps() - ns0) / 1e3);
        }
    } step_trace(ch, trace_s);
    bnpc_legacy_gauss *gauss = (bnpc_legacy_gauss *)ch->gauss;
    // the binding may have changed anything between two calls
    w.rows_current = false;
    w.pc_is_rows = false;
    ch->need = BNPC_NEED_NONE;
    int rc;
    bool done;
    if (ch->phase == BNPC_PHASE_ASSIGN) {
        ch->move = -1;
        ch->sm_accepted = 0;
        ch->sm_cells = 0;
        ch->alpha_updated = ch->errors_updated = 0;
        ch->FP_accepted = ch->FN_accepted = 0;
        ch->par_declined = ch->par_accepted = 0;
        ch->rec_params_done = 0;
    }
    if (ch->phase <= BNPC_PHASE_ASSIGN && !ch->fix_assign) {
        if (mt_double(rng) < ch->sm_prob) {
            // update_assignments_split_merge (libs/CRP.py:417-431)
            int move;
            if (ch->K == 1)
                move = 0;
            else if (ch->K == ch->N)
                move = 1;
            else
                move = (int)np_choice_p(ch->sm_ratios, 2, w.cdf,
                                        mt_double(rng));
            if (move > 1) {
                bnpc_set_error("bad argument: split/merge ratios");
                return 2;
            }
            ch->move = move;
            Lap lap(ch, -1);
            rc = move_phase(ctx, k, rng, ch, w, move, &done);
            if (rc) return rc;
            if (!done) {
                ch->need = BNPC_NEED_MOVE;
                return 0;
            }
            lap.slot = 1 + 2 * move + (ch->sm_accepted ? 0 : 1);
            lap.stop();
        } else {
            ch->move = 2;
            Lap lap(ch, 0);
            rc = gibbs_phase(ctx, k, rng, ch, w, &done);
            if (rc) return rc;
            if (!done) {
                ch->need = BNPC_NEED_GIBBS;
                return 0;
            }
            lap.stop();
        }
    }
    if (ch->phase <= BNPC_PHASE_ALPHA && !ch->fix_assign) {
        if (mt_double(rng) < ch->dpa_prob) {
            Lap lap(ch, 5);
            rc = alpha_phase(k, rng, ch, w);
            if (rc) return rc;
            lap.stop();
        }
    }
    if (ch->phase <= BNPC_PHASE_PARAMS) {
        Lap lap(ch, 6);
        rc = params_phase(ctx, k, rng, ch, w, &done);
        if (rc) return rc;
        if (!done) {
            ch->need = BNPC_NEED_PARAMS;
            return 0;
        }
        lap.stop();
    }
    if (ch->phase <= BNPC_PHASE_ERRORS && ch->learning) {
        if (mt_double(rng) < ch->error_prob) {
            Lap lap(ch, 7);
            const Snapshot before(rng, gauss);
            const double FP0 = ch->FP, FN0 = ch->FN;
            rc = ensure_counts(ctx, ch, w);
            if (rc) return rc;
            gather_rows(ch, w);
            for (int which = 0; which < 2; which++) {
                rc = error_rate(ctx, k, rng, ch, w, which, &done);
                if (rc) return rc;
                if (!done) {
                    before.put_back(rng, gauss);
                    ch->FP = FP0;
                    ch->FN = FN0;
                    ch->FP_accepted = ch->FN_accepted = 0;
                    ch->need = BNPC_NEED_ERRORS;
                    return 0;
                }
            }
            ch->errors_updated = 1;
            lap.stop();
        }
    }
    // do_step alone (no recording target at all: the caller records through
    // update_results, which evaluates likelihood and prior itself)
    bool record = ch->rec_assignment || ch->rec_params;
    for (int i = 0; i < 5; i++) record = record || ch->rec_scalars[i];
    if (record) {
        Lap lap(ch, 8);
        rc = record_phase(ctx, k, ch, w, &done);
        if (rc) return rc;
        if (!done) {
            ch->need = BNPC_NEED_RECORD;
            return 0;
        }
        lap.stop();
    }
    ch->steps++;
    return 0;
}
