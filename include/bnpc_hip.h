/*
 * bnpc_hip.h - C-ABI of libbnpc_hip.so: the MI355X (gfx950) implementation of
 * BnpC's Bernoulli error-model log-likelihood hot path.
 *
 * The reference (cbg-ethz/BnpC) is pure Python and has no FFI of its own; the
 * drop-in boundary is the duck-typed class surface of libs/CRP.py and
 * libs/CRP_learning_errors.py.  This header is the
 * C-ABI that sits UNDER that surface; bnpc_amd/_lib.py binds it with ctypes
 * and bnpc_amd/model.py mirrors the reference classes on top of it.  Each
 * entry point cites the reference expression(s) it replaces
 * (paths relative to /root/reference).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message
 *     is available from bnpc_last_error() (thread local).  HIP errors never
 *     abort the process.
 *   - all host buffers are caller-allocated, C-contiguous and only borrowed
 *     for the duration of the call; the context owns all device memory.
 *   - one context per process/chain; calls on one context must not overlap.
 *   - threading: the library keeps ONE host thread team per process (the
 *     parameter batches, the team scan of a first sweep).  Entry points that
 *     use it - bnpc_mh_batch, bnpc_log_accept, bnpc_beta_logpdf_f32,
 *     bnpc_gibbs_sweep, bnpc_rg_scan_step - may be called from several host
 *     threads at once (each on its own context / buffers): they take turns
 *     on the team, one job at a time.  The team is rebuilt in a fork()ed
 *     child on first use; a child must not be forked WHILE a call is running.
 *   - floating point: tables, accumulators and outputs are float64; theta is
 *     float32 and (1 - theta) is evaluated in float32, as in the reference.
 *
 * Device data layout
 *   row planes   rows[N][W] of {ones, zeros} 64-bit words, W = ceil(M/64)
 *   slot views   masks[block][m] of {ones, zeros} 64-bit LANE MASKS over a
 *                block of 64 cells ("slots"): bit s of masks[b][m].ones says
 *                cell slot 64*b+s has x = 1 at mutation m.  View 0 is the
 *                identity view (all N cells in order); views 1.. are gathered
 *                cell lists (restricted-Gibbs moves, single cells, the two
 *                alternating tiles of a tiled sweep).
 */
#ifndef BNPC_HIP_H
#define BNPC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bnpc_ctx bnpc_ctx;

#define BNPC_MAX_VIEWS 7
#define BNPC_TILE_SLOTS 4    /* tiles in flight (bnpc_ll_rows_issue) */
#define BNPC_MAX_TRIALS 4

/* ---- library / device ---------------------------------------------------- */
const char *bnpc_last_error(void);
int bnpc_abi_version(void);
int bnpc_device_count(int *count);
/* name[len] receives the gfx arch name, *cus the CU count */
int bnpc_device_info(int device, char *name, int len, int *cus);

/* PCI address "dddd:bb:dd.f" of a device: the binding reads the device's NUMA
 * node from sysfs with it and keeps the chain's host threads on that node
 * (the sweep walks a matrix the GPU has just written to host memory there). */
int bnpc_device_pci_bus_id(int device, char *bus_id, int len);

/* Diagnostic: `jobs` team jobs of 1..max_tasks counted tasks each on up to
 * `ranks` ranks, and beside every other one a counting job on the process's
 * aside thread (the thread that takes a parameter batch's draws ahead)
 * (tests/test_native_sweeps.py stress test; the ThreadSanitizer build runs
 * it).  *done receives the number of tasks executed, which must equal
 * *expected. */
int bnpc_team_stress(int64_t jobs, int max_tasks, int ranks, uint64_t seed,
                     int64_t *done, int64_t *expected);
/* ranks the host thread team has after growing it to `threads` (fewer when
 * the system refuses threads; capped at 255) */
int bnpc_team_size(int threads);

/* Rows of a chain's sample trace (the reference's `results['params']`,
 * libs/MCMC.py:267-282: np.zeros at the first post-burn-in sample, np.pad for
 * every new largest cluster count, np.append when a run by time grows it)
 * written for the first time on the host team: row r of dst (dst_stride
 * bytes apart) = `copy` bytes of row r of src (src_stride apart; src unused if
 * copy == 0), then zeros up to `width` bytes.  copy <= width <= dst_stride. */
int bnpc_rows_copy_zero(void *dst, int64_t dst_stride, const void *src,
                        int64_t src_stride, int64_t rows, int64_t copy,
                        int64_t width, int threads);

/* ---- context: data of one chain ------------------------------------------
 * Replaces the float64 N x M `self.data` (NaN = missing) of libs/CRP.py:30-31
 * by two bit planes resident in HBM.  data_nan holds 0 | 1 | NaN. */
int bnpc_create(int device, int64_t N, int64_t M, const double *data_nan,
                bnpc_ctx **out);
/* same, from int8 codes 0 | 1 | 3 (the reference's on-disk code,
 * libs/dpmmIO.py:27-98 maps 3 -> NaN, 2 -> 1) */
int bnpc_create_codes(int device, int64_t N, int64_t M, const int8_t *codes,
                      bnpc_ctx **out);
/* same, from the packed form itself (libs/dpmmIO.py:27-98 parsed once, kept
 * as a bit-plane file: bnpc_amd/bitplanes.py): planes[N][W][2] 64-bit words
 * {ones, zeros}, W = ceil(M / 64), bit b of word w = mutation 64 w + b.  No
 * float64 or int8 matrix is ever materialised. */
int bnpc_create_planes(int device, int64_t N, int64_t M,
                       const uint64_t *planes, bnpc_ctx **out);
/* host-side packing / unpacking of that form: codes 0 | 1 | 2 (-> 1) | 3 with
 * arbitrary element strides (a transposed view packs without a copy);
 * unpack gathers rows `cells` (NULL: all N rows, n == N). */
int bnpc_pack_codes(const int8_t *codes, int64_t N, int64_t M,
                    int64_t row_stride, int64_t col_stride, uint64_t *planes);
int bnpc_unpack_codes(const uint64_t *planes, int64_t N, int64_t M,
                      const int64_t *cells, int64_t n, int8_t *codes);
int bnpc_destroy(bnpc_ctx *ctx);
/* Kernel-selection switches (BNPC_* environment variables, README) are read
 * when a context is created; this reads them again (A/B tools, tests). */
int bnpc_reload_options(bnpc_ctx *ctx);
int bnpc_shape(const bnpc_ctx *ctx, int64_t *N, int64_t *M);

/* per-cell counts of observed 1s and 0s (row sums of the planes).  With them
 * libs/CRP.py:230-234 (get_lpost_single_new_cluster) is n1*c1 + n0*c0. */
int bnpc_cell_counts(bnpc_ctx *ctx, int32_t *n1, int32_t *n0);

/* ---- slot views ----------------------------------------------------------
 * Gather the rows of `cells` (n indices into 0..N-1, any order, repeats
 * allowed) and transpose them into 64-slot lane-mask blocks.  Replaces the
 * fancy-index gathers `self.data[cells]` of libs/CRP.py:360, 557-560, 636-637,
 * 726-728.  view in 1..BNPC_MAX_VIEWS-1 (view 0 = all cells, built at create). */
int bnpc_view_set(bnpc_ctx *ctx, int view, const int64_t *cells, int64_t n);
/* The same for a tile of a tiled sweep (the rows `self.data[cells]` of
 * libs/CRP.py:270 for a stretch of the permutation): the list is staged in the
 * pinned buffer of tile slot `slot` and the call returns without waiting; the
 * view is for bnpc_ll_rows_issue on that slot. */
int bnpc_view_set_slot(bnpc_ctx *ctx, int view, const int64_t *cells,
                       int64_t n, int slot);
int bnpc_view_size(const bnpc_ctx *ctx, int view, int64_t *n);

/* ---- log-likelihood of every slot of a view under K parameter vectors ----
 * out[s*K + k] = sum over mutations m, IN INDEX ORDER, skipping missing
 *   entries, of log(theta[k,m]*P(x|1) + (1-theta[k,m])*P(x|0))
 * = CRP._calc_ll(x, theta)            libs/CRP.py:197-204 (axis=1 form)
 *   CRP._Bernoulli_FN / _Bernoulli_FP libs/CRP.py:207-212
 * One launch replaces the N calls of CRP.get_lpost_single per Gibbs sweep
 * (libs/CRP.py:223-227, 270) and the pair of calls of CRP._rg_get_ll
 * (libs/CRP.py:635-638).  theta is K x M float32 row-major.  The per-element
 * logs are evaluated once per (k, m) on the device (2*K*M logs instead of
 * N*K*M).  out has row stride ldo >= K doubles (0 means K; columns K..ldo
 * are left for clusters opened later in the sweep).  out may be NULL
 * (compute only; used for timing - theta must then stay alive until
 * bnpc_sync / bnpc_timer_stop). */
int bnpc_ll_theta(bnpc_ctx *ctx, int view, const float *theta, int64_t K,
                  double FP, double FN, double *out, int64_t ldo);

/* bnpc_ll_theta into a context-owned PINNED host buffer: *host points at
 * slots x ldo doubles (DMA target, no pageable bounce) that stay valid until
 * the next *_pinned call on this context; the caller may
 * write into it (columns K..ldo are free for clusters opened mid-sweep). */
int bnpc_ll_theta_pinned(bnpc_ctx *ctx, int view, const float *theta,
                         int64_t K, double FP, double FN, int64_t ldo,
                         double **host);

/* bnpc_ll_theta_pinned plus, per slot, the four largest entries of
 *   out[s, k] + col_prior[k]   (k < K <= 64)
 * with the columns and log-likelihoods of the three largest: a HINT for the
 * sequential sweep, which then scores a cell in O(1) instead of O(K) whenever
 * the winner is beyond doubt or the cell is torn between two or three
 * columns with everything else far below (bnpc_gibbs_state.hint).  col_prior[k] is the log prior of column k's
 * cluster at launch time (CRP_prior[size], libs/CRP.py:226).  *top2 points
 * into pinned host memory, valid until the next call on the context. */
typedef struct bnpc_top2 {
    double best, second;    /* largest / second largest entry of the row */
    /* third / fourth largest (-inf with fewer columns) as float32 ROUNDED UP:
     * they only bound "everything else" from above (61 below the candidates) */
    float third, fourth;
    /* exp(ll_second - ll_best), exp(ll_third - ll_best) as float32: the
     * weights of the row's second / third column relative to its first
     * WITHOUT the priors.  The prior of a live cluster is log(size) minus a
     * term common to all (libs/CRP.py:83-85), so under the CURRENT sizes the
     * candidates' probabilities are proportional to size1 : e2 size2 : e3
     * size3 - no exp() in the loop; the float's 6e-8 is covered by the band of
     * 1e-6 the quick picks keep between the uniform and an interval end
     * (nearer: the scan's own arithmetic decides) */
    float e2, e3;
    /* the log-likelihoods (no prior) behind best / second / third: with them
     * the loop re-scores the row's candidates under the CURRENT priors
     * exactly as a scan would (row[c] + prior[c]) - a cell torn between two
     * or three close clusters (the pieces of a fresh split), everything else
     * far below, is decided from these entries alone */
    double ll_best, ll_second, ll_third;
    int16_t col;            /* column of the largest (first one on ties) */
    int16_t col2, col3;     /* of the second / third largest, -1 if none */
    int16_t row_here;       /* 1: the row's first K entries are already in the
                             * host matrix (written through by the hint kernel
                             * because there is no clear winner and a fourth
                             * entry is within reach: a row the sweep will
                             * scan) - no need to wait for the
                             * matrix copy to read them */
} bnpc_top2;                /* 64 bytes: one cache line per cell */
int bnpc_ll_theta_pinned_top2(bnpc_ctx *ctx, int view, const float *theta,
                              int64_t K, double FP, double FN, int64_t ldo,
                              const double *col_prior, double **host,
                              bnpc_top2 **top2);
/* the same without the final wait (when hints are produced): the caller
 * prepares the sweep under the launch and calls bnpc_sync before it reads
 * the hints */
int bnpc_ll_theta_pinned_top2_issue(bnpc_ctx *ctx, int view,
                                    const float *theta, int64_t K, double FP,
                                    double FN, int64_t ldo,
                                    const double *col_prior, double **host,
                                    bnpc_top2 **top2);
/* the hints of the last ..._issue call are complete on return */
int bnpc_hints_wait(bnpc_ctx *ctx);
/* bnpc_ll_theta_pinned_top2_issue in two halves, for a sweep that reads its
 * hints IN VISITING ORDER: ..._sums_issue queues the element tables and the
 * sums and returns; the caller draws its visiting order under that launch
 * (np.random.permutation is the sweep's first draw, libs/CRP.py:259, and
 * consumes no result of the device); bnpc_hints_in_order_issue(ctx, order,
 * &top2) then queues the hint kernel with record r of *top2 made from row
 * order[r] of the matrix - the loop (libs/CRP.py:260-288) walks the records
 * front to back instead of all over 3 MB of device-written memory (config 5:
 * 14 -> 11 ns per cell) and the order's 0.2 ms are hidden.  The matrix
 * itself stays indexed by cell.  bnpc_hints_wait as usual; the loop takes the
 * records with bnpc_gibbs_state.hint_in_order = 1.  (A sweep in visiting
 * order AND in row chunks, the loop running under the sums, was measured in
 * round 5 and is slower: profiles/r05/chunked_sweep_experiment.) */
int bnpc_ll_theta_pinned_sums_issue(bnpc_ctx *ctx, int view,
                                    const float *theta, int64_t K, double FP,
                                    double FN, int64_t ldo,
                                    const double *col_prior, double **host);
int bnpc_hints_in_order_issue(bnpc_ctx *ctx, const int64_t *order,
                              bnpc_top2 **top2);
/* When *top2 is returned non-NULL the matrix behind *host has NOT been copied
 * yet: it stays on the device until bnpc_matrix_wait fetches it (the sweep
 * reads it only where a hint is in doubt - a settled sweep never does; when
 * the previous hinted sweep did, the copy is queued behind the hints at once
 * and bnpc_matrix_wait only waits for it).
 * Valid until the next log-likelihood call on the context.  bnpc_gibbs_sweep
 * calls it itself through bnpc_gibbs_state.matrix_wait. */
int bnpc_matrix_wait(bnpc_ctx *ctx);

/* Resident parameter rows for tiled sweeps: store row r holds the float32
 * parameter vector of cluster id r (libs/CRP.py:155-180 keeps them in an
 * N x M array indexed by id).  bnpc_theta_put copies R rows starting at row0;
 * bnpc_ll_rows_pinned is bnpc_ll_theta_pinned with cluster k taken from store
 * row rows[k] - no host-side gather `parameters[cl_ids]` (libs/CRP.py:224) and
 * no re-upload per tile. */
int bnpc_theta_put(bnpc_ctx *ctx, int64_t row0, const float *theta, int64_t R);
int bnpc_ll_rows_pinned(bnpc_ctx *ctx, int view, const int64_t *rows,
                        int64_t K, double FP, double FN, int64_t ldo,
                        double **host);

/* The same evaluation split in two, so that a tiled sweep overlaps the device
 * work of tile t+1 with the host's sequential loop over tile t
 * (libs/CRP.py:260-288 is sequential per cell, the likelihood rows are not):
 * bnpc_ll_rows_issue enqueues tables + sums + the copy into pinned buffer
 * `slot` (0 .. BNPC_TILE_SLOTS-1) and returns at once; bnpc_ll_rows_wait
 * blocks until that buffer is complete and returns it (slots x ldo doubles,
 * valid until the next issue on the same slot).  Several tiles may be in
 * flight: the sums of consecutive issues alternate between two device buffers
 * and the copies run on their own stream, so the copy of tile t overlaps the
 * sums of tile t+1.  Other calls on the context may be made in between; they
 * run beside the issued work on a side stream. */
int bnpc_ll_rows_issue(bnpc_ctx *ctx, int view, const int64_t *rows, int64_t K,
                       double FP, double FN, int64_t ldo, int slot);
int bnpc_ll_rows_wait(bnpc_ctx *ctx, int slot, double **host);
/* The same with the tile's HINTS (bnpc_gibbs_state.hint for a tile): per row
 * the largest entry of out[s, k] + col_prior[k] over the K issued columns, its
 * column and the largest entry among the other columns (col_prior[k] = the log
 * prior of column k's cluster at issue time, CRP_prior[size]).  K may be tens
 * of thousands: the record is a bnpc_top2 whose column is the 32-bit number
 * (uint16) col | (uint16) col2 << 16, col3 = -1, row_here = 2, third = fourth
 * = -inf.  From the moment the true clusters exist nearly every cell of a
 * first sweep is dominated by one of them: the host's sequential loop then
 * looks at one record per cell instead of walking 30 000 columns
 * (libs/CRP.py:268-277).  *hint: rows of the view, valid until the next issue
 * on the slot; NULL if the tile was issued without hints. */
int bnpc_ll_rows_issue_hint(bnpc_ctx *ctx, int view, const int64_t *rows,
                            int64_t K, double FP, double FN, int64_t ldo,
                            int slot, const double *col_prior);
int bnpc_ll_rows_wait_hint(bnpc_ctx *ctx, int slot, double **host,
                           bnpc_top2 **hint);

/* Same sums from caller-built element tables: L1[k,m] is the value an
 * observed 1 contributes, L0[k,m] an observed 0 (both K x M float64).  With
 * tables built by the caller's NumPy this reproduces the reference's sums bit
 * for bit (same elements, same order); used for CRP._rg_init_split
 * (libs/CRP.py:547-561), whose `ll_j > ll_i` comparison is the one discrete
 * decision on the path, and for get_lpost_single_new_cluster. */
int bnpc_ll_tables(bnpc_ctx *ctx, int view, const double *L1, const double *L0,
                   int64_t K, double *out, int64_t ldo);

/* ---- column counts --------------------------------------------------------
 * For G segments of cells (CSR: cells[seg_offsets[g] .. seg_offsets[g+1])),
 * n1[g*M + m] / n0[g*M + m] = number of cells of the segment with an observed
 * 1 / 0 at mutation m.  Exact integers.  They turn the per-mutation sums over
 * a cell subset of CRP._get_log_A (libs/CRP.py:359-368), the Beta shapes of
 * CRP._init_cl_params_new (libs/CRP.py:183-188) and the flat sums of
 * CRP._get_ll_ratio (libs/CRP.py:716-733) into O(M) table work. */
int bnpc_colcounts(bnpc_ctx *ctx, const int64_t *cells,
                   const int64_t *seg_offsets, int64_t G,
                   int32_t *n1, int32_t *n0);

/* Column counts of G segments of a slot view, from its lane masks:
 * labels[s] = segment (0..G-1) of slot s, or < 0 for none.  Same integers as
 * bnpc_colcounts on the corresponding cell lists, without cell lists, atomics
 * or zero-fill: the restricted-Gibbs launch clusters of a split/merge move
 * are the two segments of the move's own view (libs/CRP.py:590-606). */
int bnpc_view_counts(bnpc_ctx *ctx, int view, const int64_t *labels,
                     int64_t G, int32_t *n1, int32_t *n0);

/* Per-cluster column counts for the K cluster ids `ids` (segment g = cells
 * with assignment == ids[g]).  The counts also stay resident on the device
 * for bnpc_ll_total.  n1/n0 may be NULL (device-resident only). */
int bnpc_colcounts_by_label(bnpc_ctx *ctx, const int64_t *assignment,
                            const int64_t *ids, int64_t K,
                            int32_t *n1, int32_t *n0);

/* Total log-likelihood of the data under the resident per-cluster counts,
 *   out[e] = sum_k sum_m n1[k,m]*log(theta*(1-FN_e) + (1-theta)*FP_e)
 *                      + n0[k,m]*log(theta*FN_e + (1-theta)*(1-FP_e))
 * for E <= BNPC_MAX_TRIALS trial error pairs in one launch.  Replaces
 * CRP.get_ll_full (libs/CRP.py:237-238) and
 * CRP_errors_learning.get_ll_full_error (libs/CRP_learning_errors.py:58-63).
 * theta is K x M float32 (row g = parameters of ids[g]). */
int bnpc_ll_total(bnpc_ctx *ctx, const float *theta, int64_t K,
                  const double *FP, const double *FN, int E, double *out);
/* The same in two halves: issue launches the sum and returns; wait picks the
 * E totals up.  The driver records the state of a step as likelihood + prior
 * (libs/MCMC.py:252-254): the prior is host work that runs under the launch.
 * Other calls on the context may be made in between (they first let the
 * pending kernel finish reading its staged parameters); one total at a time. */
int bnpc_ll_total_issue(bnpc_ctx *ctx, const float *theta, int64_t K,
                        const double *FP, const double *FN, int E);
int bnpc_ll_total_wait(bnpc_ctx *ctx, double *out);

/* ---- timing on the context's stream (HIP events) -------------------------- */
/* Re-issue the cells x clusters x mutations kernel of the last bnpc_ll_theta /
 * bnpc_ll_tables call `reps` times on the resident tables, bracketed by HIP
 * events on the context's stream; *ms_per_launch = average kernel duration.
 * Measurement only (bench.py roofline). */
int bnpc_bench_ll(bnpc_ctx *ctx, int reps, float *ms_per_launch);
/* the same for the WHOLE evaluation of that call (element tables, sums and,
 * for mutation-split launches, the combine): what a converged-K call costs on
 * the device.  Valid right after the call (its inputs are still staged). */
int bnpc_bench_ll_full(bnpc_ctx *ctx, int reps, float *ms_per_call);
/* which kernel(s) that call launched ("k_ll8_asm<2, true> + k_ll_combine"),
 * for how many clusters, in how many mutation chunks */
int bnpc_last_launch(const bnpc_ctx *ctx, char *name, int len, int64_t *K,
                     int *mutation_chunks);
/* Parameter batches whose draws were taken ahead of the batch on a copy of
 * the stream (bnpc_chain_step): walkers started / adopted
 * by a batch / rows of draws adopted, since the context was created. */
int bnpc_mh_ahead_stats(bnpc_ctx *ctx, int64_t *begun, int64_t *taken,
                        int64_t *rows);
/* Per-launch device timers.  on = 1: from now on every kernel this library
 * launches (any context of the process) carries a start / stop event pair
 * that takes the dispatch's own timestamps.  on = 0: timers off, the device
 * synchronised, *device_ms = the sum of the kernel durations since they were
 * switched on, *launches = their number (either may be NULL).  Measurement
 * only (bench.py: window.device_ms_per_step, taken on steps AFTER the timed
 * window - a pair of events costs the host ~2 us per launch). */
int bnpc_launch_timers(bnpc_ctx *ctx, int on, double *device_ms,
                       int64_t *launches);
int bnpc_timer_start(bnpc_ctx *ctx);
int bnpc_timer_stop(bnpc_ctx *ctx, float *ms);
int bnpc_sync(bnpc_ctx *ctx);

/* ---- native sequential sweeps (host side, exact legacy-MT19937 replica) ---
 * NumPy's legacy global stream (np.random.seed / get_state) is MT19937; C
 * and Python draw from ONE stream in the reference's order (SURVEY.md
 * Appendix B).  bnpc_mt19937 has the layout of NumPy's own mt19937_state, so
 * the binding passes the address of the global RandomState's state
 * (bit_generator.ctypes.state_address) and the draws happen in place; a copy
 * exchanged through np.random.get_state() / set_state() works as well.
 */
typedef struct bnpc_mt19937 {
    uint32_t key[624];
    int32_t pos;
} bnpc_mt19937;

/* np.random.random_sample(): ((a >> 5) * 2^26 + (b >> 6)) / 2^53 */
double bnpc_mt_random_sample(bnpc_mt19937 *rng);
/* np.random.permutation(n): arange + Fisher-Yates from the top with the
 * masked-rejection legacy random_interval on 32-bit draws */
int bnpc_mt_permutation(bnpc_mt19937 *rng, int64_t n, int64_t *out);

/* The three draws of CRP.MH_cluster_params (libs/CRP.py:328-335) for G
 * clusters in the reference's per-cluster order: choice(sd, M) as indices
 * 0..n_sd-1 (legacy randint), the M uniforms of truncnorm.rvs, random(M). */
int bnpc_mt_mh_draws(bnpc_mt19937 *rng, int64_t G, int64_t M, int64_t n_sd,
                     int32_t *sd_idx, double *U, double *u);

/* NumPy's legacy Beta sampler on the same stream (np.random.beta of
 * libs/CRP.py:183-188, 155-180, 563-567): Johnk's algorithm for shapes <= 1,
 * else two legacy standard gammas (Marsaglia-Tsang on the polar-method
 * Gaussian / the shape < 1 rejection), libm pow / log / exp / sqrt.  The
 * polar method caches its second variate in NumPy's RandomState
 * (has_gauss, gauss), state every normal / gamma draw of the stream shares:
 * `g` points at it (in place when the binding can locate it, else a copy the
 * binding writes back). */
typedef struct bnpc_legacy_gauss {
    int32_t has_gauss;
    int32_t pad_;
    double gauss;
} bnpc_legacy_gauss;
/* out[i] = Beta(a[i], b[i]), i in order (the element order of NumPy's
 * broadcast loop) */
int bnpc_mt_beta(bnpc_mt19937 *rng, bnpc_legacy_gauss *g, int64_t n,
                 const double *a, const double *b, double *out);
/* theta[m] = float32(clip(Beta(p + n1[m] * fkt, q + n0[m] * fkt), tmin,
 * tmax)): a profile row from column counts (CRP._init_cl_params_new) */
int bnpc_mt_beta_theta(bnpc_mt19937 *rng, bnpc_legacy_gauss *g, int64_t M,
                       double p, double q, const int32_t *n1,
                       const int32_t *n0, double fkt, double tmin,
                       double tmax, float *theta);

/* np.random.gamma(shape, scale) on the same stream (legacy_gamma = scale *
 * legacy_standard_gamma): the draws of CRP.update_DP_alpha
 * (libs/CRP.py:386-410). */
int bnpc_mt_gamma(bnpc_mt19937 *rng, bnpc_legacy_gauss *g, double shape,
                  double scale, double *out);

/* log(exp(log_p[i]) - exp(log_q[i])) for log_q <= log_p, evaluated with the
 * arithmetic of scipy.special.logsumexp([log_p, log_q + pi*1j], axis=0).real
 * - which is how scipy.stats.truncnorm computes the Gaussian mass of an
 * interval that lies left of zero (scipy/stats/_continuous_distns.py
 * _log_diff / _log_gauss_mass; reached from truncnorm.logpdf / rvs at
 * libs/CRP.py:331,351-357 whenever a profile entry sits on the upper clip):
 *     E = exp(q - p); out = (log(hypot(1 - E, E * sin(pi))) + 0) + p
 * with the C library's exp / sincos / hypot / log, i.e. the functions NumPy's
 * complex exp and log1p resolve to.  The binding bit-compares it against SciPy
 * once per process and does not use it if anything differs. */
int bnpc_log_diff_pi(const double *log_p, const double *log_q, int64_t n,
                     double *out);

/* ---- batched Metropolis-Hastings update of cluster parameters -------------
 * CRP.MH_cluster_params (libs/CRP.py:314-344) with _get_log_A (:347-383) for
 * G clusters in ONE call: the draws of every cluster in the reference's order
 * on the caller's stream (choice(sd, M) -> the M uniforms of truncnorm.rvs ->
 * random(M)), then the per-element arithmetic - truncated-normal proposal
 * (ppf of the uniform), forward / reverse proposal log-density, the Beta prior
 * log-density, the four element-table logs, the acceptance test - on a team of
 * host threads, without the interpreter.
 *
 * The trajectory contract (identical assignments under a fixed seed) needs
 * the proposal's float64 bits to equal SciPy's: a 1-ulp difference flips the
 * float32 cast of a parameter with probability ~2e-9 per element.  So the
 * transcendental kernels are NOT restated here: the caller hands over the
 * addresses of the very functions SciPy and NumPy evaluate (scipy.special's
 * C entry points as exported by scipy.special.cython_special.__pyx_capi__,
 * NumPy's float64 inner loops of log / exp / log1p / expm1 as registered in
 * the ufunc objects), and this library calls them in the order, association
 * and dtypes of scipy.stats.truncnorm._ppf / _logpdf / _log_gauss_mass and
 * beta._logpdf (scipy/stats/_continuous_distns.py) and of the reference's
 * expressions.  The binding bit-compares the whole batch against the
 * SciPy-level evaluation once per process and does not use it if anything
 * differs. */
typedef double (*bnpc_sf1)(double, int);            /* cython_special f(x)   */
typedef double (*bnpc_sf2)(double, double, int);    /* cython_special f(x,y) */
typedef void (*bnpc_uloop)(char **args, const intptr_t *dimensions,
                           const intptr_t *steps, void *data);

typedef struct bnpc_host_kernels {
    bnpc_sf1 ndtr, log_ndtr, ndtri_exp, sc_log1p;   /* scipy.special */
    bnpc_sf2 xlogy, xlog1py, betaln;
    bnpc_uloop np_log, np_exp, np_log1p, np_expm1;  /* NumPy d->d loops */
    void *np_log_data, *np_exp_data, *np_log1p_data, *np_expm1_data;
    double norm_pdf_logC;   /* scipy.stats._continuous_distns._norm_pdf_logC */
    int left_ok;            /* bnpc_log_diff_pi verified against SciPy */
    bnpc_sf1 gammaln;       /* scipy.special.gammaln (bnpc_sm_move), or NULL */
} bnpc_host_kernels;

typedef struct bnpc_mh_args {
    int64_t G, M;
    const float *old_theta;     /* G x M current parameters (float32) */
    const int32_t *n1, *n0;     /* G x M observed 1s / 0s of each cluster */
    const double *sd;           /* n_sd proposal standard deviations */
    int64_t n_sd;
    double tmin, tmax;          /* truncation bounds, libs/CRP.py:13-14 */
    double FP, FN;
    double p, q;                /* Beta prior shapes */
    int uniform_prior;          /* p == q == 1: prior terms are 0 */
    int trans_prob;             /* libs/CRP.py:339-342: clip A, log(1-e^A) */
    /* optional cache of the prior log-density: where known_theta[g,m] has
     * the bits of old_theta[g,m], known_prior[g,m] is its density */
    const float *known_theta;
    const double *known_prior;
    /* the draws, written first (G x M each) */
    int32_t *sd_idx;
    double *U, *u;
    /* results */
    float *new_theta;           /* G x M */
    double *prior_out;          /* G x M density of new_theta, or NULL */
    double *A;                  /* G x M log acceptance ratios (work/out) */
    double *log_prob;           /* G: sum of A in index order (trans_prob) */
    int64_t *declined;          /* G */
    int threads;                /* <= 1: the calling thread only */
    /* optional verdicts of the device screen (bnpc_mh_screen), G x M bytes:
     * 0 = declined for certain (the element keeps old_theta and is not
     * evaluated), 2 = accepted for certain (only the proposal's bits and
     * its prior density are evaluated), 3 = accepted for certain AND the
     * proposal's float32 bits are screen_theta[g, m] (only its prior density
     * is evaluated; without screen_theta: as 2), anything else = evaluate in
     * full.  Needs rng == NULL (the screen saw the draws) and trans_prob == 0
     * (scored batches need every A). */
    const uint8_t *screen;
    /* optional, G x M: the proposals of the entries flagged 3 (the device
     * evaluated them in float64 and found them further from both float32
     * rounding boundaries than its value and SciPy's can be apart) */
    const float *screen_theta;
    /* screened batches, optional: flagged_estimate > 0 = about how many
     * entries carry a non-zero flag (the team is sized by it; 0: the flags are
     * counted first, one pass of the calling thread); flag_counts != NULL
     * receives the numbers found: [0] in doubt, [1] accepted for certain,
     * [2] accepted with the device's bits */
    int64_t flagged_estimate;
    int64_t *flag_counts;
    /* screened batches with prior_out, optional: *prior_seq_sum receives the
     * sum of prior_out over the whole batch in INDEX order - row by row, one
     * accumulator: the order of the reference's bn.nansum over
     * param_prior.logpdf(parameters[cl_ids]) (libs/CRP.py:247-250).  On entry
     * NaN: the sum starts with the first entry; anything else: it continues
     * from that value (the parts of one batch chain their sums).  One rank of
     * the team adds the rows up behind the others as they complete them:
     * 250 000 dependent adds (config 5: 0.17 ms of every recorded step) run
     * under the batch instead of after it. */
    double *prior_seq_sum;
} bnpc_mh_args;

/* *status = 0: done.  *status = 1: the draws were taken (sd_idx, U, u are
 * valid, the stream has advanced) but some element needs a branch this
 * library leaves to SciPy (an interval right of zero, non-finite terms, a
 * uniform that is exactly 0): the caller evaluates the batch from the draws.
 * rng == NULL: the caller has filled sd_idx / U / u already. */
int bnpc_mh_batch(const bnpc_host_kernels *k, bnpc_mt19937 *rng,
                  const bnpc_mh_args *a, int *status);

/* The device screen of a parameter batch (libs/CRP.py:314-383, decision
 * only): flags[g, m] = 0 where the proposal of element (g, m) is declined
 * FOR CERTAIN - its log acceptance ratio, evaluated on the device in plain
 * float64 with an explicit bound on everything that can separate it from the
 * host's SciPy-exact value, lies below log(u) - 2 where it is accepted for
 * certain (the same bound the other way), and 1 where the host has to
 * evaluate the element in full (in doubt, or on a branch the kernel does not
 * model).  The counts come from the device: counts_src 0 = the
 * resident per-cluster counts of the last bnpc_colcounts_by_label (G = its
 * K), 1 = the two segments of the last bnpc_view_counts (G = 2, or 3 with
 * row 2 = their sum: the merged cluster of a restricted scan).  a->sd_idx /
 * U / u hold the draws; a->trans_prob must be 0. */
/* ... 3 where, in addition to 2, the float32 bits of the proposal are
 * beyond doubt: new32[g, m] (optional output, defined where the flag is 3;
 * BNPC_MH_SCREEN=2: no 3s). */
int bnpc_mh_screen(bnpc_ctx *ctx, int counts_src, const bnpc_mh_args *a,
                   uint8_t *flags, float *new32);
/* bnpc_mh_batch with that screen in front: the draws are taken (rng != NULL)
 * straight into pinned memory, the device screens them against the resident
 * counts, the host evaluates what is left (a few per cent) exactly as
 * bnpc_mh_batch does - same results bit for bit, a tenth of the host work.
 * Scored batches (trans_prob), small ones and BNPC_MH_SCREEN=0 go to
 * bnpc_mh_batch directly.  On return the caller's sd_idx / U / u hold the
 * draws. */
int bnpc_mh_batch_dev(bnpc_ctx *ctx, const bnpc_host_kernels *k,
                      bnpc_mt19937 *rng, const bnpc_mh_args *a,
                      int counts_src, int *status);
/* CRP.update_parameters (libs/CRP.py:302-311) in one call:
 * bnpc_colcounts_by_label for `assignment` / ids[0..a->G) followed by
 * bnpc_mh_batch_dev on those counts, the screens queued behind the counts
 * kernel (one wait less).  a->n1 / a->n0 are OUTPUTS here: the caller's
 * K x M arrays receive the counts. */
int bnpc_label_counts_and_batch(bnpc_ctx *ctx, const bnpc_host_kernels *k,
                                bnpc_mt19937 *rng, const int64_t *assignment,
                                const int64_t *ids, const bnpc_mh_args *a,
                                int *status);
/* elements screened so far on this context / of those, left to the host */
int bnpc_mh_screen_stats(bnpc_ctx *ctx, int64_t *screened, int64_t *kept);

/* CRP._get_log_A (libs/CRP.py:347-383) for GIVEN rows: the log acceptance
 * ratio of moving old -> new under proposal standard deviations std, forward
 * truncation bounds (fmin, fmax) around old, reverse bounds (tmin, tmax)
 * around new (libs/CRP.py:674-681: TMIN / TMAX both; :777-799: 0 / 1 forward),
 * A[g, m] clipped at 0 if clip, sum[g] its sum in index order.  Same kernel
 * table and *status convention as bnpc_mh_batch. */
typedef struct bnpc_accept_args {
    int64_t G, M;
    const float *new_theta, *old_theta;     /* G x M */
    const double *std;                      /* G x M */
    const int32_t *n1, *n0;                 /* G x M column counts */
    double fmin, fmax, tmin, tmax;
    double FP, FN, p, q;
    int uniform_prior, clip;
    double *A;                              /* G x M */
    double *sum;                            /* G */
    int threads;
} bnpc_accept_args;
int bnpc_log_accept(const bnpc_host_kernels *k, const bnpc_accept_args *a,
               int *status);

/* One intermediate restricted-Gibbs scan of a split/merge move in one call
 * (libs/CRP.py:535-537 -> :570-606): log-likelihoods of the view's slots
 * under the parameter rows 0 and 1 of mh->old_theta (device), the 2-way
 * assignment scan over slots 1..n-2 on the stream (bnpc_rg_scan mode 0),
 * column counts of the two launch clusters for the new assignment (slot 0
 * belongs to the first, slot n-1 to the second; row 2 = their sum) into n1 /
 * n0 (G x M, the buffers mh->n1 / mh->n0 point at), then bnpc_mh_batch on the
 * rows: mh->G == 3 (launch clusters + merged cluster, an intermediate scan)
 * or 2 (the launch clusters only: the scored final scan of a split,
 * libs/CRP.py:672, with mh->trans_prob set; *scan_log_prob then carries the
 * assignment scan's part).  *status as bnpc_mh_batch. */
int bnpc_rg_scan_step(bnpc_ctx *ctx, const bnpc_host_kernels *k,
                      bnpc_mt19937 *rng, int view, int64_t n,
                      int64_t *rg_assignment, double DP_a,
                      const bnpc_mh_args *mh, int32_t *n1, int32_t *n0,
                      double *scan_log_prob, int *status);

/* Beta(p, q) log-density of n float32 values (scipy.stats.beta._logpdf with
 * the public wrapper's support handling), re-using known_prior[i] where
 * known_theta[i] has the bits of x[i]; *seq_sum (optional) receives the sum
 * in index order.  libs/CRP.py:249 (get_lprior_full), :371-376. */
int bnpc_beta_logpdf_f32(const bnpc_host_kernels *k, const float *x, int64_t n,
                         double p, double q, const float *known_theta,
                         const double *known_prior, double *out,
                         double *seq_sum, int threads);

/* scipy.stats.truncnorm.logpdf(x, a, b, loc, scale) for scalar arguments (the
 * error-rate proposals and priors, libs/CRP_learning_errors.py:47-49, 85-91)
 * on the same kernel table; *status = 1: not evaluated (left to SciPy). */
int bnpc_tn_logpdf_scalar(const bnpc_host_kernels *k, double x, double a,
                          double b, double loc, double scale, double *out,
                          int *status);

/* scipy.stats.truncnorm.ppf(q, a, b, loc, scale) for scalars: the error-rate
 * proposal given its one uniform (libs/CRP_learning_errors.py:81-84). */
int bnpc_tn_ppf_scalar(const bnpc_host_kernels *k, double q, double a,
                       double b, double loc, double scale, double *out,
                       int *status);

/* Checker hook: the cumulative sums np.cumsum(p) holds for the probability
 * vector p[top] = 1.0, p[a != top] = 1e-15-floor (a = 0..A) - the case in
 * which one cluster dominates _normalize_log_probs (libs/CRP.py:88-100).  The
 * native sweep evaluates them in closed form instead of walking the array;
 * the tests compare this with NumPy's cumsum bit for bit. */
int bnpc_dominated_cdf(int64_t A, int64_t top, double *cdf);
/* Checker hooks: the pick of a Gibbs cell torn between two live entries (`top`
 * the first maximum, `sec` the runner-up d2 <= 0 below it, the other A - 1 of
 * the A + 1 entries on the 1e-15 floor) and of a cell of a restricted 2-way
 * scan, given the uniform u.  quick = 0: the reference's arithmetic
 * (_normalize_log_probs / _normalize_log + np.random.choice, four libm calls);
 * quick = 1: the decision the native loops try first - one exp(), u compared
 * with the interval ends it implies, *pick = -1 when u is within 1e-11 of an
 * end (the loops then take the quick = 0 path); quick = 2 (pair and triple):
 * the form the sweep loop uses with a hint record - the entries' weights with
 * the exponential rounded to float32 (bnpc_top2.e2 / e3) and a band of 1e-6.
 * A decided quick pick must equal the full one: the tests compare them around
 * the ends. */
int bnpc_pair_pick(int quick, double d2, int64_t A, int64_t top, int64_t sec,
                   double u, int64_t *pick);
int bnpc_two_way_pick(int quick, double p0, double p1, double u,
                      int64_t *pick);
/* ... and of a cell with three live entries q[0..3) at list positions
 * a[0..3) (the other A - 2 entries far below): quick = 1 from two exp(), -1 =
 * not decided; quick = 0 by the scan's arithmetic over the whole row. */
int bnpc_triple_pick(int quick, const double *q, const int64_t *a, int64_t A,
                     double u, int64_t *pick);

/* The sequential per-cell loop of CRP.update_assignments_Gibbs
 * (libs/CRP.py:260-288, with _normalize_log_probs :88-100 and
 * np.random.choice(p=...) :276-277) over a precomputed log-likelihood matrix.
 *
 * Columns of `ll` are clusters ("col"); a column keeps its cluster id for the
 * whole sweep.  `order[0..n_active)` lists the live columns in the insertion
 * order of the reference's cells_per_cluster dict (a cluster that loses its
 * last cell is deleted; a cluster opened later is appended).
 *
 * The function processes perm[pos..pos_end) and returns 0 when
 *   - the window is complete: st->pos == st->pos_end, st->new_cell == -1, or
 *   - the cell perm[st->pos - 1] drew a NEW cluster: st->new_cell == that
 *     cell.  The cell has already been removed from its old cluster.  The
 *     caller opens the cluster in Python (lowest free id, Beta draws on the
 *     same stream - libs/CRP.py:291-299, 183-188), writes its log-likelihood
 *     column into ll[:, n_cols], registers it (col_id, col_size = 1,
 *     col_of_id, order[n_active++], n_cols++, assignment[cell]) and calls
 *     again.
 */
typedef struct bnpc_gibbs_state {
    int64_t n_cells;    /* N */
    int64_t ld;         /* allocated columns of ll (row stride) */
    int64_t n_cols;     /* columns in use */
    int64_t n_active;   /* live clusters = entries of order[] */
    int64_t pos;        /* next position of perm to process */
    int64_t new_cell;   /* out: cell that drew a new cluster, else -1 */
    int64_t pos_end;    /* process perm[pos..pos_end) (N for a whole sweep) */
    int64_t row_base;   /* -1: row of ll = cell id (whole matrix resident);
                           >= 0: row of ll = position in perm - row_base (the
                           matrix of one permutation-ordered tile of cells) */
    int64_t threads;    /* host threads for the scan of a cell over thousands
                           of live clusters (first sweeps); <= 1: none.  The
                           result does not depend on it. */
    /* optional (NULL: none): per ROW of ll (row_base < 0: per cell, at most
     * 64 columns, from bnpc_ll_theta_pinned_top2; row_base >= 0: per position
     * of the tile, any number of columns, from bnpc_ll_rows_wait_hint) the
     * largest entries of ll[row, k] + hint_prior[k] over the first hint_cols
     * columns.  Where those, widened by how far the priors have moved since,
     * leave no doubt about the winner, the cell is not scanned.  The result
     * does not depend on it. */
    const struct bnpc_top2 *hint;
    const double *hint_prior;
    int64_t hint_cols;
    int64_t hint_used;  /* out: cells decided from the hint, accumulated */
    /* if not NULL: called (once) with matrix_wait_arg before the first read
     * of ll, e.g. bnpc_matrix_wait with its context */
    int (*matrix_wait)(void *);
    void *matrix_wait_arg;
    int64_t pair_used;  /* out: of hint_used, cells decided between the row's
                         * two best columns (accumulated) */
    /* Optional: clusters are OPENED inside the call (libs/CRP.py:281-282,
     * 291-299, 183-188) instead of returning to the caller - the lowest free
     * id, its profile row drawn with NumPy's legacy Beta sampler on the same
     * stream from the one cell's observations (bnpc_mt_beta), written to
     * theta_host[id], its column evaluated on the device and written into
     * ll.  birth_ctx == NULL: off.  A birth the call cannot make (no spare
     * column, born[] full) returns through new_cell as before. */
    struct bnpc_ctx *birth_ctx;
    int32_t birth_view;     /* the view whose slots are the rows of ll */
    int32_t birth_put;      /* != 0: also store the row with bnpc_theta_put */
    int64_t birth_rows;     /* rows of ll */
    float *theta_host;      /* parameter rows, row = cluster id, stride M */
    double beta_p, beta_q, tmin, tmax, FP, FN;
    void *gauss;            /* bnpc_legacy_gauss of the stream */
    int64_t *born;          /* out: ids opened in this call, in order */
    int64_t born_cap, n_born;
    int64_t triple_used;    /* out: of hint_used, cells decided among the row's
                             * three best columns */
    /* row_base < 0 and != 0: hint[p] belongs to the cell at POSITION p of
     * perm (bnpc_hints_in_order_issue), not to cell p; the rows of ll stay
     * indexed by cell */
    int64_t hint_in_order;
    /* out: of hint_used, the cells decided in the loop's LANE (accumulated):
     * a compact second loop inside bnpc_gibbs_sweep for the cells a record
     * decides - dominated, or picked between two / among three candidates -
     * while no column has been born since the launch; it stops in front of
     * everything else and the general iteration takes over (same steps, same
     * arithmetic, same stream: the result does not depend on it;
     * BNPC_SWEEP_LANE=0 switches it off). */
    int64_t lane_used;
    /* out: of lane_used, the cells taken whole runs at a time (accumulated):
     * cells dominated by the cluster they sit in, their uniforms peeked out
     * of the state block and found clear of 0 and 1 - the lane's STRIDE
     * (bnpc_sweeps.cpp: record_lane; off with the lane, or alone with
     * BNPC_SWEEP_LANE=nostride). */
    int64_t stride_used;
} bnpc_gibbs_state;

int bnpc_gibbs_sweep(bnpc_gibbs_state *st, bnpc_mt19937 *rng,
                     const int64_t *perm,      /* N visiting order */
                     const double *ll,         /* rows x ld, see row_base */
                     const double *post_new,   /* N new-cluster log posterior */
                     const double *crp_prior,  /* N+2 log prior by size */
                     int64_t *assignment,      /* N cluster ids, in/out */
                     int64_t *col_of_id,       /* N column of a live id / -1 */
                     int64_t *col_id,          /* ld cluster id of a column */
                     int64_t *col_size,        /* ld cells in that cluster */
                     int64_t *order,           /* ld live columns, dict order */
                     double *scratch);         /* 2 * (ld + 1) doubles */

/* The sequential 2-way loops of the restricted Gibbs scans over an S x 2
 * log-likelihood matrix (row s = non-anchor cell s, column 0/1 = launch
 * cluster i/j):
 *   mode 0  CRP._rg_scan_assign (libs/CRP.py:616-629): permutation(S), then
 *           per cell one 2-way draw (one uniform each);
 *   mode 1  the scoring loop of CRP._rg_get_split_prob (libs/CRP.py:808-818):
 *           index order, no RNG, cells are forced to `target`.
 * rg_assignment (S, values 0/1) is updated in place; *log_prob receives the
 * sum in index order of the chosen log-probabilities (mode 1: required; mode
 * 0: NULL for an unscored scan - the loop may then take a cell's pick from
 * one exp() where the uniform is clear of the boundary, bnpc_two_way_pick).  Uses
 * CRP._normalize_log (libs/CRP.py:103-116) and CRP.log_CRP_prior (:83-85). */
int bnpc_rg_scan(bnpc_mt19937 *rng, int mode, int64_t S, const double *ll,
                 double DP_a, int64_t *rg_assignment, const int64_t *target,
                 double *log_prob);

/* ---- a whole split / merge move in one call --------------------------------
 * CRP.do_split_move / do_merge_move (libs/CRP.py:434-524): the proposal (which
 * cluster(s), which anchors - np.random.choice with and without p, with and
 * without replacement, on the caller's stream), run_rg_nc (:527-544: the
 * launch state from the anchors' rows, `scan_no` restricted scans, the scored
 * last scan), the four ratios of the acceptance test (:641-820) and, when the
 * move is accepted, the new assignment and parameter rows written in place.
 * The sequence of device and host calls is the one the binding makes through
 * the entry points above; the NumPy expressions in between are restated on
 * NumPy's own log loop (k->np_log), SciPy's gammaln (k->gammaln) and np.sum's
 * pairwise order.
 *   in:  move 0 = split, 1 = merge; ids / sizes = the K live clusters in the
 *        order of the binding's dict; assignment (N, updated on acceptance);
 *        parameters (row = cluster id, param_stride floats apart, rows cl_i /
 *        cl_j updated on acceptance); fill = the value a missing entry of an
 *        anchor's row stands for (libs/CRP.py:557-560); view = a slot view
 *        the move may overwrite; gauss = the stream's bnpc_legacy_gauss.
 *   out: accepted; split: cl_i = the cluster that was split, cl_j = the id
 *        the `moved` cells went to; merge: cl_i = the cluster that remains,
 *        cl_j = the one whose `moved` cells joined it; n_cells, log_A.
 * *status = 1: not done here (a move of at most 2 cells, an element of a
 * parameter batch the kernel table leaves to SciPy, no host copy of the rows):
 * the stream and the cached Gaussian are where they were before the call and
 * nothing else was modified - the caller runs the move step by step. */
typedef struct bnpc_move_state {
    int32_t move, scan_no, view, uniform_prior, threads, threads_wide;
    int64_t K;
    const int64_t *ids, *sizes;
    int64_t N, M;
    int64_t *assignment;
    float *parameters;
    int64_t param_stride;
    double DP_a;
    const double *sd;
    int64_t n_sd;
    double FP, FN, p, q, tmin, tmax, fill;
    void *gauss;
    int32_t accepted, pad_;
    int64_t cl_i, cl_j, moved, n_cells;
    double log_A;
} bnpc_move_state;
int bnpc_sm_move(bnpc_ctx *ctx, const bnpc_host_kernels *k, bnpc_mt19937 *rng,
                 bnpc_move_state *st, int *status);
/* Checker hooks: the proposal alone (cells = [i, S..., j], n_first = cells of
 * the first cluster of a merge, picked = positions in ids, size_data =
 * libs/CRP.py:452-455 / :507-508, others = the K - 1 other sizes of a split),
 * and np.sum of a float64 vector. */
int bnpc_move_propose(const bnpc_host_kernels *k, bnpc_mt19937 *rng,
                      const bnpc_move_state *st, int64_t *cells,
                      int64_t *n_cells, int64_t *n_first, int64_t *picked,
                      double *size_data, int64_t *others, int *status);
int bnpc_np_sum(const double *a, int64_t n, double *out);

/* ---- a whole MCMC step in one call -----------------------------------------
 * Chain_steps.do_step + Chain.update_results (libs/MCMC.py:320-342, 242-282):
 *
 *   [u < sm_prob ? split/merge : Gibbs] -> [u < dpa_prob ? DP alpha]
 *   -> update_parameters -> [learning & u < error_prob ? error rates]
 *   -> record (ML = get_ll_full, MAP = ML + get_lprior_full, alpha, FN, FP,
 *      the assignment, the parameter rows of the populated clusters)
 *
 * on the caller's stream, in the reference's draw order, without the
 * interpreter in between.  Nothing here is new arithmetic: the call strings
 * together bnpc_sm_move, bnpc_gibbs_sweep (on bnpc_ll_theta_pinned_top2_issue),
 * bnpc_label_counts_and_batch / bnpc_mh_batch_dev, bnpc_ll_total(_issue),
 * bnpc_beta_logpdf_f32, bnpc_tn_ppf_scalar / bnpc_tn_logpdf_scalar exactly as
 * the binding (bnpc_amd/model.py, bnpc_amd/mcmc.py) strings them together, and
 * restates the scalar NumPy / SciPy expressions between them - CRP.update_DP_alpha
 * (libs/CRP.py:386-410: legacy beta / gamma draws, init_DP_prior :191-194),
 * CRP_errors_learning.MH_error_rates (libs/CRP_learning_errors.py:66-111),
 * CRP.get_lprior_full (libs/CRP.py:241-251; the Gamma log-density of alpha on
 * SciPy's xlogy / gammaln) - on the kernel table.
 *
 * The model's state lives in the CALLER's arrays, updated in place:
 * assignment, parameters, the live clusters (ids / sizes, in the insertion
 * order of the reference's cells_per_cluster dict; capacity N), CRP_prior,
 * DP_a, FP, FN.  A phase the call does not make itself - a Gibbs sweep whose
 * matrix (N x (K + spare columns) doubles) does not fit `sweep_bytes` or has
 * more than 32 767 columns (first steps: tiled by the binding), a move / batch
 * / scalar the kernel table leaves to SciPy - ends the call BEFORE that phase, with the
 * stream where the reference would have it at that point: `need` names the
 * phase, the binding runs it through its own methods and calls again with
 * `phase` = the next one.  need == 0: the step is complete and recorded.
 */
#define BNPC_PHASE_ASSIGN 0     /* split/merge or Gibbs */
#define BNPC_PHASE_ALPHA 1      /* the DP-alpha draw and update */
#define BNPC_PHASE_PARAMS 2
#define BNPC_PHASE_ERRORS 3
#define BNPC_PHASE_RECORD 4
#define BNPC_NEED_NONE 0
#define BNPC_NEED_MOVE 1        /* do_split_move (move 0) / do_merge_move (1) */
#define BNPC_NEED_GIBBS 2       /* update_assignments_Gibbs */
#define BNPC_NEED_PARAMS 3      /* update_parameters */
#define BNPC_NEED_ERRORS 4      /* update_error_rates */
#define BNPC_NEED_RECORD 5      /* get_ll_full / get_lprior_full + the traces */
#define BNPC_STEP_CLOCKS 10

typedef struct bnpc_chain {
    /* ---- the model (caller-owned, updated in place) ---- */
    int64_t N, M;
    int64_t *assignment;        /* N */
    float *parameters;          /* row = cluster id, param_stride floats apart */
    int64_t param_stride;
    int64_t *ids, *sizes;       /* capacity N; [0, K) = the live clusters */
    int64_t K;
    double *crp_prior;          /* N + 2 (CRP.CRP_prior) */
    double DP_a, dpa_shape, dpa_rate;   /* DP_a_gamma (libs/CRP.py:47-56) */
    double FP, FN;
    double p, q, tmin, tmax;
    double mix0, mix1;          /* _beta_mix_const (libs/CRP.py:42-44) */
    int32_t uniform_prior, learning;
    const double *sd;           /* param_proposal_sd */
    int64_t n_sd;
    double FP_prior[4], FN_prior[4];    /* a, b, mean, sd of the priors */
    double FP_sd[3], FN_sd[3];          /* proposal sds of the error rates */
    /* ---- move schedule (libs/MCMC.py:26-60) ---- */
    double sm_prob, dpa_prob, error_prob, sm_ratios[2];
    int32_t sm_steps, fix_assign;
    /* ---- host resources ---- */
    int32_t threads, threads_wide;      /* team ranks; for >= wide_from */
    int64_t wide_from;                  /* ... batch entries */
    int64_t sweep_bytes;                /* budget of a whole-matrix sweep */
    int32_t view_move, sweep_hint;      /* slot view of the moves; hints on */
    void *gauss;                        /* the stream's bnpc_legacy_gauss */
    /* ---- control ---- */
    int32_t phase;                      /* in: BNPC_PHASE_* to (re)start at */
    int32_t need;                       /* out: BNPC_NEED_* */
    /* ---- where the recorded state goes (NULL: nowhere) ---- */
    double *rec_scalars[5];             /* ML, MAP, DP_alpha, FN, FP slots */
    int64_t *rec_assignment;            /* N labels */
    float *rec_params;                  /* rec_params_cap x M, or NULL */
    int64_t rec_params_cap;
    /* ---- what the step did ---- */
    int32_t move;               /* -1 none, 0 split, 1 merge, 2 Gibbs */
    int32_t sm_accepted;
    int64_t sm_cells;
    int32_t alpha_updated, errors_updated, FP_accepted, FN_accepted;
    int64_t par_declined, par_accepted;
    int32_t rec_params_done, pad_;
    double ML, lprior;
    /* ---- running statistics ---- */
    int64_t swept, hint_used, pair_used, triple_used, native_moves, steps;
    int64_t lane_used;          /* of hint_used: bnpc_gibbs_state.lane_used */
    int64_t stride_used;        /* of lane_used: bnpc_gibbs_state.stride_used */
    /* wall time by part of the step, ns / calls: 0 Gibbs, 1 split accepted,
     * 2 split rejected, 3 merge accepted, 4 merge rejected, 5 DP alpha,
     * 6 parameters, 7 error rates, 8 record, 9 of 0: the sweep waiting for
     * the device's evaluation after its own preparations */
    int64_t clock_ns[BNPC_STEP_CLOCKS], clock_calls[BNPC_STEP_CLOCKS];
    void *work;                 /* bnpc_chain_open / bnpc_chain_close */
} bnpc_chain;

/* ch->work: the step's private caches (the new-cluster term per (FP, FN), the
 * label counts with the state they belong to, the prior-density cache, scratch).
 * open once after filling N and M; close releases them. */
int bnpc_chain_open(bnpc_chain *ch);
int bnpc_chain_close(bnpc_chain *ch);
int bnpc_chain_step(bnpc_ctx *ctx, const bnpc_host_kernels *k,
                    bnpc_mt19937 *rng, bnpc_chain *ch);
/* Checker hook: CRP.update_DP_alpha + init_DP_prior (libs/CRP.py:386-410,
 * 191-194) alone, as the step makes them: reads K, DP_a, dpa_shape, dpa_rate,
 * N; writes DP_a and crp_prior.  Host only. */
int bnpc_chain_update_alpha(const bnpc_host_kernels *k, bnpc_mt19937 *rng,
                            bnpc_chain *ch);
/* Checker hook: scipy.stats.gamma.logpdf(x, a, loc) (scale 1) on the kernel
 * table - xlogy(a - 1, x - loc) - (x - loc) - gammaln(a) - as the step
 * evaluates the prior of DP_a (libs/CRP.py:245); *status = 1: left to SciPy. */
int bnpc_gamma_logpdf_scalar(const bnpc_host_kernels *k, double x, double a,
                             double loc, double *out, int *status);

/* ---- data ingest ---------------------------
 * Body scanner for the reference's text matrix format (libs/dpmmIO.py:27-98):
 * fields 0|1|2|3 (2 -> 1; 3 or empty -> missing) separated by `sep`.  Skips
 * `skip_rows` header lines and, if skip_index, the first field of each line.
 * Call once with out == NULL to get rows/cols, then with an int8 buffer of
 * rows*cols (file orientation; codes 0, 1, 3).  Host only, no GPU. */
int bnpc_parse_matrix(const char *path, char sep, int skip_rows,
                      int skip_index, int8_t *out, int64_t *rows,
                      int64_t *cols);

/* ---- posterior co-clustering distance ------
 * differ[(i,j)] = number of the S posterior samples in which cells i < j carry
 * different cluster labels, condensed in scipy pdist order (N*(N-1)/2 int32).
 * Replaces the per-sample pdist accumulation of utils.get_dist
 * (libs/utils.py:90-97); the mean distance is differ / S.  assignments is
 * S x N int32 row-major.  Stand-alone: allocates and frees its own device
 * memory. */
int bnpc_codist(int device, const int32_t *assignments, int64_t S, int64_t N,
                int32_t *differ);

/* The posterior estimator as a device pipeline (libs/utils.py:90-145): the
 * pair counts of bnpc_codist stay on the device (bnpc_post); the mean
 * distance differ / S is fetched once for SciPy's Ward linkage; every
 * candidate cut of the tree is then scored in ONE pass over the counts:
 * bnpc_post_mpear returns, per candidate clustering,
 *   same_differ[c] = sum over pairs i < j with equal labels of differ_ij
 * (int64, exact, order-free), from which MPEAR (Fritsch & Ickstadt 2009, eq.
 * 13) follows with the label counts and *differ_sum - the float64 similarity
 * matrix `pi` and the reference's O(N^2) pass per candidate are never made.
 * labels: C x N uint16, candidate-major. */
typedef struct bnpc_post bnpc_post;
int bnpc_post_create(int device, const int32_t *assignments, int64_t S,
                     int64_t N, bnpc_post **out, int64_t *differ_sum);
int bnpc_post_fetch(bnpc_post *post, int32_t *differ, double *dist);
int bnpc_post_mpear(bnpc_post *post, const uint16_t *labels, int64_t C,
                    int64_t *same_differ);
/* Ward linkage of the mean distances (`linkage(dist, method='ward')`,
 * libs/utils.py:104) on the device: SciPy's nearest-neighbour-chain algorithm
 * (scipy/cluster/_hierarchy.pyx nn_chain + _ward) inside one launch, the
 * float64 distance vector never leaving the device.  Z_raw[(N - 1) x 4]: the
 * merges in the order the chain makes them (slot indices x < y, height, size);
 * the binding applies SciPy's final stable sort by height and relabelling.
 * Return code 5 (and nothing else): the working set does not fit the device's
 * free memory - the caller may take SciPy's routine on the condensed vector;
 * every other failure is an error. */
int bnpc_post_ward(bnpc_post *post, double *Z_raw);
/* diagnostic: full row scans and chain steps of the last bnpc_post_ward (a
 * chain step whose row still knows its nearest neighbour needs no scan) */
int bnpc_post_ward_stats(const bnpc_post *post, int64_t *scans,
                         int64_t *steps);
int bnpc_post_destroy(bnpc_post *post);

#ifdef __cplusplus
}
#endif
#endif /* BNPC_HIP_H */
