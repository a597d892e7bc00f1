// bnpc_kernels.hip - gfx950 (MI355X, CDNA4) kernels and the device half of the
// C-ABI declared in include/bnpc_hip.h.
//
// Design (DESIGN.md sections 3-4), written for wave64 / CDNA4, no MFMA:
//
//   * The data matrix lives in HBM twice, as bit planes:
//       rows [N][W]      {ones, zeros} 64-bit words, cell-major (gather source)
//       masks[blk][m]    {ones, zeros} 64-bit LANE MASKS over 64 cells
//     A mask word is exactly an EXEC mask: lane s of the wave owns cell slot
//     64*blk+s, so "add L1[k][m] to the accumulators of all cells that observed
//     a 1 at mutation m" is ONE exec-masked v_add_f64 whose mask comes straight
//     from memory (s_load -> s_and_saveexec) and whose addend is an SGPR pair
//     (the table element is wave-uniform).  No per-lane bit tests, no selects,
//     no cross-lane reduction.
//   * Each lane accumulates its cell's sum over mutations IN INDEX ORDER, so
//     the result is the reference's strictly sequential bn.nansum
//     (libs/CRP.py:204) bit for bit once the table elements are equal.
//   * The per-element logs are hoisted into tables T[g][m][2*KW] (2*K*M logs
//     instead of N*K*M), laid out so that the 2*KW doubles a wave needs per
//     mutation are one contiguous run for s_load_dwordx16.
//   * FP64 throughout (accumulators, tables, outputs).
//
// Reference expressions are cited at each kernel (paths relative to
// /root/reference).

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <thread>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <sys/mman.h>
#include <time.h>
#include <mutex>
#include <vector>
#include <algorithm>

#include "bnpc_hip.h"
#include "bnpc_internal.h"

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void bnpc_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *bnpc_last_error(void) { return g_err; }
extern "C" int bnpc_abi_version(void) { return 12; }

#define HIPCHK(expr)                                                         \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                              \
            bnpc_set_error("%s failed: %s (%s:%d)", #expr,                   \
                           hipGetErrorString(e_), __FILE__, __LINE__);       \
            return 1;                                                        \
        }                                                                    \
    } while (0)

#define ARGCHK(cond, msg)                                                    \
    do {                                                                     \
        if (!(cond)) {                                                       \
            bnpc_set_error("bad argument: %s", msg);                         \
            return 2;                                                        \
        }                                                                    \
    } while (0)

// ---------------------------------------------------------------------------
// per-launch device timers (bnpc_launch_timers: bench.py's
// window.device_ms_per_step).  Off: a plain launch.  On: the launch carries a
// start / stop event pair that takes the dispatch's own begin / end
// timestamps (what rocprofv3's kernel trace reads), summed when the timers
// are read.  One chain per process launches from one thread (the caller, or
// rank 0 of a team job = the caller): no lock.
// ---------------------------------------------------------------------------
struct LaunchTimers {
    bool on = false;
    std::vector<hipEvent_t> ev;     // pairs
    size_t used = 0;
    hipEvent_t *next_pair()
    {
        if (used + 2 > ev.size()) {
            const size_t old = ev.size();
            ev.resize(old + 512, nullptr);
            for (size_t i = old; i < ev.size(); i++)
                if (hipEventCreate(&ev[i]) != hipSuccess) {
                    ev.resize(i & ~(size_t)1);
                    break;
                }
            if (used + 2 > ev.size()) return nullptr;
        }
        used += 2;
        return &ev[used - 2];
    }
};
static LaunchTimers g_timers;

#define BNPC_LAUNCH(kern, grid, block, lds, stream, ...)                      \
    do {                                                                      \
        hipEvent_t *e_ = g_timers.on ? g_timers.next_pair() : nullptr;        \
        if (e_)                                                               \
            hipExtLaunchKernelGGL(kern, grid, block, lds, stream, e_[0],      \
                                  e_[1], 0, __VA_ARGS__);                     \
        else                                                                  \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);  \
    } while (0)

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct View {
    DevBuf masks;       // ulonglong2 [nblk][Mpad]
    int64_t n = 0;      // slots in use
    int64_t nblk = 0;
};

// Switches, read from the environment when a context is created and again
// by bnpc_reload_options (tests); never on the launch path.  README lists
// them; the tuning constants that used to be switches (chunking of split
// launches, zero-copy sizes, the screen's minimum batch ...) are the measured
// values below.
struct Tunables {
    int msplit = 1;                 // BNPC_MSPLIT: mutation-split small launches
    int force_kw = 0;               // BNPC_KW: force the cluster tile (tests)
    int zero_copy = 1;              // BNPC_ZERO_COPY: small payloads are read /
                                    // written in place in pinned host memory
    int mask_counts_max = 64;       // BNPC_MASK_COUNTS_MAX: segments for the
                                    // mask-popcount counts (tests lower it)
    int mh_screen = 1;              // BNPC_MH_SCREEN: device screen of the
                                    // parameter batches
    int done_words = 1;             // BNPC_DONE_WORDS: completion words written
                                    // by the kernels (0: stream synchronisation)
    int msplit_chunks = 0;          // BNPC_MSPLIT = N >= 2: force the chunk count
                                    // of split launches (tools/msplit_sweep.py)
    int screen_theta = 1;           // BNPC_MH_SCREEN = 2: verdicts only - not the
                                    // float32 bits of the proposals it accepts
    int mh_ahead = 1;               // BNPC_MH_AHEAD: the draws of the next
                                    // parameter batch taken ahead on the aside
                                    // thread (0: never; 2: for a batch of any
                                    // size - tests; 3: taken and then thrown
                                    // away - tests of the discard path)
    size_t mh_pin_max = (size_t)512 << 20;  // pinned block of a screened
                                    // parameter batch at most: twice
                                    // BNPC_SWEEP_BYTES, the host budget of a
                                    // sweep's matrix (default 256 MiB -> 512:
                                    // 37 bytes per entry, 14.5 M entries -
                                    // config 4's K0 x M batch fits, config 5's
                                    // 158 M are screened in slices of rows
                                    // that reuse the block)
};

#define MSPLIT_MAX 64               // chunks of a split launch at most
#define ASM2_MIN_WGS 448            // workgroups from which a wave takes 2 blocks
#define TABLES_FLAT_MAX (1 << 20)   // table elements up to which one thread
                                    // builds one element
#define ZC_IN_MAX ((int64_t)256 << 10)      // zero-copy inputs / results up to
#define ZC_OUT_MAX ((int64_t)512 << 10)
#define MH_SCREEN_MIN 512           // batch entries from which the screen pays
#define MH_THREADED_MIN 65536       // batch entries from which rank 0 issues
                                    // draws and launches ahead of the waits
                                    // (the pinned block of a screened batch
                                    // is at most Tunables::mh_pin_max bytes)
#define MH_PIN_NO_MEMORY 77         // mh_pin_get: the host refused the block
#define MH_AHEAD_MIN 8192           // batch entries from which its draws are
                                    // taken ahead (config 3's 10-16 thousand:
                                    // parameters 0.102 -> 0.090 ms, five
                                    // interleaved pairs, profiles/r06/
                                    // c3_walker_ab; config 2's 2000 cost less
                                    // than the hand-over)
#define MH_AHEAD_SCAN_MIN 2048      // ... of a restricted scan's batch (2-3 rows:
                                    // the walker has the scan's sums and loop,
                                    // 50 us and more, for 10-35 us of draws)
#define MH_AHEAD_MAX_ROWS 1024      // ... and rows up to which (a stream state
                                    // is kept per row: 2.5 KB)
#define HINT_COLS_MAX 32767         // columns of a hinted sweep (int16 in the
                                    // record)
#define HINT_THROUGH_MAX 1024       // ... up to which rows that will be scanned
                                    // are written through to the host
#define LDS_TABLE_MIN_M 3072        // k_ll8_lds: mutations (padded) from which,
#define LDS_TABLE_MIN_WGS 4096      // ... and workgroups from which it wins

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

static void read_tunables(Tunables &t)
{
    t.msplit = env_int("BNPC_MSPLIT", 1);
    t.force_kw = env_int("BNPC_KW", 0);
    if (t.force_kw != 1 && t.force_kw != 2 && t.force_kw != 4
        && t.force_kw != 8)
        t.force_kw = 0;
    t.zero_copy = env_int("BNPC_ZERO_COPY", 1);
    t.mask_counts_max = env_int("BNPC_MASK_COUNTS_MAX", 64);
    t.mh_screen = env_int("BNPC_MH_SCREEN", 1);
    t.done_words = env_int("BNPC_DONE_WORDS", 1);
    t.screen_theta = t.mh_screen != 2;
    t.mh_ahead = env_int("BNPC_MH_AHEAD", 1);
    {
        const char *e = getenv("BNPC_SWEEP_BYTES");
        const long long b = e ? atoll(e) : 0;
        t.mh_pin_max = 2 * (size_t)(b > 0 ? b : (long long)256 << 20);
    }
    t.msplit_chunks = t.msplit >= 2 ? t.msplit : 0;
}

#define DONE_SLOTS 3     // 0, 1: the launches of a call; 2: the deferred total
// what a kernel needs to tell the host that it is done (signal_done below)
struct DoneSignal {
    unsigned *count;    // device word, zero between launches
    unsigned *flag;     // pinned host word (device address)
    unsigned seq;
};

// the column priors of a hint launch of up to 64 columns (kernel argument)
struct Top2Prior {
    double v[64];
};

struct bnpc_ctx {
    Tunables tun;
    int device = 0;
    int64_t N = 0, M = 0;
    int W = 0;          // 64-bit words per row
    int Mpad = 0;       // W * 64
    int Mt = 0;         // table row count per group: M rounded up to 8
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ulonglong2 *rows = nullptr;           // [N][W]
    std::vector<ulonglong2> host_rows;    // the same words on the host: the
                                          // observations of ONE cell shape the
                                          // Beta draws of a cluster it opens
    std::vector<int32_t> cell_n1, cell_n0;
    View views[BNPC_MAX_VIEWS];
    // scratch
    DevBuf theta, tabs, tab_in, out, cells, chunks, cnt, partial, part;
    DevBuf theta_store, row_idx;    // resident parameter rows + selection
    int64_t store_rows = 0;
    const long long *use_rows = nullptr;    // non-null: tables from the store
    // resident per-cluster counts of the last bnpc_colcounts_by_label
    DevBuf lab_cnt;
    int64_t lab_K = 0;
    uint64_t lab_gen = 0;       // bumped by every bnpc_colcounts_by_label
    int64_t cnt_rows = 0;       // segments of the last bnpc_view_counts (c->cnt)
    // pinned host buffers: the sweep's ll matrix / small reductions
    void *pin = nullptr;
    size_t pin_cap = 0;
    void *pin_small = nullptr;
    // side lane: a second stream with its own scratch, used by the small
    // synchronous calls (one column for a cluster opened mid-sweep) while an
    // issued tile occupies the main stream - they must not queue behind it
    hipStream_t side_stream = nullptr;
    DevBuf side_theta, side_tabs, side_out, side_part;
    // pinned staging arena for small host <-> device payloads (parameter
    // rows, cell lists, counts): a copy from/to pinned memory is a plain DMA
    // enqueue, a copy from/to pageable memory is staged by the runtime at
    // ~10 us apiece.  Reset at the start of every call that uses it; every
    // such call ends with a stream synchronisation.
    void *stage = nullptr;
    char *stage_dev = nullptr;      // the arena as the device addresses it
    size_t stage_used = 0;
    // small results written by kernels straight into pinned host memory
    void *zc_out = nullptr;
    char *zc_out_dev = nullptr;
    void *hint_pin = nullptr;       // the sweep's per-cell hints (pinned)
    size_t hint_cap = 0;
    DevBuf hint_prior;              // priors of a hinted sweep with > 64 columns
    void *hint_prior_pin = nullptr; // ... staged here (pinned, HINT_COLS_MAX)
    // bnpc_ll_theta_pinned_sums_issue: a hinted sweep whose hint kernel is
    // launched later (bnpc_hints_in_order_issue), when the caller has drawn
    // its visiting order under the sums - what that launch needs
    struct {
        // 0: no sums issued; 1: issued, the hint kernel is to be launched;
        // 2: issued without a hint buffer (the matrix was copied instead)
        int state = 0;
        int64_t n = 0, K = 0, ldo = 0;
        size_t bytes = 0;
        Top2Prior prior;            // K <= 64 (more: c->hint_prior)
        void *hint_dev = nullptr;
        double *rows_dev = nullptr;
    } hint_later;
    void *order_pin = nullptr;      // the visiting order, pinned (N entries)
    DevBuf order_dev;               // ... and on the device
    // pinned block of a screened parameter batch (bnpc_mh_batch_dev): the
    // draws, the old parameter rows and the screen's verdicts, read / written
    // in place by k_mh_screen
    void *mh_pin = nullptr;
    char *mh_dev = nullptr;
    size_t mh_cap = 0;
    size_t mh_capE = 0;             // entries the block is laid out for
    struct MhAhead *ahead = nullptr;    // draws taken ahead (bnpc_mh_ahead_*)
    int64_t ahead_begun = 0, ahead_taken = 0, ahead_rows_taken = 0;
    double mh_flagged_share = 0.25; // host work the last screened batch left,
                                    // per entry (sizes the next one's team)
    hipEvent_t mh_ev[2] = {};
    int64_t screened = 0, screen_kept = 0;  // elements seen / left to the host
    size_t pin_lazy_bytes = 0;      // sweep matrix still on the device (c->out)
    // ... unless the previous hinted sweep had to fetch it: then the copy is
    // queued right behind the hint kernel and lands while the host prepares
    // the sweep (a running chain scans ~9 % of its cells: it always needs it;
    // a settled one never does)
    bool matrix_eager = false, lazy_fetched = false, pin_copy_queued = false;
    hipEvent_t ev_hints = nullptr;
    // completion words (DoneSignal): two slots, so that two launches of one
    // call may be in flight (the two halves of a screened batch)
    unsigned *done_count = nullptr;         // device, DONE_SLOTS words
    unsigned *done_pin = nullptr;           // pinned host, DONE_SLOTS x 16 words
    unsigned *done_dev = nullptr;           // ... as the device addresses it
    unsigned done_seq = 0;
    unsigned total_seq = 0;                 // of the pending bnpc_ll_total
    int total_slot = -1;
    DoneSignal sig_next = {nullptr, nullptr, 0};    // for the last kernel of
    bool sig_attached = false;                      // the next issue_ll
    // bnpc_ll_theta_begin / _end: an evaluation whose result is written in
    // place for the host and picked up later (the caller works in between)
    bool defer_next = false, defer_set = false;
    struct {
        void *zc_host;
        unsigned seq;
        double *out;
        size_t bytes;
        int64_t n, K, ldo;
    } defer = {nullptr, 0, nullptr, 0, 0, 0, 0};
    // bnpc_view_set's own pinned cell list (N entries) and the event that
    // says the last gather has read it
    void *view_cells_pin = nullptr;
    const long long *view_cells_dev = nullptr;
    hipEvent_t view_cells_read = nullptr;
    bool view_cells_busy = false;
    bool total_pending = false;     // a deferred bnpc_ll_total_issue
    int total_blocks = 0, total_E = 0;
    // where the kernels of the current call read their inputs from: device
    // scratch filled by a DMA copy, or the staging arena in place
    const float *theta_src = nullptr;
    const double *tab_src = nullptr;
    const long long *cells_src = nullptr;
    // Issued (asynchronous) tiles: up to BNPC_TILE_SLOTS in flight, each with
    // its own pinned result buffer.  The sums of consecutive tiles alternate
    // between two device buffers and the copy to the host runs on its own
    // stream, so the copy of one tile overlaps the sums of the next.
    void *tile_pin[BNPC_TILE_SLOTS] = {};
    size_t tile_cap[BNPC_TILE_SLOTS] = {};
    size_t tile_bytes[BNPC_TILE_SLOTS] = {};
    void *tile_rows[BNPC_TILE_SLOTS] = {};      // pinned staging of the ids
    size_t tile_rows_cap[BNPC_TILE_SLOTS] = {};
    void *tile_cells[BNPC_TILE_SLOTS] = {};     // ... and of the tile's cells
    size_t tile_cells_cap[BNPC_TILE_SLOTS] = {};
    void *tile_hint[BNPC_TILE_SLOTS] = {};      // pinned: the tile's hints
    size_t tile_hint_cap[BNPC_TILE_SLOTS] = {};
    void *tile_prior[BNPC_TILE_SLOTS] = {};     // pinned staging of the priors
    size_t tile_prior_cap[BNPC_TILE_SLOTS] = {};
    bool tile_hinted[BNPC_TILE_SLOTS] = {};
    DevBuf tile_prior_dev[2];                   // by parity, like tile_out
    hipEvent_t tile_done[BNPC_TILE_SLOTS] = {}; // copy landed in tile_pin
    bool tile_pending[BNPC_TILE_SLOTS] = {};
    DevBuf tile_out[2];                         // by parity of the issue count
    hipEvent_t tile_summed[2] = {};             // sums written to tile_out
    hipEvent_t tile_out_free[2] = {};           // its last copy has left
    uint64_t tile_seq = 0;
    hipStream_t copy_stream = nullptr;
    bool any_tile_pending() const
    {
        for (bool p : tile_pending)
            if (p) return true;
        return false;
    }
    // configuration of the last k_ll launch (bnpc_bench_ll re-issues it)
    int last_kw = 0, last_view = -1, last_ms = 1, last_mchunk = 0;
    int64_t last_K = 0, last_ldo = 0;
    double *last_out = nullptr;
    double *dst_override = nullptr; // device-addressable result buffer
    bool last_from_theta = false;
    double last_FP = 0.0, last_FN = 0.0;
    char last_name[96] = "";
};

static void mh_ahead_destroy(bnpc_ctx *c);      // (with MhAhead, below)

static int ensure(DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    if (b.p) HIPCHK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t cap = bytes + bytes / 4 + 256;
    HIPCHK(hipMalloc(&b.p, cap));
    b.cap = cap;
    return 0;
}

// Large pinned host buffers (result matrices, tiles: hundreds of MiB).
// hipHostMalloc pins 4 KiB pages - 45-48 ms per 300 MiB on the MI355X host,
// and a first sweep needs two or three of them.  Anonymous memory on
// transparent huge pages, touched and then registered, costs 17 + 1 ms for the
// same size and is the same DMA target (57 GB/s either way;
// tools/ubench/pin_probe.hip).  Falls back to hipHostMalloc when huge pages
// are switched off (4 KiB pages would make this route the slower one) or
// anything fails.  `cap` identifies the route at release time: huge-page
// buffers have a capacity that is a multiple of 2 MiB and are remembered.
#define PIN_HUGE_MIN ((size_t)16 << 20)
#define PIN_HUGE_ALIGN ((size_t)2 << 20)

static bool thp_available()
{
    static const bool ok = [] {
        FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
        if (!f) return false;
        char line[128] = {0};
        const bool got = fgets(line, sizeof line, f) != nullptr;
        fclose(f);
        return got && !strstr(line, "[never]");
    }();
    return ok;
}

static std::vector<void *> &huge_pins()
{
    static std::vector<void *> v;
    return v;
}

static std::mutex &huge_pins_lock()         // contexts may live on threads
{
    static std::mutex m;
    return m;
}

static int pinned_alloc(void **out, size_t *cap, size_t bytes)
{
    *out = nullptr;
    *cap = 0;
    if (bytes >= PIN_HUGE_MIN && thp_available()) {
        const size_t len = (bytes + PIN_HUGE_ALIGN - 1) & ~(PIN_HUGE_ALIGN - 1);
        char *raw = (char *)mmap(nullptr, len + PIN_HUGE_ALIGN,
                                 PROT_READ | PROT_WRITE,
                                 MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (raw != (char *)MAP_FAILED) {
            char *p = (char *)(((uintptr_t)raw + PIN_HUGE_ALIGN - 1)
                               & ~(uintptr_t)(PIN_HUGE_ALIGN - 1));
            if (p > raw) munmap(raw, (size_t)(p - raw));
            const size_t tail = (size_t)(raw + len + PIN_HUGE_ALIGN - (p + len));
            if (tail) munmap(p + len, tail);
            (void)madvise(p, len, MADV_HUGEPAGE);
            // fault the huge pages in before pinning: one touch per 2 MiB
            // (the kernel zeroes them; anything left on small pages is
            // faulted in by the registration itself)
            for (size_t off = 0; off < len; off += PIN_HUGE_ALIGN)
                ((volatile char *)p)[off] = 0;
            if (hipHostRegister(p, len, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> hold(huge_pins_lock());
                huge_pins().push_back(p);
                *out = p;
                *cap = len;
                return 0;
            }
            (void)hipGetLastError();
            munmap(p, len);
        }
    }
    HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
    *cap = bytes;
    return 0;
}

static void pinned_free(void *p, size_t cap)
{
    if (!p) return;
    bool huge = false;
    {
        std::lock_guard<std::mutex> hold(huge_pins_lock());
        std::vector<void *> &v = huge_pins();
        auto it = std::find(v.begin(), v.end(), p);
        if (it != v.end()) {
            v.erase(it);
            huge = true;
        }
    }
    if (huge) {
        (void)hipHostUnregister(p);
        munmap(p, cap);
    } else {
        (void)hipHostFree(p);
    }
}

#define STAGE_BYTES ((size_t)4 << 20)
#define ZC_OUT_BYTES ((size_t)1 << 20)

// a slot of the staging arena, or nullptr when the payload is too large
static void *stage_slot(bnpc_ctx *c, size_t bytes)
{
    if (!c->stage) {
        if (hipHostMalloc(&c->stage, STAGE_BYTES, hipHostMallocDefault)
                != hipSuccess) {
            c->stage = nullptr;
            return nullptr;
        }
        void *dev = nullptr;
        if (hipHostGetDevicePointer(&dev, c->stage, 0) == hipSuccess)
            c->stage_dev = (char *)dev;
    }
    const size_t at = (c->stage_used + 255) & ~(size_t)255;
    if (at + bytes > STAGE_BYTES) return nullptr;
    c->stage_used = at + bytes;
    return (char *)c->stage + at;
}

// Start of a call that stages inputs: the arena is free again - unless a
// deferred total (bnpc_ll_total_issue) may still be reading its parameters
// from it; then that kernel is waited for first (its result stays parked).
static int arena_reset(bnpc_ctx *c)
{
    if (c->total_pending) HIPCHK(hipStreamSynchronize(c->stream));
    c->stage_used = 0;
    return 0;
}

// A DoneSignal for the next launch on slot 0 / 1 (the words are made on first
// use; without them - or with BNPC_DONE_WORDS=0 - the signal is empty and the
// caller synchronises as before).  *seq receives the number to wait for.
static DoneSignal make_signal(bnpc_ctx *c, int slot, unsigned *seq)
{
    DoneSignal none = {nullptr, nullptr, 0};
    *seq = 0;
    if (!c->tun.done_words) return none;
    if (!c->done_count) {
        void *pin = nullptr, *dev = nullptr, *cnt = nullptr;
        if (hipHostMalloc(&pin, DONE_SLOTS * 64, hipHostMallocDefault) != hipSuccess
            || hipHostGetDevicePointer(&dev, pin, 0) != hipSuccess
            || hipMalloc(&cnt, DONE_SLOTS * sizeof(unsigned)) != hipSuccess
            || hipMemset(cnt, 0, DONE_SLOTS * sizeof(unsigned))
                != hipSuccess) {
            (void)hipGetLastError();
            if (pin) (void)hipHostFree(pin);
            if (cnt) (void)hipFree(cnt);
            return none;
        }
        memset(pin, 0, DONE_SLOTS * 64);
        c->done_pin = (unsigned *)pin;
        c->done_dev = (unsigned *)dev;
        c->done_count = (unsigned *)cnt;
    }
    if (++c->done_seq == 0) c->done_seq = 1;    // 0 = "no signal"
    *seq = c->done_seq;
    DoneSignal d = {c->done_count + slot, c->done_dev + 16 * slot, *seq};
    return d;
}

// Wait for the word of a signalled launch; seq == 0 (no signal was attached)
// or a word that does not come within the spin: hipStreamSynchronize.
static int wait_done(bnpc_ctx *c, int slot, unsigned seq)
{
    if (seq) {
        const volatile unsigned *f = c->done_pin + 16 * slot;
        for (int spins = 0; spins < 20000; spins++) {       // ~100-200 us
            // (launches of a stream finish in order and the numbers only
            // grow: a later number on the word says this one is done too)
            if ((int)(*f - seq) >= 0) {
                std::atomic_thread_fence(std::memory_order_acquire);
                return 0;
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// Zero-copy input: the payload is copied into the pinned arena and the
// kernels of this call read it there, over the host link, instead of from a
// device buffer filled by a DMA copy - for payloads of a few hundred KiB a
// copy engine launch costs more than the bytes.  Returns the DEVICE address,
// or nullptr (too large / switched off): then the caller copies.  Every call
// that uses the arena ends with a stream synchronisation.
static const void *stage_in_place(bnpc_ctx *c, const void *src, size_t bytes)
{
    if (!c->tun.zero_copy || (int64_t)bytes > ZC_IN_MAX) return nullptr;
    void *slot = stage_slot(c, bytes);
    if (!slot || !c->stage_dev) return nullptr;
    memcpy(slot, src, bytes);
    return c->stage_dev + ((char *)slot - (char *)c->stage);
}

// Zero-copy output: `bytes` of pinned host memory the kernels of this call may
// write their (small) result to; *dev receives the device address.  nullptr:
// not available for this size.
static void *zc_result(bnpc_ctx *c, size_t bytes, void **dev)
{
    if (!c->tun.zero_copy || (int64_t)bytes > ZC_OUT_MAX
        || bytes > ZC_OUT_BYTES)
        return nullptr;
    if (!c->zc_out) {
        if (hipHostMalloc(&c->zc_out, ZC_OUT_BYTES, hipHostMallocDefault)
                != hipSuccess) {
            c->zc_out = nullptr;
            return nullptr;
        }
        void *d = nullptr;
        if (hipHostGetDevicePointer(&d, c->zc_out, 0) != hipSuccess) {
            (void)hipHostFree(c->zc_out);
            c->zc_out = nullptr;
            return nullptr;
        }
        c->zc_out_dev = (char *)d;
    }
    *dev = c->zc_out_dev;
    return c->zc_out;
}

// host -> device on the context's stream; `src` may be released on return
// only if the caller synchronises the stream before it returns itself
static int h2d(bnpc_ctx *c, void *dst, const void *src, size_t bytes)
{
    void *slot = stage_slot(c, bytes);
    if (slot) {
        memcpy(slot, src, bytes);
        src = slot;
    }
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}

// device -> host, completed by the caller's stream synchronisation followed
// by d2h_finish (which moves the staged bytes to their destination)
struct D2H {
    void *dst, *slot;
    size_t bytes;
};

static int d2h_begin(bnpc_ctx *c, D2H &t, void *dst, const void *src,
                     size_t bytes)
{
    t.dst = dst;
    t.bytes = bytes;
    t.slot = stage_slot(c, bytes);
    HIPCHK(hipMemcpyAsync(t.slot ? t.slot : dst, src, bytes,
                          hipMemcpyDeviceToHost, c->stream));
    return 0;
}

static void d2h_finish(const D2H &t)
{
    if (t.slot) memcpy(t.dst, t.slot, t.bytes);
}

static int ensure_pin(bnpc_ctx *c, size_t bytes)
{
    if (c->pin_copy_queued) {   // a queued copy still targets the buffer
        HIPCHK(hipStreamSynchronize(c->stream));
        c->pin_copy_queued = false;
    }
    c->pin_lazy_bytes = 0;      // a new request supersedes a matrix not fetched
    if (bytes <= c->pin_cap) return 0;
    pinned_free(c->pin, c->pin_cap);
    c->pin = nullptr;
    c->pin_cap = 0;
    return pinned_alloc(&c->pin, &c->pin_cap, bytes + bytes / 4 + 4096);
}

// While a tile is in flight, run a call on the side lane: swap the stream and
// the scratch buffers the likelihood path uses, restore on scope exit.
struct SideLane {
    bnpc_ctx *c;
    bool on;
    explicit SideLane(bnpc_ctx *ctx)
        : c(ctx), on(ctx->side_stream && ctx->any_tile_pending())
    {
        if (on) flip();
    }
    ~SideLane()
    {
        if (on) flip();
    }
    void flip()
    {
        std::swap(c->stream, c->side_stream);
        std::swap(c->theta, c->side_theta);
        std::swap(c->tabs, c->side_tabs);
        std::swap(c->out, c->side_out);
        std::swap(c->part, c->side_part);
    }
};

// the side lane, created at the first tile of a context: calls made while
// tiles are in flight (a column for a cluster just opened, the columns of
// clusters born since a tile was issued) run beside 10 ms kernels that fill
// the chip; on a stream of the highest priority their few workgroups get the
// next free slots instead of waiting for a whole tile.  (Compute units of
// their own - the tile kernels on a CU-masked stream, the side lane on the
// rest - were tried in round 4: the masked stream ran the whole sweep 12 %
// slower, 0.53 against 0.475 s, for births that are bound by their host-side
// Beta draws anyway.)
static int ensure_lanes(bnpc_ctx *c)
{
    if (c->side_stream) return 0;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess
        || hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking,
                                       greatest) != hipSuccess) {
        (void)hipGetLastError();        // no priorities here: a plain stream
        c->side_stream = nullptr;
        HIPCHK(hipStreamCreateWithFlags(&c->side_stream,
                                        hipStreamNonBlocking));
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Completion words.  A converged step is a chain of small dependent launches
// whose results the host waits for (~7 waits per step), and the runtime's
// completion signal reaches the host 11-14 us after a launch began - 5-8 us
// after a word the kernel writes itself (tools/ubench/sync_probe.hip: 5.3
// us).  So the kernels whose results the host waits for take a DoneSignal:
// every workgroup, its stores complete (the barrier), has one thread publish
// them at system scope and count itself in; the workgroup that counts in
// last resets the counter and writes the launch's sequence number to a word
// in pinned host memory, which the host polls (wait_done) - falling back to
// hipStreamSynchronize after a bounded spin, so a lost word costs time, not
// correctness.  count == NULL: no signal (the caller synchronises).
// ---------------------------------------------------------------------------
// EVERY thread of the workgroup must get here (no early return before it)
__device__ __forceinline__ void signal_done(const DoneSignal &d)
{
    if (!d.count) return;
    __syncthreads();                // each wave's stores are issued and acked
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
        __threadfence_system();     // ... and visible to the host
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned n = gridDim.x * gridDim.y * gridDim.z;
        const unsigned old = __hip_atomic_fetch_add(
            d.count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == n - 1) {
            __hip_atomic_store(d.count, 0u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------
// K1: gather rows of a cell list and transpose 64x64 bit tiles into lane masks
//   replaces `self.data[cells]` (libs/CRP.py:360, 557-560, 636-637, 726-728)
// one wave per (slot block, row word); lane s loads word w of its cell, then
// 64 ballots turn bit b of all 64 lanes into the lane mask of mutation 64w+b,
// which lane b keeps and stores (coalesced 1 KiB per wave).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_transpose(
    const ulonglong2 *__restrict__ rows, const long long *__restrict__ cells,
    long long n, int W, ulonglong2 *__restrict__ masks, int Mpad)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (w >= W) return;
    const long long blk = blockIdx.x;
    const long long slot = blk * 64 + lane;
    long long cell = -1;
    if (slot < n) cell = cells ? cells[slot] : slot;
    ulonglong2 r = make_ulonglong2(0ull, 0ull);
    if (cell >= 0) r = rows[(size_t)cell * W + w];
    ulonglong2 mine = make_ulonglong2(0ull, 0ull);
#pragma unroll
    for (int b = 0; b < 64; b++) {
        unsigned long long mo = __ballot((r.x >> b) & 1ull);
        unsigned long long mz = __ballot((r.y >> b) & 1ull);
        if (lane == b) {
            mine.x = mo;
            mine.y = mz;
        }
    }
    masks[(size_t)blk * Mpad + (size_t)w * 64 + lane] = mine;
}

// ---------------------------------------------------------------------------
// K4a: element tables from float32 parameters
//   L1 = log(theta*(1-FN) + (1-theta)*FP)      value of an observed 1
//   L0 = log(theta*FN     + (1-theta)*(1-FP))  value of an observed 0
//   = the argument of np.log in CRP._calc_ll (libs/CRP.py:198-200) with
//     _Bernoulli_FN/_FP (libs/CRP.py:207-212) evaluated at x = 1 and x = 0;
//     theta is float32, (1 - theta) is float32 arithmetic, products and sum
//     float64, each rounded separately (compiled with -ffp-contract=off).
// layout T[g][m][2*KW]: [0,KW) = L1 of clusters g*KW.., [KW,2KW) = L0.
// ---------------------------------------------------------------------------
template <int KW>
__global__ __launch_bounds__(256) void k_tables_theta(
    const float *__restrict__ theta, const long long *__restrict__ rows,
    int K, int M, int Mt, double FP, double FN, double *__restrict__ T)
{
    // One thread per table ELEMENT: consecutive threads write consecutive
    // doubles of T[g][m][0 .. 2 KW) (a thread per mutation that loops over
    // the group's clusters stores 8 bytes per lane at a 128-byte stride: 2.6
    // ms per 2.5 GB of tables at config 5, about a quarter of what the
    // memory system takes).  Same expression per element, same values.
    constexpr int E = 2 * KW;                   // elements per mutation
    constexpr int MB = 256 / E;                 // mutations per workgroup
    const int e = threadIdx.x % E;
    const int m = blockIdx.x * MB + threadIdx.x / E;
    const int g = blockIdx.y;
    if (m >= Mt) return;
    const int j = e < KW ? e : e - KW;
    const int k = g * KW + j;
    double v = 0.0;
    if (k < K && m < M) {
        const float th = theta[(size_t)(rows ? rows[k] : k) * M + m];
        const double th64 = (double)th;
        const double om64 = (double)(1.0f - th);
        // (1-FN)**1 * FN**0 and (1-FP)**1 * FP**0 of the reference
        v = e < KW ? log(th64 * (1.0 - FN) + om64 * FP)
                   : log(th64 * FN + om64 * (1.0 - FP));
    }
    T[((size_t)g * Mt + m) * E + e] = v;
}

// K4a for small launches: one thread per (cluster, mutation) instead of one
// per mutation with a loop over the group's clusters - K x Mt threads rather
// than G x Mt, so a converged K ~ 10 still fills some of the chip.  Same
// expression, same values.
template <int KW>
__global__ __launch_bounds__(256) void k_tables_theta_flat(
    const float *__restrict__ theta, const long long *__restrict__ rows,
    int K, int M, int Mt, int G, double FP, double FN, double *__restrict__ T)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)G * KW * Mt) return;
    const int kk = (int)(idx / Mt);             // padded cluster index
    const int m = (int)(idx - (long long)kk * Mt);
    const int g = kk / KW, j = kk - g * KW;
    double l1 = 0.0, l0 = 0.0;
    if (kk < K && m < M) {
        const float th = theta[(size_t)(rows ? rows[kk] : kk) * M + m];
        const double th64 = (double)th;
        const double om64 = (double)(1.0f - th);
        l1 = log(th64 * (1.0 - FN) + om64 * FP);
        l0 = log(th64 * FN + om64 * (1.0 - FP));
    }
    double *t = T + ((size_t)g * Mt + m) * (2 * KW);
    t[j] = l1;
    t[KW + j] = l0;
}

// K4b: re-layout caller-built tables L1/L0 [K][M] into T[g][m][2*KW]
template <int KW>
__global__ __launch_bounds__(256) void k_tables_relayout(
    const double *__restrict__ L1, const double *__restrict__ L0, int K, int M,
    int Mt, double *__restrict__ T)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (m >= Mt) return;
    double *t = T + ((size_t)g * Mt + m) * (2 * KW);
#pragma unroll
    for (int j = 0; j < KW; j++) {
        const int k = g * KW + j;
        const bool live = (k < K) && (m < M);
        t[j] = live ? L1[(size_t)k * M + m] : 0.0;
        t[KW + j] = live ? L0[(size_t)k * M + m] : 0.0;
    }
}

// ---------------------------------------------------------------------------
// K2: the cells x clusters x mutations op
//   out[s][k] = sum_m [x_sm = 1] L1[k][m] + [x_sm = 0] L0[k][m]
//   = CRP._calc_ll(data[[cell]], parameters[cl_ids]) for every cell at once
//     (libs/CRP.py:197-204, called from get_lpost_single :223-227 in the
//     Gibbs sweep :270 and from _rg_get_ll :635-638)
// lane <-> cell slot, wave <-> (64-slot block, group of KW clusters);
// masks and table elements are wave-uniform -> scalar loads; the adds are
// exec-masked v_add_f64 with an SGPR-pair addend; m runs sequentially.
// ---------------------------------------------------------------------------
// Software pipeline: a stage is U consecutive mutations (U mask pairs and
// U*2*KW table doubles, all in SGPRs); the scalar loads of the next stage are
// in flight while the exec-masked adds of the current one issue.  The table
// stride Mt is M rounded up to 8 with zero padding and the mask rows are
// padded to 64 mutations with empty masks, so no tail handling is needed; the
// prefetch of the stage after the last one reads the slack the allocations
// keep.
//
// Grid: 1-D, one workgroup = 4 waves = 4 consecutive slot blocks x one
// cluster group.  Workgroups are dealt round-robin over the 8 XCDs (observed,
// MI355X_MICROARCH.md), each with its own L2: the XCD-aware remap below gives
// every XCD a CONTIGUOUS range of virtual ids, and virtual ids enumerate the
// slot blocks of one cluster group before moving to the next group, so the
// workgroups that stream the same table run on one XCD and share it in that
// L2 (speed only; any placement is correct).
template <int KW>
struct LLStage {
    static constexpr int U = (KW >= 8) ? 1 : (KW >= 2 ? 2 : 4);
};

__device__ __forceinline__ void ll_tile_coords(unsigned nbx, unsigned G,
                                               int xcd_remap, long long &bx,
                                               long long &g)
{
    const unsigned nwg = nbx * G;           // < 2^31, checked by the host
    unsigned v = blockIdx.x;
    if (xcd_remap) {
        const unsigned q = nwg >> 3, r = nwg & 7;
        const unsigned xcd = v & 7, idx = v >> 3;
        v = xcd * q + (xcd < r ? xcd : r) + idx;
    }
    const unsigned gg = v / nbx;
    g = gg;
    bx = v - gg * nbx;
}

// MS > 1 splits the mutations of a (slot block, cluster group) over MS waves
// (small launches are otherwise one long dependent chain of scalar loads per
// wave): wave ms sums mutations [ms*m_chunk, (ms+1)*m_chunk) into
// part[ms][slot][k]; k_ll_combine adds the MS partial sums in index order.
template <int KW>
__global__ __launch_bounds__(256) void k_ll(
    const ulonglong2 *__restrict__ masks, int Mpad, int Mt, long long n,
    long long nblk, const double *__restrict__ T, int K, long long ldo,
    double *__restrict__ out, int xcd_remap, int MS, int m_chunk)
{
    constexpr int U = LLStage<KW>::U;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    long long bx, g;
    ll_tile_coords((unsigned)((nblk + 3) >> 2) * (unsigned)MS,
                   (unsigned)((K + KW - 1) / KW), xcd_remap, bx, g);
    const int ms = (int)(bx % MS);
    bx /= MS;
    const long long blk = bx * 4 + wave;
    if (blk >= nblk) return;     // whole wave leaves together
    const int m_begin = ms * m_chunk;
    const int m_end = (m_begin + m_chunk < Mt) ? m_begin + m_chunk : Mt;

    const ulonglong2 *__restrict__ mk =
        masks + (size_t)blk * Mpad + m_begin;
    const double *__restrict__ t =
        T + ((size_t)g * Mt + m_begin) * (2 * KW);

    double acc[KW];
#pragma unroll
    for (int j = 0; j < KW; j++) acc[j] = 0.0;

    ulonglong2 cm[U];
    double ct[U][2 * KW];
#pragma unroll
    for (int u = 0; u < U; u++) {
        cm[u] = mk[u];
#pragma unroll
        for (int j = 0; j < 2 * KW; j++) ct[u][j] = t[u * 2 * KW + j];
    }
    const int nb = (m_end - m_begin) / U;
    for (int b = 0; b < nb; b++) {
        ulonglong2 nm[U];
        double nt[U][2 * KW];
        const ulonglong2 *__restrict__ mkn = mk + (size_t)(b + 1) * U;
        const double *__restrict__ tn = t + (size_t)(b + 1) * U * 2 * KW;
#pragma unroll
        for (int u = 0; u < U; u++) {
            nm[u] = mkn[u];
#pragma unroll
            for (int j = 0; j < 2 * KW; j++) nt[u][j] = tn[u * 2 * KW + j];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (__builtin_amdgcn_inverse_ballot_w64(cm[u].x)) {
#pragma unroll
                for (int j = 0; j < KW; j++) acc[j] += ct[u][j];
            }
            if (__builtin_amdgcn_inverse_ballot_w64(cm[u].y)) {
#pragma unroll
                for (int j = 0; j < KW; j++) acc[j] += ct[u][KW + j];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            cm[u] = nm[u];
#pragma unroll
            for (int j = 0; j < 2 * KW; j++) ct[u][j] = nt[u][j];
        }
    }

    const long long slot = blk * 64 + lane;
    if (slot < n) {
        // partial sums: part[ms][slot][K]; final sums: out[slot][ldo]
        double *o = (MS > 1)
            ? out + ((size_t)ms * n + slot) * K + (size_t)g * KW
            : out + (size_t)slot * ldo + (size_t)g * KW;
#pragma unroll
        for (int j = 0; j < KW; j++)
            if (g * KW + j < K) o[j] = acc[j];
    }
}

__global__ __launch_bounds__(256) void k_ll_combine(
    const double *__restrict__ part, long long n, int K, int MS,
    long long ldo, double *__restrict__ out, DoneSignal done)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n * K) {
        const long long slot = i / K;
        const int k = (int)(i - slot * K);
        // partial sums are added in index order (deterministic); the loads
        // are independent of the adds, 8 in flight
        const size_t stride = (size_t)n * K;
        double s = part[i];
        int ms = 1;
        for (; ms + 8 <= MS; ms += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                v[u] = part[(size_t)(ms + u) * stride + i];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; ms < MS; ms++) s += part[(size_t)ms * stride + i];
        out[(size_t)slot * ldo + k] = s;
    }
    // (ONE call per wave, where its lanes have converged: the barrier inside)
    signal_done(done);
}

// One mutation for 8 clusters: EXEC <- lane mask of the cells that observed a
// 1, eight v_add_f64 with SGPR-pair addends; EXEC <- the 0 mask, eight more;
// EXEC restored to all lanes.  Hand-placed so that the scalar pipe issues 3
// instructions per 16 vector adds (the compiler's s_and_saveexec / s_or pairs
// plus its stage copies keep the CU's single scalar unit as busy as the
// vector units).  The wave is fully active here by construction.
__device__ __forceinline__ void ll_step8(double (&a)[8],
                                         const ulonglong2 &m,
                                         const double (&t)[16])
{
    asm volatile(
        "s_mov_b64 exec, %8\n\t"
        "v_add_f64 %0, %0, %10\n\t"
        "v_add_f64 %1, %1, %11\n\t"
        "v_add_f64 %2, %2, %12\n\t"
        "v_add_f64 %3, %3, %13\n\t"
        "v_add_f64 %4, %4, %14\n\t"
        "v_add_f64 %5, %5, %15\n\t"
        "v_add_f64 %6, %6, %16\n\t"
        "v_add_f64 %7, %7, %17\n\t"
        "s_mov_b64 exec, %9\n\t"
        "v_add_f64 %0, %0, %18\n\t"
        "v_add_f64 %1, %1, %19\n\t"
        "v_add_f64 %2, %2, %20\n\t"
        "v_add_f64 %3, %3, %21\n\t"
        "v_add_f64 %4, %4, %22\n\t"
        "v_add_f64 %5, %5, %23\n\t"
        "v_add_f64 %6, %6, %24\n\t"
        "v_add_f64 %7, %7, %25\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]),
          "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
        : "s"(m.x), "s"(m.y), "s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]),
          "s"(t[4]), "s"(t[5]), "s"(t[6]), "s"(t[7]), "s"(t[8]), "s"(t[9]),
          "s"(t[10]), "s"(t[11]), "s"(t[12]), "s"(t[13]), "s"(t[14]),
          "s"(t[15]));
}

// One mutation for 8 clusters, the table values in VGPRs (every lane holds
// the same 16 doubles: broadcast LDS reads) - the step of k_ll8_lds.
__device__ __forceinline__ void ll_step8v(double (&a)[8],
                                          const ulonglong2 &m,
                                          const double (&t)[16])
{
    asm volatile(
        "s_mov_b64 exec, %8\n\t"
        "v_add_f64 %0, %0, %10\n\t"
        "v_add_f64 %1, %1, %11\n\t"
        "v_add_f64 %2, %2, %12\n\t"
        "v_add_f64 %3, %3, %13\n\t"
        "v_add_f64 %4, %4, %14\n\t"
        "v_add_f64 %5, %5, %15\n\t"
        "v_add_f64 %6, %6, %16\n\t"
        "v_add_f64 %7, %7, %17\n\t"
        "s_mov_b64 exec, %9\n\t"
        "v_add_f64 %0, %0, %18\n\t"
        "v_add_f64 %1, %1, %19\n\t"
        "v_add_f64 %2, %2, %20\n\t"
        "v_add_f64 %3, %3, %21\n\t"
        "v_add_f64 %4, %4, %22\n\t"
        "v_add_f64 %5, %5, %23\n\t"
        "v_add_f64 %6, %6, %24\n\t"
        "v_add_f64 %7, %7, %25\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]),
          "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
        : "s"(m.x), "s"(m.y), "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]),
          "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]), "v"(t[8]), "v"(t[9]),
          "v"(t[10]), "v"(t[11]), "v"(t[12]), "v"(t[13]), "v"(t[14]),
          "v"(t[15]));
}

// The whole-row form of the 8-cluster kernel for LONG mutation streams
// (M in the thousands: first-sweep tiles of configs 4 and 5, config 5's
// sweeps).  k_ll8_asm streams a cluster group's table through the scalar
// cache; with 640 KB of table per group (M = 5000) and ~28 groups in flight
// per XCD the streams no longer sit in the XCD's 4 MiB L2, a fifth of the
// scalar loads go to memory (L2 hits 81 %, SQ_WAIT_ANY 37 %) and the rate
// falls from 93 % of the two-add issue peak at M = 1000 to 68 % at M = 5000
// (tools/tile_shape_bench.py).  Here the table reaches the waves through LDS:
// the 4 waves of a workgroup (4 x CB slot blocks of ONE cluster group) fetch
// the table of the next-but-one 64 mutations together - 8 KiB, two coalesced
// 16-byte loads per thread, a whole chunk (~8000 cycles) ahead of its use, so
// it does not matter where the bytes come from - write it to one of two LDS
// buffers, and every wave reads a mutation's 16 doubles with broadcast
// ds_read_b128s into VGPRs (two ping-pong stages).  The lane masks stay on
// the scalar path (1.3 MB per tile, shared by every workgroup: L2-resident).
// One s_barrier per 64 mutations.  Same sums, same order, same bits.
template <int CB>
__global__ __launch_bounds__(256) void k_ll8_lds(
    const ulonglong2 *__restrict__ masks, int Mpad, int Mt, long long n,
    long long nblk, const double *__restrict__ T, int K, long long ldo,
    double *__restrict__ out, int xcd_remap)
{
    constexpr int KW = 8;
    __shared__ double2 tab[2][64 * 8];      // [buffer][mutation][8 x double2]
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int tid = threadIdx.x;
    long long bx, g;
    ll_tile_coords((unsigned)((nblk + 4 * CB - 1) / (4 * CB)),
                   (unsigned)((K + KW - 1) / KW), xcd_remap, bx, g);
    const long long blk0 = (bx * 4 + wave) * CB;
    // (a wave without blocks still takes part in the staging and barriers)
    const bool has_blocks = blk0 < nblk;
    size_t mo[CB];
#pragma unroll
    for (int c = 0; c < CB; c++)
        mo[c] = (size_t)((blk0 + c < nblk) ? blk0 + c : (has_blocks ? blk0 : 0))
            * Mpad;
    const double2 *__restrict__ tg =
        (const double2 *)(T + (size_t)g * Mt * (2 * KW));
    const int n_chunks = (Mt + 63) >> 6;
    const long long t_elems = (long long)Mt * 8;    // double2 of this group

    double acc[CB][KW];
#pragma unroll
    for (int c = 0; c < CB; c++)
#pragma unroll
        for (int j = 0; j < KW; j++) acc[c][j] = 0.0;

    const double2 zero2 = {0.0, 0.0};
    // chunk 0 straight into buffer 0, chunk 1 into registers
    double2 r0, r1;
    {
        const long long e0 = tid, e1 = tid + 256;
        tab[0][tid] = e0 < t_elems ? tg[e0] : zero2;
        tab[0][tid + 256] = e1 < t_elems ? tg[e1] : zero2;
        const long long f0 = 512 + tid, f1 = 512 + tid + 256;
        r0 = f0 < t_elems ? tg[f0] : zero2;
        r1 = f1 < t_elems ? tg[f1] : zero2;
    }
    __syncthreads();
    for (int ch = 0; ch < n_chunks; ch++) {
        const int cur = ch & 1;
        // the next chunk's table (fetched a chunk ago) goes to the other
        // buffer - free since the barrier that ended the previous chunk -,
        // the one after it is requested now
        if (ch + 1 < n_chunks) {
            tab[cur ^ 1][tid] = r0;
            tab[cur ^ 1][tid + 256] = r1;
        }
        if (ch + 2 < n_chunks) {
            const long long f0 = (long long)(ch + 2) * 512 + tid,
                f1 = f0 + 256;
            r0 = f0 < t_elems ? tg[f0] : zero2;
            r1 = f1 < t_elems ? tg[f1] : zero2;
        }
        const int m0 = ch << 6;
        const int m_len = (Mt - m0 < 64) ? Mt - m0 : 64;    // multiple of 8
        const double2 *__restrict__ tb = tab[cur];
        ulonglong2 ma[CB], mb[CB];
        double ta[16], tbv[16];
#pragma unroll
        for (int c = 0; c < CB; c++) ma[c] = masks[mo[c] + m0];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const double2 v = tb[j];
            ta[2 * j] = v.x;
            ta[2 * j + 1] = v.y;
        }
        for (int m = 0; m < m_len; m += 2) {
            // stage B <- mutation m + 1 (loads issued before A's adds)
#pragma unroll
            for (int c = 0; c < CB; c++) mb[c] = masks[mo[c] + m0 + m + 1];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const double2 v = tb[(m + 1) * 8 + j];
                tbv[2 * j] = v.x;
                tbv[2 * j + 1] = v.y;
            }
#pragma unroll
            for (int c = 0; c < CB; c++) ll_step8v(acc[c], ma[c], ta);
            // stage A <- mutation m + 2 (the row past the chunk's end is the
            // next buffer's first row or padding: loaded, never added)
            const int m2 = m + 2 < 64 ? m + 2 : 63;
#pragma unroll
            for (int c = 0; c < CB; c++) ma[c] = masks[mo[c] + m0 + m2];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const double2 v = tb[m2 * 8 + j];
                ta[2 * j] = v.x;
                ta[2 * j + 1] = v.y;
            }
#pragma unroll
            for (int c = 0; c < CB; c++) ll_step8v(acc[c], mb[c], tbv);
        }
        __syncthreads();
    }
    if (!has_blocks) return;
#pragma unroll
    for (int c = 0; c < CB; c++) {
        const long long slot = (blk0 + c) * 64 + lane;
        if (blk0 + c < nblk && slot < n) {
            double *o = out + (size_t)slot * ldo + (size_t)g * KW;
#pragma unroll
            for (int j = 0; j < KW; j++)
                if (g * KW + j < K) o[j] = acc[c][j];
        }
    }
}

// 8 clusters x CB slot blocks per wave, two ping-pong stages of one mutation
// each (no stage copies), hand-placed inner block.  With CB = 2 a wave owns
// 128 cells: the 16 table doubles of a mutation are loaded once and feed 32
// masked adds, halving the scalar-pipe and scalar-cache work per add.
// Same sums, same order, same bits as k_ll<8>.
// SPLIT = false: one wave sums all mutations of its tile (the large launches:
// first sweep, tiles).  SPLIT = true: the mutation-split form for small
// launches (partial sums + k_ll_combine).  Two symbols, so that per-kernel
// profiles do not mix millisecond launches with microsecond ones.
template <int CB, bool SPLIT>
__global__ __launch_bounds__(256) void k_ll8_asm(
    const ulonglong2 *__restrict__ masks, int Mpad, int Mt, long long n,
    long long nblk, const double *__restrict__ T, int K, long long ldo,
    double *__restrict__ out, int xcd_remap, int MS_arg, int m_chunk,
    const ulonglong2 *masks_pf, const double *T_pf, DoneSignal done)
{
    // masks_pf / T_pf: the same two arrays again, or NULL (no prefetch).  The
    // prefetch below hands addresses to inline assembly; were they derived
    // from the __restrict__ parameters, the compiler would have to assume the
    // assembly may write there and would turn every SCALAR load of the kernel
    // into a vector load.
    constexpr int KW = 8;
    const bool PF = masks_pf != nullptr;
    const int MS = SPLIT ? MS_arg : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    long long bx, g;
    // SPLIT = false: the 4 waves of a workgroup own 4 different sets of CB
    // blocks.  SPLIT = true: they own 4 consecutive mutation chunks of the
    // SAME CB blocks and add their sums in wave order through LDS, so only
    // ceil(MS / 4) partial planes travel to k_ll_combine.
    const int MSq = (MS + 3) >> 2;
    ll_tile_coords(SPLIT
            ? (unsigned)((nblk + CB - 1) / CB) * (unsigned)MSq
            : (unsigned)((nblk + 4 * CB - 1) / (4 * CB)),
        (unsigned)((K + KW - 1) / KW), xcd_remap, bx, g);
    const int q = SPLIT ? (int)(bx % MSq) : 0;  // chunk quad of this workgroup
    if (SPLIT) bx /= MSq;
    const int ms = SPLIT ? q * 4 + wave : 0;    // mutation chunk of this wave
    const long long blk0 = SPLIT ? bx * CB : (bx * 4 + wave) * CB;
    if (blk0 >= nblk) {                 // SPLIT: the whole workgroup leaves
        if constexpr (SPLIT) signal_done(done);
        return;
    }
    const int m_begin = SPLIT ? ms * m_chunk : 0;
    const int m_len = !SPLIT ? Mt
        : (ms >= MS ? 0
            : ((m_begin + m_chunk < Mt) ? m_chunk : Mt - m_begin));

    // mask rows of the wave's CB blocks, as offsets from the one `masks` base
    // (a block past the end re-reads the wave's first block, never stored)
    size_t mo[CB];
#pragma unroll
    for (int c = 0; c < CB; c++)
        mo[c] = (size_t)((blk0 + c < nblk) ? blk0 + c : blk0) * Mpad
            + (m_len > 0 ? m_begin : 0);
    const size_t t_off = ((size_t)g * Mt + (m_len > 0 ? m_begin : 0))
        * (2 * KW);                     // of this wave's first table stage
    const double *__restrict__ tp = T + t_off;

    double acc[CB][KW];
#pragma unroll
    for (int c = 0; c < CB; c++)
#pragma unroll
        for (int j = 0; j < KW; j++) acc[c][j] = 0.0;

    // Scalar loads return out of order, so the only wait is lgkmcnt(0): wait
    // for the current stage FIRST, then issue the next stage's loads, then
    // the masked adds run under those loads.
    //
    // L2 prefetch (PF): with M in the thousands the streams a wave walks no
    // longer sit in its XCD's L2 - 80 KB of masks per slot block (782 blocks
    // at config 5) or 640 KB of table per cluster group (thousands of groups
    // in a first-sweep tile) - and a scalar load that goes to HBM costs
    // several stages of adds.  Before its first stage and then every 64
    // mutations the wave touches the masks and the table of the NEXT 64 with
    // vector loads into a register nobody reads (never waited for until the
    // wave ends): by the time the scalar loads get there the lines are in L2.
    // 10 vector loads per 2048 adds.  A wave of a split launch (a chunk of a
    // few dozen mutations, its table written a moment ago by another XCD)
    // gets its whole chunk under way with the first block.
    // (No "memory" clobber on these: it would cost the kernel its scalar
    // loads - the compiler only keeps uniform loads scalar while nothing in
    // the kernel may write memory.)
    typedef unsigned pf_u32x4 __attribute__((ext_vector_type(4)));
    pf_u32x4 pf_sink = {0u, 0u, 0u, 0u};
    // wave-uniform bases in SGPRs + a per-lane byte offset that does not
    // depend on m (a lane-dependent induction variable would make the
    // compiler express the SCALAR loads through it, i.e. as vector loads)
    const unsigned pf_lo = (unsigned)lane * 16u, pf_hi = pf_lo + 4096u;
    // mutations [MP, MP + 64) of this wave's chunk: one load per slot block
    // for the masks (lanes past the padded row stay out), 8 stages of table
    // (16 doubles each) per KiB
#define PF_TAB(MP, I, OFF, IMM)                                               \
    if ((MP) + 8 * (I) < m_len)                                               \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #IMM            \
                     : "+v"(pf_sink) : "v"(OFF), "s"(t64_));
#define PF_BLOCK(MP)                                                          \
    {                                                                         \
        if (m_begin + (MP) + lane < Mpad) {                                   \
            _Pragma("unroll") for (int c = 0; c < CB; c++) {                  \
                const ulonglong2 *a_ = masks_pf + mo[c] + (MP);               \
                asm volatile("global_load_dwordx4 %0, %1, %2"                 \
                             : "+v"(pf_sink) : "v"(pf_lo), "s"(a_));          \
            }                                                                 \
        }                                                                     \
        const double *t64_ = T_pf + t_off + (size_t)(MP) * 16;                \
        PF_TAB(MP, 0, pf_lo, 0) PF_TAB(MP, 1, pf_lo, 1024)                    \
        PF_TAB(MP, 2, pf_lo, 2048) PF_TAB(MP, 3, pf_lo, 3072)                 \
        PF_TAB(MP, 4, pf_hi, 0) PF_TAB(MP, 5, pf_hi, 1024)                    \
        PF_TAB(MP, 6, pf_hi, 2048) PF_TAB(MP, 7, pf_hi, 3072)                 \
    }
    if (PF && m_len > 0) PF_BLOCK(0)

    ulonglong2 ma[CB], mb[CB];
    double ta[16], tb[16];
#pragma unroll
    for (int c = 0; c < CB; c++) ma[c] = masks[mo[c]];
#pragma unroll
    for (int j = 0; j < 16; j++) ta[j] = tp[j];

    for (int m = 0; m < m_len; m += 2) {        // chunks are multiples of 8
        if (PF && (m & 63) == 0 && m + 64 < m_len) PF_BLOCK(m + 64)
        __builtin_amdgcn_s_waitcnt(0xC07F);     // stage A landed
#pragma unroll
        for (int c = 0; c < CB; c++) mb[c] = masks[mo[c] + m + 1];
#pragma unroll
        for (int j = 0; j < 16; j++) tb[j] = tp[16 + j];
#pragma unroll
        for (int c = 0; c < CB; c++) ll_step8(acc[c], ma[c], ta);
        __builtin_amdgcn_s_waitcnt(0xC07F);     // stage B landed
#pragma unroll
        for (int c = 0; c < CB; c++) ma[c] = masks[mo[c] + m + 2];
#pragma unroll
        for (int j = 0; j < 16; j++) ta[j] = tp[32 + j];
#pragma unroll
        for (int c = 0; c < CB; c++) ll_step8(acc[c], mb[c], tb);
        tp += 32;
    }
#undef PF_BLOCK
#undef PF_TAB

    if (PF)     // the sink stays allocated to the end; nothing in flight
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink));

    if constexpr (SPLIT) {
        // sums of the 4 chunks of this workgroup, added in wave order
        __shared__ double red[4][CB * KW][64];
#pragma unroll
        for (int c = 0; c < CB; c++)
#pragma unroll
            for (int j = 0; j < KW; j++) red[wave][c * KW + j][lane] = acc[c][j];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < (CB * KW) / 4; i++) {
            const int ck = wave + 4 * i;
            const int c = ck / KW, j = ck - c * KW;
            const double s = ((red[0][ck][lane] + red[1][ck][lane])
                + red[2][ck][lane]) + red[3][ck][lane];
            const long long slot = (blk0 + c) * 64 + lane;
            if (blk0 + c < nblk && slot < n && g * KW + j < K) {
                double *o = (MSq > 1)
                    ? out + ((size_t)q * n + slot) * K + (size_t)g * KW
                    : out + (size_t)slot * ldo + (size_t)g * KW;
                o[j] = s;
            }
        }
        signal_done(done);
    } else {
#pragma unroll
        for (int c = 0; c < CB; c++) {
            const long long slot = (blk0 + c) * 64 + lane;
            if (blk0 + c < nblk && slot < n) {
                double *o = out + (size_t)slot * ldo + (size_t)g * KW;
#pragma unroll
                for (int j = 0; j < KW; j++)
                    if (g * KW + j < K) o[j] = acc[c][j];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K2p: the same sums over CALLER-BUILT tables in strict mutation order - the
// bit-exact path of CRP._rg_init_split (libs/CRP.py:547-561, whose
// `ll_j > ll_i` is the one discrete decision on the path) and of the
// new-cluster term (libs/CRP.py:230-234); K is 1 or 2 there and the launch
// has a handful of (block, cluster) chains on a chip with a thousand idle
// SIMDs, so what matters is the latency of ONE chain of M dependent adds.
// (Rounds 1-2 ran a chain on one wave - k_ll<2>: 144 us, then k_ll_seq with
// v_readlane'd masks and LDS tables: 34 us; both are gone, the pipeline
// below takes 9 us.)  A workgroup is one chain on four SIMDs:
//   waves 1-3 (producers), lane <-> MUTATION m0 + j of the current 64-chunk:
//     the lane masks {ones, zeros}[m0 + j] and the table pair {L1, L0}[m0 + j]
//     arrive by coalesced loads (4 chunks ahead); for each of its ~21 cells c
//     the value that cell adds at that mutation,
//         x[c][j] = bit c of ones ? L1 : bit c of zeros ? L0 : +0.0,
//     is selected with integer ops (sign-extended bit AND the table words) and
//     written to LDS - conflict-free, lane j = consecutive 8 bytes;
//   wave 0 (consumer), lane <-> CELL c: reads its row x[c][0..63] two values
//     per ds_read_b128 (row stride 66 doubles: conflict-free) and performs the
//     chain's adds, one v_add_f64 per mutation in mutation order.
// Adding +0.0 where the reference adds nothing is exact (a partial sum is never -0.0:
// it starts at +0.0 and +0.0 + -0.0 = +0.0).  Two x stages: the producers
// fill chunk i + 1 while the consumer adds chunk i; one s_barrier per chunk,
// with an LDS-only wait so that the global prefetch stays in flight.
// ---------------------------------------------------------------------------
#define SEQP_XS 66                      // doubles per cell row of a stage
#define SEQP_DEPTH 4                    // chunks of global loads in flight
#define SEQP_LDS (2 * 64 * SEQP_XS * sizeof(double))

__device__ __forceinline__ void seqp_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef unsigned seqp_u32x4 __attribute__((ext_vector_type(4)));

template <int C0, int C1>
__device__ __forceinline__ void seqp_select(const seqp_u32x4 &mk, double l1,
                                            double l0, double *xs)
{
    // mk = {ones lo, ones hi, zeros lo, zeros hi}
    const unsigned l1lo = (unsigned)__double2loint(l1);
    const unsigned l1hi = (unsigned)__double2hiint(l1);
    const unsigned l0lo = (unsigned)__double2loint(l0);
    const unsigned l0hi = (unsigned)__double2hiint(l0);
#pragma unroll
    for (int c = C0; c < C1; c++) {
        const unsigned t1 = (unsigned)__builtin_amdgcn_sbfe(
            (int)(c < 32 ? mk.x : mk.y), c & 31, 1);
        const unsigned t0 = (unsigned)__builtin_amdgcn_sbfe(
            (int)(c < 32 ? mk.z : mk.w), c & 31, 1);
        xs[c * SEQP_XS] = __hiloint2double(
            (int)((t1 & l1hi) | (t0 & l0hi)), (int)((t1 & l1lo) | (t0 & l0lo)));
    }
}

// One producer wave: cells C0 .. C1 - 1 of every chunk.  The global loads are
// spelled out with their own vmcnt bookkeeping - three loads per chunk,
// SEQP_DEPTH chunks in flight, so a chunk is complete when at most
// 3 (SEQP_DEPTH - 1) younger loads are outstanding (the compiler's own counter
// insertion gives up at the loop edge and waits for everything, i.e. one
// memory round trip per chunk).  The chunk count is rounded up to the depth
// (straight-line loop body); a chunk past the end re-reads the last one and
// is ignored by the consumer, a mutation past M re-reads table entry M - 1,
// which the empty padding masks turn into +0.0.
template <int C0, int C1>
__device__ __forceinline__ void seqp_producer(
    const ulonglong2 *__restrict__ mk, const double *__restrict__ t1,
    const double *__restrict__ t0, int nch, int nch_up, int M, int lane,
    double *xs)
{
    static_assert(SEQP_DEPTH == 4, "vmcnt below is 3 * (SEQP_DEPTH - 1)");
    seqp_u32x4 mq[SEQP_DEPTH];
    double l1q[SEQP_DEPTH], l0q[SEQP_DEPTH];
#define SEQP_LOAD(D, CH)                                                      \
    {                                                                         \
        const int ch_ = ((CH) < nch) ? (CH) : nch - 1;                        \
        const int m_ = (ch_ << 6) + lane;                                     \
        const int mt_ = (m_ < M) ? m_ : M - 1;                                \
        const unsigned om_ = (unsigned)m_ * 16u, ot_ = (unsigned)mt_ * 8u;    \
        asm volatile("global_load_dwordx4 %0, %1, %2"                         \
                     : "=&v"(mq[D]) : "v"(om_), "s"(mk) : "memory");          \
        asm volatile("global_load_dwordx2 %0, %1, %2"                         \
                     : "=&v"(l1q[D]) : "v"(ot_), "s"(t1) : "memory");         \
        asm volatile("global_load_dwordx2 %0, %1, %2"                         \
                     : "=&v"(l0q[D]) : "v"(ot_), "s"(t0) : "memory");         \
    }
#pragma unroll
    for (int d = 0; d < SEQP_DEPTH; d++) SEQP_LOAD(d, d)
    for (int ch = 0; ch < nch_up; ch += SEQP_DEPTH) {
#pragma unroll
        for (int d = 0; d < SEQP_DEPTH; d++) {
            asm volatile("s_waitcnt vmcnt(9)"
                         : "+v"(mq[d]), "+v"(l1q[d]), "+v"(l0q[d]));
            seqp_select<C0, C1>(mq[d], l1q[d], l0q[d],
                                xs + (size_t)(d & 1) * 64 * SEQP_XS);
            SEQP_LOAD(d, ch + d + SEQP_DEPTH)
            seqp_barrier();
        }
    }
    // nothing may be in flight into dead registers at the end of the wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef SEQP_LOAD
}

__global__ __launch_bounds__(256) void k_ll_seqp(
    const ulonglong2 *__restrict__ masks, int Mpad, int M, long long n,
    long long nblk, const double *__restrict__ L1,
    const double *__restrict__ L0, int K, long long ldo,
    double *__restrict__ out, DoneSignal done)
{
    extern __shared__ double seqp_x[];          // [2][64][SEQP_XS]
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const long long blk = blockIdx.x;
    const int k = blockIdx.y;
    const int nch = Mpad >> 6;
    const int nch_up = (nch + SEQP_DEPTH - 1) & ~(SEQP_DEPTH - 1);
    if (wave == 0) {
        double acc = 0.0;
        for (int ch = 0; ch < nch; ch++) {
            seqp_barrier();                     // chunk ch is complete
            const double2 *xr = (const double2 *)(seqp_x
                + (size_t)(ch & 1) * 64 * SEQP_XS + (size_t)lane * SEQP_XS);
#pragma unroll
            for (int u = 0; u < 32; u++) {
                const double2 v = xr[u];
                acc += v.x;
                acc += v.y;
            }
        }
        for (int ch = nch; ch < nch_up; ch++) seqp_barrier();
        const long long slot = blk * 64 + lane;
        if (slot < n) out[(size_t)slot * ldo + k] = acc;
    } else {
        // wave-uniform bases (SGPR pairs); the lane enters as a 32-bit offset
        const ulonglong2 *__restrict__ mk = masks + (size_t)blk * Mpad;
        const double *__restrict__ t1 = L1 + (size_t)k * M;
        const double *__restrict__ t0 = L0 + (size_t)k * M;
        double *xs = seqp_x + lane;
        if (wave == 1)
            seqp_producer<0, 22>(mk, t1, t0, nch, nch_up, M, lane, xs);
        else if (wave == 2)
            seqp_producer<22, 43>(mk, t1, t0, nch, nch_up, M, lane, xs);
        else
            seqp_producer<43, 64>(mk, t1, t0, nch, nch_up, M, lane, xs);
    }
    signal_done(done);
}

// tables of a K2p launch, pinned arena -> device: ONE trip over the host link
// (the chains of a launch would each re-read them in place)
__global__ __launch_bounds__(256) void k_stage_copy(
    const double2 *__restrict__ src, double2 *__restrict__ dst, long long n2)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) dst[i] = src[i];
}

// ---------------------------------------------------------------------------
// K2t: per slot the four largest entries of out[s][k] + prior[k], k < K <= 64,
// the columns of the three largest (first on ties) and their log-likelihoods:
// the sweep's hint.  The
// priors travel as kernel arguments (no memory to fetch them from); the matrix
// was just written and is read from L2.
// ---------------------------------------------------------------------------
// (the record's float fields: third / fourth rounded UP - they are upper
// bounds on "everything else"; e2 / e3 = the likelihood weights of the second
// / third column relative to the first, see bnpc_top2 in the header)
__device__ __forceinline__ void top2_floats(bnpc_top2 &t, double third,
    double fourth, double lb, double ls, double lt, bool has2, bool has3)
{
    t.third = __double2float_ru(third);
    t.fourth = __double2float_ru(fourth);
    t.e2 = has2 ? (float)exp(ls - lb) : 0.0f;
    t.e3 = has3 ? (float)exp(lt - lb) : 0.0f;
}

__global__ __launch_bounds__(256) void k_row_top2(
    const double *__restrict__ ll, long long n, long long ldo, int K,
    Top2Prior prior, bnpc_top2 *__restrict__ out,
    double *__restrict__ host_ll, const long long *__restrict__ order)
{
    const long long slot = (long long)blockIdx.x * 256 + threadIdx.x;
    if (slot >= n) return;
    // (order: record `slot` is made from row order[slot] - the hints in the
    // sweep's visiting order)
    const long long row = order ? order[slot] : slot;
    const double *__restrict__ r = ll + (size_t)row * ldo;
    double best = -INFINITY, second = -INFINITY, third = -INFINITY;
    double fourth = -INFINITY;
    double lb = 0.0, ls = 0.0, lt = 0.0;
    int col = 0, col2 = -1, col3 = -1;
    for (int k = 0; k < K; k++) {
        const double l = r[k];
        const double v = l + prior.v[k];
        if (v > best) {
            fourth = third;
            third = second;
            lt = ls;
            col3 = col2;
            second = best;
            ls = lb;
            col2 = best > -INFINITY ? col : -1;
            best = v;
            lb = l;
            col = k;
        } else if (v > second) {
            fourth = third;
            third = second;
            lt = ls;
            col3 = col2;
            second = v;
            ls = l;
            col2 = k;
        } else if (v > third) {
            fourth = third;
            third = v;
            lt = l;
            col3 = k;
        } else if (v > fourth) {
            fourth = v;
        }
    }
    bnpc_top2 t;
    t.best = best;
    t.second = second;
    t.ll_best = lb;
    t.ll_second = ls;
    t.ll_third = lt;
    t.col = (int16_t)col;
    t.col2 = (int16_t)col2;
    t.col3 = (int16_t)col3;
    top2_floats(t, third, fourth, lb, ls, lt, col2 >= 0, col3 >= 0);
    // A row without a clear winner and with a fourth entry within reach of
    // the runner-up is one the sweep will have to scan (the host decides
    // cells among up to three candidates from the hint alone; a clear winner
    // needs none): it is written through to the host's
    // copy of the matrix here, with the hints, so that the scan does not have
    // to wait for the whole matrix to be copied (row_here = 1).
    int through = 0;
    // (on the record's own rounded value: the host can tell from the record)
    if (host_ll && (double)t.fourth > second - 72.0 && second > best - 48.0) {
        double *__restrict__ h = host_ll + (size_t)row * ldo;
        for (int k = 0; k < K; k++) h[k] = r[k];
        through = 1;
    }
    t.row_here = (int16_t)through;
    out[slot] = t;
}

// The same record for rows of MORE than 64 columns (a running chain with
// hundreds of clusters, the first sweep of a data set whose whole matrix fits
// the host budget: up to 32767 columns): one wave per row, lane l takes
// columns l, l + 64, ... and keeps its own four largest; the 64 lists are
// merged by an xor butterfly under the total order (value descending, column
// ascending), so every lane ends with the row's record - the first column on
// ties, as the one-thread kernel.  Priors from device memory.  Rows the sweep
// will scan are written through to the host matrix when that is a few KiB per
// row (through_max columns), coalesced.
struct Top4 {
    double b, s, t, f;      // the four largest entries of ll + prior
    double lb, ls, lt;      // the log-likelihoods behind the first three
    int cb, cs, ct;         // their columns (INT_MAX: none)
};

__device__ __forceinline__ bool top4_before(double v, int k, double x, int cx)
{
    return v > x || (v == x && k < cx);
}

__device__ __forceinline__ void top4_insert(Top4 &q, double v, double l, int k)
{
    if (top4_before(v, k, q.b, q.cb)) {
        q.f = q.t;
        q.t = q.s; q.lt = q.ls; q.ct = q.cs;
        q.s = q.b; q.ls = q.lb; q.cs = q.cb;
        q.b = v; q.lb = l; q.cb = k;
    } else if (top4_before(v, k, q.s, q.cs)) {
        q.f = q.t;
        q.t = q.s; q.lt = q.ls; q.ct = q.cs;
        q.s = v; q.ls = l; q.cs = k;
    } else if (top4_before(v, k, q.t, q.ct)) {
        q.f = q.t;
        q.t = v; q.lt = l; q.ct = k;
    } else if (v > q.f) {
        q.f = v;
    }
}

__global__ __launch_bounds__(256) void k_row_top4_wave(
    const double *__restrict__ ll, long long n, long long ldo, int K,
    const double *__restrict__ prior, bnpc_top2 *__restrict__ out,
    double *__restrict__ host_ll, int through_max,
    const long long *__restrict__ order)
{
    const int lane = threadIdx.x & 63;
    const long long slot = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= n) return;
    const long long row = order ? order[slot] : slot;
    const double *__restrict__ r = ll + (size_t)row * ldo;
    const int none = 0x7fffffff;
    Top4 q = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, 0.0, 0.0, 0.0,
              none, none, none};
    for (int k = lane; k < K; k += 64) {
        const double l = r[k];
        top4_insert(q, l + prior[k], l, k);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Top4 o;
        o.b = __shfl_xor(q.b, off);
        o.s = __shfl_xor(q.s, off);
        o.t = __shfl_xor(q.t, off);
        o.f = __shfl_xor(q.f, off);
        o.lb = __shfl_xor(q.lb, off);
        o.ls = __shfl_xor(q.ls, off);
        o.lt = __shfl_xor(q.lt, off);
        o.cb = __shfl_xor(q.cb, off);
        o.cs = __shfl_xor(q.cs, off);
        o.ct = __shfl_xor(q.ct, off);
        // (an entry that is not there compares below everything: -inf, none)
        top4_insert(q, o.b, o.lb, o.cb);
        top4_insert(q, o.s, o.ls, o.cs);
        top4_insert(q, o.t, o.lt, o.ct);
        if (o.f > q.f) q.f = o.f;
    }
    int through = 0;
    const float f_up = __double2float_ru(q.f);
    if (host_ll && K <= through_max && (double)f_up > q.s - 72.0
        && q.s > q.b - 48.0) {
        double *__restrict__ h = host_ll + (size_t)row * ldo;
        for (int k = lane; k < K; k += 64) h[k] = r[k];
        through = 1;
    }
    // (the record's two exponentials side by side on lanes 0 and 1)
    const float e_mine = (float)exp((lane == 0 ? q.ls : q.lt) - q.lb);
    const float e_next = __shfl(e_mine, 1);
    if (lane == 0) {
        bnpc_top2 t;
        t.best = q.b;
        t.second = q.s;
        t.ll_best = q.lb;
        t.ll_second = q.ls;
        t.ll_third = q.lt;
        t.col = (int16_t)(q.cb == none ? 0 : q.cb);
        t.col2 = (int16_t)(q.cs == none || !(q.s > -INFINITY) ? -1 : q.cs);
        t.col3 = (int16_t)(q.ct == none || !(q.t > -INFINITY) ? -1 : q.ct);
        t.third = __double2float_ru(q.t);
        t.fourth = f_up;
        t.e2 = t.col2 >= 0 ? e_mine : 0.0f;
        t.e3 = t.col3 >= 0 ? e_next : 0.0f;
        t.row_here = (int16_t)through;
        out[slot] = t;
    }
}

// ---------------------------------------------------------------------------
// K3: column counts of 1s and 0s over chunks of cell segments
//   n1[g][m] = #{c in segment g : x_cm = 1},  n0 likewise
//   = the sums over a cell subset inside CRP._get_log_A (libs/CRP.py:359-368),
//     CRP._init_cl_params_new (libs/CRP.py:183-188) and the flat sums of
//     CRP._get_ll_ratio (libs/CRP.py:716-733), as exact integers.
// thread <-> mutation; block <-> (chunk of <= 32 cells of one segment, 256
// mutations); a row word is shared by the 64 lanes of a wave (broadcast load).
// ---------------------------------------------------------------------------
// The hint of a first-sweep TILE (thousands of columns): per row the largest
// entry of ll + prior, its column (first one on ties) and the largest entry
// among all other columns - enough for the sweep's dominance test.  One
// workgroup per row; priors from device memory.  The record is bnpc_top2 with
// the column as 32 bits in (col | col2 << 16), col3 = -1, row_here = 2.
__global__ __launch_bounds__(256) void k_row_top2_wide(
    const double *__restrict__ ll, long long ldo, int K,
    const double *__restrict__ prior, bnpc_top2 *__restrict__ out)
{
    __shared__ double s_b[4], s_s[4];
    __shared__ int s_c[4];
    const long long slot = blockIdx.x;
    const double *__restrict__ r = ll + (size_t)slot * ldo;
    double best = -INFINITY, second = -INFINITY;
    int col = 0x7fffffff;
    for (int k = threadIdx.x; k < K; k += 256) {
        const double v = r[k] + prior[k];
        if (v > best) {                 // (ascending k: the first maximum)
            second = best;
            best = v;
            col = k;
        } else if (v > second) {
            second = v;
        }
    }
    auto merge = [&](double ob, int oc, double os) {
        if (ob > best || (ob == best && oc < col)) {
            second = os > best ? os : best;
            best = ob;
            col = oc;
        } else {
            const double m = ob > os ? ob : os;
            if (m > second) second = m;
        }
    };
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(best, off), os = __shfl_down(second, off);
        const int oc = __shfl_down(col, off);
        merge(ob, oc, os);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_b[wave] = best;
        s_s[wave] = second;
        s_c[wave] = col;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) merge(s_b[w], s_c[w], s_s[w]);
        bnpc_top2 t;
        t.best = best;
        t.second = second;
        t.third = -INFINITY;
        t.fourth = -INFINITY;
        t.e2 = t.e3 = 0.0f;
        t.ll_best = t.ll_second = t.ll_third = 0.0;
        const unsigned c32 = (unsigned)(col == 0x7fffffff ? 0 : col);
        t.col = (int16_t)(uint16_t)(c32 & 0xffffu);
        t.col2 = (int16_t)(uint16_t)(c32 >> 16);
        t.col3 = -1;
        t.row_here = 2;
        out[slot] = t;
    }
}

struct Chunk {
    long long begin, end;   // range in the cells[] list
    long long seg;
};

__global__ __launch_bounds__(256) void k_colcounts(
    const ulonglong2 *__restrict__ rows, int W, int M,
    const long long *__restrict__ cells, const Chunk *__restrict__ chunks,
    int *__restrict__ n1, int *__restrict__ n0)
{
    const Chunk ch = chunks[blockIdx.y];
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int w = m >> 6, b = m & 63;
    int c1 = 0, c0 = 0;
    // chunks are short (<= 32 cells) so that a launch has thousands of
    // workgroups; 4 independent row loads in flight per thread
    long long i = ch.begin;
    for (; i + 4 <= ch.end; i += 4) {
        const long long ca = cells[i], cb = cells[i + 1], cc = cells[i + 2],
            cd = cells[i + 3];
        const ulonglong2 ra = rows[(size_t)ca * W + w];
        const ulonglong2 rb = rows[(size_t)cb * W + w];
        const ulonglong2 rc = rows[(size_t)cc * W + w];
        const ulonglong2 rd = rows[(size_t)cd * W + w];
        c1 += (int)((ra.x >> b) & 1ull) + (int)((rb.x >> b) & 1ull)
            + (int)((rc.x >> b) & 1ull) + (int)((rd.x >> b) & 1ull);
        c0 += (int)((ra.y >> b) & 1ull) + (int)((rb.y >> b) & 1ull)
            + (int)((rc.y >> b) & 1ull) + (int)((rd.y >> b) & 1ull);
    }
    for (; i < ch.end; i++) {
        const long long cell = cells[i];
        const ulonglong2 r = rows[(size_t)cell * W + w];
        c1 += (int)((r.x >> b) & 1ull);
        c0 += (int)((r.y >> b) & 1ull);
    }
    if (c1) atomicAdd(&n1[(size_t)ch.seg * M + m], c1);
    if (c0) atomicAdd(&n0[(size_t)ch.seg * M + m], c0);
}

// ---------------------------------------------------------------------------
// K3b: column counts of G <= a few dozen segments of a VIEW, from its lane
// masks: segment g is given as membership words member[g][blk] (bit s = slot
// 64*blk + s belongs to g), and
//   n1[g][m] = sum_blk popcount(masks[blk][m].ones  & member[g][blk])
//   n0[g][m] = sum_blk popcount(masks[blk][m].zeros & member[g][blk])
// Same integers as K3.  No cell lists, no atomics, no zero-fill: thread <->
// mutation (coalesced 16-byte mask loads), the 4 waves of a workgroup take
// every 4th block and add up through LDS, 8 segments per thread share each
// loaded mask.  Results go to a device buffer (kept for K6) and, if given, to
// pinned host memory in place.
// ---------------------------------------------------------------------------
#define CM_SEG 8

__global__ __launch_bounds__(256) void k_counts_masks(
    const ulonglong2 *__restrict__ masks, int Mpad, int M, long long nblk,
    const unsigned long long *__restrict__ member, int G,
    int *__restrict__ n1, int *__restrict__ n0, int *__restrict__ h1,
    int *__restrict__ h0, DoneSignal done)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m = blockIdx.x * 64 + lane;
    const int g0 = blockIdx.y * CM_SEG;
    // The membership words of this workgroup's segments, fetched ONCE with
    // coalesced loads (they may sit in host memory: a dependent load per
    // loop iteration would pay the link's latency every time) into LDS,
    // [blk][CM_SEG]; segments past G read as empty.
    extern __shared__ unsigned long long mem_lds[];
    for (long long i = threadIdx.x; i < nblk * CM_SEG; i += 256) {
        const long long b = i / CM_SEG;
        const int j = (int)(i - b * CM_SEG);
        mem_lds[i] = (g0 + j < G) ? member[(size_t)(g0 + j) * nblk + b] : 0ull;
    }
    __syncthreads();
    int c1[CM_SEG], c0[CM_SEG];
#pragma unroll
    for (int j = 0; j < CM_SEG; j++) c1[j] = c0[j] = 0;
    for (long long b = wave; b < nblk; b += 4) {
        const ulonglong2 mk = masks[(size_t)b * Mpad + m];
        const unsigned long long *mem = mem_lds + b * CM_SEG;
#pragma unroll
        for (int j = 0; j < CM_SEG; j++) {
            c1[j] += __popcll(mk.x & mem[j]);
            c0[j] += __popcll(mk.y & mem[j]);
        }
    }
    // the membership words are done with: the same LDS takes the partial
    // counts of the 4 waves (the launch reserves at least that much)
    __syncthreads();
    int (*red)[2 * CM_SEG][64] = (int (*)[2 * CM_SEG][64])mem_lds;
#pragma unroll
    for (int j = 0; j < CM_SEG; j++) {
        red[wave][2 * j][lane] = c1[j];
        red[wave][2 * j + 1][lane] = c0[j];
    }
    __syncthreads();
    // wave w finishes segments j = w, w + 4
#pragma unroll
    for (int i = 0; i < CM_SEG / 4; i++) {
        const int j = wave + 4 * i;
        if (g0 + j < G && m < M) {
            const int s1 = red[0][2 * j][lane] + red[1][2 * j][lane]
                + red[2][2 * j][lane] + red[3][2 * j][lane];
            const int s0 = red[0][2 * j + 1][lane] + red[1][2 * j + 1][lane]
                + red[2][2 * j + 1][lane] + red[3][2 * j + 1][lane];
            const size_t at = (size_t)(g0 + j) * M + m;
            n1[at] = s1;
            n0[at] = s0;
            if (h1) {
                h1[at] = s1;
                h0[at] = s0;
            }
        }
    }
    signal_done(done);
}

// ---------------------------------------------------------------------------
// K6: total log-likelihood from per-cluster counts, up to 4 trial error pairs
//   out[e] = sum_{k,m} n1[k][m]*L1_e(theta[k][m]) + n0[k][m]*L0_e(theta[k][m])
//   = CRP.get_ll_full (libs/CRP.py:237-238) and
//     CRP_errors_learning.get_ll_full_error (libs/CRP_learning_errors.py:58-63)
// fixed grid, fixed per-thread order, fixed reduction tree -> deterministic.
// ---------------------------------------------------------------------------
#define TOTAL_BLOCKS 256

__global__ __launch_bounds__(256) void k_ll_total(
    const float *__restrict__ theta, const int *__restrict__ n1,
    const int *__restrict__ n0, long long KM, int E, double FP0, double FN0,
    double FP1, double FN1, double FP2, double FN2, double FP3, double FN3,
    double *__restrict__ partial, DoneSignal done)
{
    const double FPs[4] = {FP0, FP1, FP2, FP3};
    const double FNs[4] = {FN0, FN1, FN2, FN3};
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < KM;
         i += stride) {
        const int c1 = n1[i], c0 = n0[i];
        if ((c1 | c0) == 0) continue;
        const float th = theta[i];
        const double th64 = (double)th;
        const double om64 = (double)(1.0f - th);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (e < E) {
                const double l1 = log(th64 * (1.0 - FNs[e]) + om64 * FPs[e]);
                const double l0 = log(th64 * FNs[e] + om64 * (1.0 - FPs[e]));
                acc[e] += (double)c1 * l1 + (double)c0 * l0;
            }
        }
    }
    __shared__ double red[4][256];
#pragma unroll
    for (int e = 0; e < 4; e++) red[e][threadIdx.x] = acc[e];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                red[e][threadIdx.x] += red[e][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[(size_t)blockIdx.x * 4 + threadIdx.x] =
        red[threadIdx.x][0];
    signal_done(done);
}


// ---------------------------------------------------------------------------
// K7: screen of a parameter batch - which proposals of CRP.MH_cluster_params
// (libs/CRP.py:314-344) are declined FOR CERTAIN.
//
// The host evaluates the MH update of a cluster profile bit for bit as SciPy
// does (bnpc_hostmath.cpp: ~10 scalar special-function calls per element,
// 127 ns each on one core) although nine proposals in ten end up declined
// and leave nothing behind but `old`.  What decides an element is
//     log(u) >= A,   A = ll(new) - ll(old) + prior(new) - prior(old)
//                        + rev - fwd                      (libs/CRP.py:347-383)
// and the decision does not need A's bits, only its value to within the gap
// to log(u), which is typically tens of units.  So every element is
// evaluated HERE first, in plain float64 with the device's own erfc / log /
// inverse normal (a few ulp), together with a bound on everything that can
// separate this A from the host's:
//   * the proposal new = float32(old + sd * ppf(U)) may differ from the
//     host's by float32 ulps (the host's ppf goes through SciPy's log-space
//     formulas): |dA / dtheta| * 4 ulp, with the derivative bounded term by
//     term (likelihood: n1 c / P1 + n0 c / P0, prior: |p-1| / t + |q-1| /
//     (1-t), truncation mass: 1 / (sd Z));
//   * rounding of the sums: 1e-12 of the terms' magnitudes, + 1e-7.
// An element whose log(u) exceeds A by more than that is flagged 0: declined
// whatever the exact arithmetic says; one whose A exceeds log(u) by more than
// that is flagged 2: accepted for certain - the host still has to produce the
// proposal's exact bits (SciPy's ppf) and its prior density, but none of the
// acceptance ratio's terms.  Everything else - in
// doubt, or on a branch this kernel does not model (an interval that does
// not straddle zero, a proposal within ulps of the truncation bounds, a draw
// that is exactly 0) - is flagged 1 and evaluated by the host in full, exactly as
// before.  The chain's bits do not change (tests: screen vs exact decisions
// on random and adversarial batches; every chain test runs through it).
// rev - fwd = log Z(old) - log Z(new): the quadratic terms are equal because
// float32(new - old) == -float32(old - new), the log(sd) terms cancel.
// ---------------------------------------------------------------------------
struct MHScreenConst {
    double sd[8];
    double FP, FN, p, q;
    float tmin32, tmax32;
    int uniform_prior;
};

__global__ __launch_bounds__(256) void k_mh_screen(
    const float *__restrict__ theta, const int *__restrict__ n1,
    const int *__restrict__ n0, const int *__restrict__ sd_idx,
    const double *__restrict__ U, const double *__restrict__ u, long long GM,
    int M, int sum_row, MHScreenConst k, unsigned char *__restrict__ flags,
    float *__restrict__ new_out, DoneSignal done)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    // (no early return: every wave reaches signal_done once, converged)
    if (i < GM) {
        const int g = (int)(i / M);
        const int m = (int)(i - (long long)g * M);
        int c1, c0;
        if (g == sum_row) {             // the merged cluster of a restricted scan
            c1 = n1[m] + n1[M + m];
            c0 = n0[m] + n0[M + m];
        } else {
            c1 = n1[(size_t)g * M + m];
            c0 = n0[(size_t)g * M + m];
        }
        const float old = theta[i];
        const double Ui = U[i], ui = u[i];
        const int si = sd_idx[i];
        unsigned char flag = 1;
        if (si >= 0 && si < 8 && Ui > 0.0 && Ui < 1.0 && ui > 0.0 && ui < 1.0
            && old >= k.tmin32 && old <= k.tmax32 && c1 >= 0 && c0 >= 0) {
            const double sd = k.sd[si];
            const double lo = (double)(k.tmin32 - old) / sd;
            const double hi = (double)(k.tmax32 - old) / sd;
            if (lo <= 0.0 && hi > 0.0) {
                const double Pa = normcdf(lo), Qb = normcdf(-hi);
                const double Z = 1.0 - Pa - Qb;
                const double pl = Pa + Ui * Z;
                double x;
                if (pl <= 0.5)
                    x = normcdfinv(pl);
                else
                    x = -normcdfinv(Qb + (1.0 - Ui) * Z);
                const double xv = x * sd + (double)old;
                const float nw = (float)xv;
                // four float32 steps inside the bounds: the host's proposal may
                // sit an ulp or two away and must still be inside
                const float in_lo = k.tmin32 * (1.0f + 6e-7f);
                const float in_hi = k.tmax32 * (1.0f - 6e-7f);
                if (nw > in_lo && nw < in_hi) {
                    const double ar = (double)(k.tmin32 - nw) / sd;
                    const double br = (double)(k.tmax32 - nw) / sd;
                    const double Zr = 1.0 - normcdf(ar) - normcdf(-br);
                    const double pFN1 = 1.0 - k.FN, pFP0 = 1.0 - k.FP;
                    const double tn = (double)nw, on = (double)(1.0f - nw);
                    const double to = (double)old, oo = (double)(1.0f - old);
                    const double P1n = tn * pFN1 + on * k.FP;
                    const double P0n = tn * k.FN + on * pFP0;
                    const double P1o = to * pFN1 + oo * k.FP;
                    const double P0o = to * k.FN + oo * pFP0;
                    const double lln = (double)c1 * log(P1n) + (double)c0 * log(P0n);
                    const double llo = (double)c1 * log(P1o) + (double)c0 * log(P0o);
                    double prn = 0.0, pro = 0.0, prs = 0.0;
                    if (!k.uniform_prior) {
                        prn = (k.q - 1.0) * log1p(-tn) + (k.p - 1.0) * log(tn);
                        pro = (k.q - 1.0) * log1p(-to) + (k.p - 1.0) * log(to);
                        prs = fabs(k.p - 1.0) / tn + fabs(k.q - 1.0) / (1.0 - tn);
                    }
                    const double A = (lln - llo) + (prn - pro) + (log(Z) - log(Zr));
                    const double cc = fabs(1.0 - k.FN - k.FP);
                    const double sens = (double)c1 * cc / P1n + (double)c0 * cc / P0n
                        + prs + 1.0 / (sd * Zr);
                    const double dtheta = 4.0 * 1.2e-7 * tn + 1e-13;
                    const double margin = sens * dtheta + 1e-7
                        + 1e-12 * (fabs(lln) + fabs(llo) + fabs(prn) + fabs(pro));
                    const double gap = log(ui) - A;
                    if (Z > 0.0 && Zr > 0.0 && gap == gap && gap < INFINITY
                        && gap > -INFINITY) {
                        if (gap > margin) flag = 0;         // declined for certain
                        else if (-gap > margin) flag = 2;   // accepted for certain
                    }
                    // An accepted proposal's float32 bits are certain too when
                    // the float64 value lies clear of the two rounding
                    // boundaries around nw by more than this value and the
                    // host's can be apart: both are the same function of the
                    // same inputs evaluated with errors of a few 1e-15 (here:
                    // erfc and its inverse to a few ulp on |x| <= 10; there:
                    // SciPy's log-space chain, whose largest term - log_ndtr
                    // of a bound ten deviations out, 53 in magnitude - carries
                    // 1e-14), times sd <= 1/2.  4e-13 + 0.4 % of the spacing:
                    // an entry nearer to a boundary (0.8 % of them; all of
                    // those below theta ~ 1e-4, where the spacing itself is
                    // 1e-12) stays with the host's arithmetic.  Flag 3: the
                    // host takes nw and evaluates its prior density only.
                    // (only within four deviations of the old
                    // value - the host evaluates the LEFT form, log_ndtr +
                    // ndtri_exp, even where Phi is close to 1, and the 5e-17
                    // its cancellation leaves in Phi is 5e-17 / pdf(x) in x:
                    // beyond the guard from six deviations on.  Six in a
                    // hundred thousand proposals lie further out than four.)
                    if (flag == 2 && new_out && fabs(x) <= 4.0) {
                        const double up = (double)nextafterf(nw, INFINITY);
                        const double dn = (double)nextafterf(nw, -INFINITY);
                        const double sp = fmax(up - (double)nw, (double)nw - dn);
                        const double guard = 4e-13 + 4e-3 * sp;
                        if (xv < 0.5 * ((double)nw + up) - guard
                            && xv > 0.5 * ((double)nw + dn) + guard) {
                            flag = 3;
                            new_out[i] = nw;
                        }
                    }
                }
            }
        }
        flags[i] = flag;
    }
    signal_done(done);
}

// ---------------------------------------------------------------------------
// host side of the C-ABI
// ---------------------------------------------------------------------------
extern "C" int bnpc_device_count(int *count)
{
    ARGCHK(count, "count is NULL");
    HIPCHK(hipGetDeviceCount(count));
    return 0;
}

extern "C" int bnpc_device_info(int device, char *name, int len, int *cus)
{
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (name && len > 0) {
        strncpy(name, prop.gcnArchName, len - 1);
        name[len - 1] = 0;
    }
    if (cus) *cus = prop.multiProcessorCount;
    return 0;
}

extern "C" int bnpc_device_pci_bus_id(int device, char *bus_id, int len)
{
    ARGCHK(bus_id && len >= 16, "bus_id buffer too small");
    HIPCHK(hipDeviceGetPCIBusId(bus_id, len, device));
    return 0;
}

static int build_view(bnpc_ctx *c, int view, const long long *d_cells,
                      int64_t n)
{
    View &v = c->views[view];
    v.n = n;
    v.nblk = (n + 63) / 64;
    if (n == 0) return 0;
    // + slack: k_ll prefetches one stage past the last block
    if (ensure(v.masks, ((size_t)v.nblk * c->Mpad + 8) * sizeof(ulonglong2)))
        return 1;
    dim3 grid((unsigned)v.nblk, (unsigned)((c->W + 3) / 4));
    BNPC_LAUNCH(k_gather_transpose, grid, dim3(256), 0, c->stream,
                       c->rows, d_cells, (long long)n, c->W,
                       (ulonglong2 *)v.masks.p, c->Mpad);
    HIPCHK(hipGetLastError());
    return 0;
}

// Context from ready bit planes: rows[N][W] of {ones, zeros} words (bits past
// M clear, no bit set in both planes - checked).
static int create_from_planes(int device, int64_t N, int64_t M,
                              const ulonglong2 *rows, bnpc_ctx **out)
{
    HIPCHK(hipSetDevice(device));
    bnpc_ctx *c = new bnpc_ctx();
    read_tunables(c->tun);
    c->device = device;
    c->N = N;
    c->M = M;
    c->W = (int)((M + 63) / 64);
    c->Mpad = c->W * 64;
    c->Mt = (int)((M + 7) / 8 * 8);
    c->cell_n1.assign(N, 0);
    c->cell_n0.assign(N, 0);
    const int tail_bits = (int)(M - (int64_t)(c->W - 1) * 64);  // 1..64
    const unsigned long long tail_mask =
        tail_bits == 64 ? ~0ull : ((1ull << tail_bits) - 1);
    for (int64_t i = 0; i < N; i++) {
        int32_t s1 = 0, s0 = 0;
        const ulonglong2 *r = rows + (size_t)i * c->W;
        for (int w = 0; w < c->W; w++) {
            const unsigned long long ok = (w == c->W - 1) ? tail_mask : ~0ull;
            if ((r[w].x & r[w].y) || ((r[w].x | r[w].y) & ~ok)) {
                delete c;
                bnpc_set_error("bit planes of row %lld are inconsistent",
                               (long long)i);
                return 2;
            }
            s1 += __builtin_popcountll(r[w].x);
            s0 += __builtin_popcountll(r[w].y);
        }
        c->cell_n1[i] = s1;
        c->cell_n0[i] = s0;
    }

#define CRCHK(expr)                                                          \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                              \
            bnpc_set_error("%s failed: %s", #expr, hipGetErrorString(e_));   \
            bnpc_destroy(c);                                                 \
            return 1;                                                        \
        }                                                                    \
    } while (0)
    const size_t bytes = (size_t)N * c->W * sizeof(ulonglong2);
    c->host_rows.assign(rows, rows + (size_t)N * c->W);
    CRCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CRCHK(hipEventCreate(&c->ev0));
    CRCHK(hipEventCreate(&c->ev1));
    CRCHK(hipMalloc((void **)&c->rows, bytes));
    CRCHK(hipMemcpyAsync(c->rows, rows, bytes, hipMemcpyHostToDevice,
                         c->stream));
    if (build_view(c, 0, nullptr, N)) {
        bnpc_destroy(c);
        return 1;
    }
    CRCHK(hipStreamSynchronize(c->stream));
#undef CRCHK
    *out = c;
    return 0;
}

template <typename GetCode>
static int create_impl(int device, int64_t N, int64_t M, GetCode code,
                       bnpc_ctx **out)
{
    ARGCHK(out, "out is NULL");
    ARGCHK(N > 0 && M > 0, "N and M must be positive");
    ARGCHK(M < (1ll << 30) && N < (1ll << 40), "matrix too large");
    *out = nullptr;
    const int W = (int)((M + 63) / 64);
    // pack on the host: 2 bits per entry
    std::vector<ulonglong2> rows((size_t)N * W);
    for (int64_t i = 0; i < N; i++) {
        for (int w = 0; w < W; w++) {
            unsigned long long o = 0, z = 0;
            const int64_t m0 = (int64_t)w * 64;
            const int64_t m1 = std::min<int64_t>(M, m0 + 64);
            for (int64_t m = m0; m < m1; m++) {
                const int v = code(i, m);
                if (v == 1) o |= 1ull << (m - m0);
                else if (v == 0) z |= 1ull << (m - m0);
                else if (v != 3) {
                    bnpc_set_error("data[%lld,%lld] is not 0, 1 or missing",
                                   (long long)i, (long long)m);
                    return 2;
                }
            }
            rows[(size_t)i * W + w] = make_ulonglong2(o, z);
        }
    }
    return create_from_planes(device, N, M, rows.data(), out);
}

extern "C" int bnpc_create_planes(int device, int64_t N, int64_t M,
                                  const uint64_t *planes, bnpc_ctx **out)
{
    ARGCHK(out && planes, "NULL argument");
    ARGCHK(N > 0 && M > 0, "N and M must be positive");
    ARGCHK(M < (1ll << 30) && N < (1ll << 40), "matrix too large");
    *out = nullptr;
    return create_from_planes(device, N, M, (const ulonglong2 *)planes, out);
}

extern "C" int bnpc_create(int device, int64_t N, int64_t M,
                           const double *data_nan, bnpc_ctx **out)
{
    ARGCHK(data_nan, "data is NULL");
    return create_impl(device, N, M, [=](int64_t i, int64_t m) -> int {
        const double v = data_nan[(size_t)i * M + m];
        if (v != v) return 3;
        if (v == 1.0) return 1;
        if (v == 0.0) return 0;
        return -1;
    }, out);
}

extern "C" int bnpc_create_codes(int device, int64_t N, int64_t M,
                                 const int8_t *codes, bnpc_ctx **out)
{
    ARGCHK(codes, "codes is NULL");
    return create_impl(device, N, M, [=](int64_t i, int64_t m) -> int {
        const int v = codes[(size_t)i * M + m];
        return v == 2 ? 1 : v;      // 2 (homozygous) -> 1, dpmmIO.py:93
    }, out);
}

extern "C" int bnpc_destroy(bnpc_ctx *c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->view_cells_pin) (void)hipHostFree(c->view_cells_pin);
    if (c->view_cells_read) (void)hipEventDestroy(c->view_cells_read);
    if (c->done_pin) (void)hipHostFree(c->done_pin);
    if (c->done_count) (void)hipFree(c->done_count);
    DevBuf *bufs[] = {&c->theta, &c->tabs, &c->tab_in, &c->out, &c->cells,
                      &c->tile_out[0], &c->tile_out[1],
                      &c->tile_prior_dev[0], &c->tile_prior_dev[1],
                      &c->chunks, &c->cnt, &c->partial, &c->part,
                      &c->lab_cnt, &c->theta_store, &c->row_idx,
                      &c->side_theta, &c->side_tabs, &c->side_out,
                      &c->side_part, &c->hint_prior, &c->order_dev};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (View &v : c->views)
        if (v.masks.p) (void)hipFree(v.masks.p);
    if (c->rows) (void)hipFree(c->rows);
    pinned_free(c->pin, c->pin_cap);
    if (c->pin_small) (void)hipHostFree(c->pin_small);
    if (c->stage) (void)hipHostFree(c->stage);
    if (c->zc_out) (void)hipHostFree(c->zc_out);
    if (c->hint_pin) (void)hipHostFree(c->hint_pin);
    if (c->hint_prior_pin) (void)hipHostFree(c->hint_prior_pin);
    if (c->order_pin) (void)hipHostFree(c->order_pin);
    mh_ahead_destroy(c);
    if (c->mh_pin) (void)hipHostFree(c->mh_pin);
    for (int p = 0; p < 2; p++)
        if (c->mh_ev[p]) (void)hipEventDestroy(c->mh_ev[p]);
    for (int s = 0; s < BNPC_TILE_SLOTS; s++) {
        pinned_free(c->tile_pin[s], c->tile_cap[s]);
        pinned_free(c->tile_rows[s], c->tile_rows_cap[s]);
        pinned_free(c->tile_cells[s], c->tile_cells_cap[s]);
        pinned_free(c->tile_hint[s], c->tile_hint_cap[s]);
        pinned_free(c->tile_prior[s], c->tile_prior_cap[s]);
        if (c->tile_done[s]) (void)hipEventDestroy(c->tile_done[s]);
    }
    for (int s = 0; s < 2; s++) {
        if (c->tile_summed[s]) (void)hipEventDestroy(c->tile_summed[s]);
        if (c->tile_out_free[s]) (void)hipEventDestroy(c->tile_out_free[s]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->ev_hints) (void)hipEventDestroy(c->ev_hints);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    delete c;
    return 0;
}

extern "C" int bnpc_reload_options(bnpc_ctx *c)
{
    ARGCHK(c, "ctx is NULL");
    read_tunables(c->tun);
    return 0;
}

extern "C" int bnpc_shape(const bnpc_ctx *c, int64_t *N, int64_t *M)
{
    ARGCHK(c, "ctx is NULL");
    if (N) *N = c->N;
    if (M) *M = c->M;
    return 0;
}

// the {ones, zeros} words of one cell's row (bnpc_sweeps.cpp: native births)
const unsigned long long *bnpc_ctx_row(const bnpc_ctx *c, int64_t cell,
                                       int64_t *M, int *W)
{
    if (!c || cell < 0 || cell >= c->N || c->host_rows.empty()) return nullptr;
    if (M) *M = c->M;
    if (W) *W = c->W;
    return (const unsigned long long *)(c->host_rows.data()
                                        + (size_t)cell * c->W);
}

extern "C" int bnpc_cell_counts(bnpc_ctx *c, int32_t *n1, int32_t *n0)
{
    ARGCHK(c && n1 && n0, "NULL argument");
    memcpy(n1, c->cell_n1.data(), c->N * sizeof(int32_t));
    memcpy(n0, c->cell_n0.data(), c->N * sizeof(int32_t));
    return 0;
}

extern "C" int bnpc_view_set(bnpc_ctx *c, int view, const int64_t *cells,
                             int64_t n)
{
    ARGCHK(c, "ctx is NULL");
    ARGCHK(view >= 1 && view < BNPC_MAX_VIEWS, "view out of range");
    ARGCHK(n >= 0 && (n == 0 || cells), "cells is NULL");
    for (int64_t i = 0; i < n; i++)
        ARGCHK(cells[i] >= 0 && cells[i] < c->N, "cell index out of range");
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) {
        c->views[view].n = 0;
        c->views[view].nblk = 0;
        return 0;
    }
    // The cell list travels through a pinned buffer of its own (N entries,
    // read in place by the gather kernel), so the call returns without
    // waiting for the device: what uses the view is queued behind the gather
    // on the same stream, and the buffer is only written again once the
    // gather that read it last has finished (an event; it has, long since,
    // in a split / merge move: a 12 us wait per move otherwise).
    if (!c->view_cells_pin) {
        void *pin = nullptr, *dev = nullptr;
        if (hipHostMalloc(&pin, (size_t)c->N * sizeof(long long),
                          hipHostMallocDefault) == hipSuccess
            && hipHostGetDevicePointer(&dev, pin, 0) == hipSuccess
            && hipEventCreateWithFlags(&c->view_cells_read,
                                       hipEventDisableTiming) == hipSuccess) {
            c->view_cells_pin = pin;
            c->view_cells_dev = (const long long *)dev;
        } else {
            (void)hipGetLastError();
            if (pin) (void)hipHostFree(pin);
        }
    }
    if (c->view_cells_pin && n <= c->N && !c->any_tile_pending()) {
        if (c->view_cells_busy) {
            HIPCHK(hipEventSynchronize(c->view_cells_read));
            c->view_cells_busy = false;
        }
        memcpy(c->view_cells_pin, cells, n * sizeof(long long));
        if (build_view(c, view, c->view_cells_dev, n)) return 1;
        HIPCHK(hipEventRecord(c->view_cells_read, c->stream));
        c->view_cells_busy = true;
        return 0;
    }
    if (arena_reset(c)) return 1;
    const long long *d_cells = (const long long *)stage_in_place(
        c, cells, n * sizeof(long long));
    if (!d_cells) {
        if (ensure(c->cells, n * sizeof(long long))) return 1;
        if (h2d(c, c->cells.p, cells, n * sizeof(long long))) return 1;
        d_cells = (const long long *)c->cells.p;
    }
    if (build_view(c, view, d_cells, n)) return 1;
    // the caller's buffer is only borrowed: finish the copy before returning
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

static int ensure_host(void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return 0;
    pinned_free(*p, *cap);
    *p = nullptr;
    *cap = 0;
    // tiles of a sweep are sized to a byte budget: little slack is needed
    return pinned_alloc(p, cap, bytes + bytes / 16 + 4096);
}

// bnpc_view_set for a tile of a tiled sweep: the cell list is staged in the
// tile slot's own pinned buffer (read in place by the gather kernel) and
// NOTHING is waited for - the stream may hold the sums of the tiles issued
// before, which a synchronisation here would serialise with the host.  The
// view is for work issued behind it on the context's stream
// (bnpc_ll_rows_issue on the same slot).
extern "C" int bnpc_view_set_slot(bnpc_ctx *c, int view, const int64_t *cells,
                                  int64_t n, int slot)
{
    ARGCHK(c, "ctx is NULL");
    ARGCHK(view >= 1 && view < BNPC_MAX_VIEWS, "view out of range");
    ARGCHK(slot >= 0 && slot < BNPC_TILE_SLOTS, "slot out of range");
    ARGCHK(n > 0 && cells, "empty cell list");
    ARGCHK(!c->tile_pending[slot], "slot has an unconsumed tile");
    for (int64_t i = 0; i < n; i++)
        ARGCHK(cells[i] >= 0 && cells[i] < c->N, "cell index out of range");
    HIPCHK(hipSetDevice(c->device));
    if (ensure_lanes(c)) return 1;
    // (for all N cells at once: tiles grow as the clusters die, and growing
    // a pinned buffer means hipHostFree - a device-wide synchronisation of
    // ~5 ms in the middle of the pipeline; measured: 33 of them, 0.17 s of a
    // config-5 first sweep)
    if (ensure_host(&c->tile_cells[slot], &c->tile_cells_cap[slot],
                    std::max<int64_t>(n, c->N) * sizeof(long long)))
        return 1;
    memcpy(c->tile_cells[slot], cells, n * sizeof(long long));
    void *d = nullptr;
    HIPCHK(hipHostGetDevicePointer(&d, c->tile_cells[slot], 0));
    // Tiles grow as the clusters die (1024 cells, then 1536, 2048, ...), and
    // growing a device buffer means hipFree - a device-wide synchronisation
    // in the middle of the pipeline.  Room for 4 x the first tile, at least
    // 16384 cells (all cells if there are fewer), is taken at once.
    View &v = c->views[view];
    int64_t room = std::max<int64_t>(4 * n, 16384);
    room = std::min<int64_t>(std::max<int64_t>(room, n), std::max(c->N, n));
    if (ensure(v.masks, ((size_t)((room + 63) / 64) * c->Mpad + 8)
                            * sizeof(ulonglong2)))
        return 1;
    return build_view(c, view, (const long long *)d, n);
}

extern "C" int bnpc_view_size(const bnpc_ctx *c, int view, int64_t *n)
{
    ARGCHK(c && n, "NULL argument");
    ARGCHK(view >= 0 && view < BNPC_MAX_VIEWS, "view out of range");
    *n = c->views[view].n;
    return 0;
}

// clusters per wave: least padded work, weighted by the scalar-pipe overhead
// that a wider tile amortises
static int pick_kw(int64_t K)
{
    const int kws[] = {8, 4, 2, 1};
    int best = 1;
    double best_cost = 1e300;
    for (int kw : kws) {
        const double cost = (double)((K + kw - 1) / kw * kw) * (1.0 + 1.0 / kw);
        if (cost < best_cost) {
            best_cost = cost;
            best = kw;
        }
    }
    return best;
}

// Mutation split of a launch with too few waves to fill the chip and hide
// its table loads.  `waves` = slot blocks x cluster groups.  The split kernel
// gives the 4 waves of a workgroup 4 consecutive chunks, so the chunk count
// is a multiple of 4 (2 or 3 chunks leave waves idle: 50000 x 5000 x 30 in 2
// chunks 1172 us, in 4 chunks 688 us) and exactly 4 needs no combine pass.
// Measured (tools/msplit_tune_big.py): 50000 x 5000: K = 53 unsplit 1322 us,
// 4 chunks 1117 us; K = 100 (10166 waves) unsplit 2060 us, 4 chunks 2105 us
// -> no split from 8192 waves on; small launches want more chunks (10000 x
// 2000 x 40, 785 waves: 6 chunks 121 us, 21 chunks 100 us, 42 chunks 92 us;
// 5000 x 1000 x 14, 158 waves: 25 / 42 / 63 chunks 25.0 / 23.3 / 27.1 us).  Chunks
// hold at least 16 mutations, a multiple of 8 (stage sizes divide 8).
static int64_t msplit_limit(const Tunables &tun)
{
    (void)tun;
    return 8192;
}

static void pick_msplit(const Tunables &tun, int64_t waves, int Mt,
                        bool allowed, int *MS, int *m_chunk)
{
    *MS = 1;
    *m_chunk = Mt;
    if (!allowed || waves >= msplit_limit(tun)) return;
    const int64_t target = waves >= 512 && waves < 2048 ? 16384 : 8192;
    int64_t want = (target + waves - 1) / waves;
    want = (want + 3) / 4 * 4;
    if (waves >= 1024) {
        // round 5 (tools/msplit_sweep.py, profiles/r05/msplit_sweep.md): from
        // a thousand waves on a launch wants ~4000 workgroups of the split
        // kernel (4 chunks each) as long as a chunk keeps 256 mutations -
        // 50000 x 5000 x 20: 4 chunks 565 us, 16 chunks 440; x 50: 1001 ->
        // 983 with 8; 5000 x 1000 x 200: 12 chunks 92 us, 4 chunks 79
        want = ((32768 + waves - 1) / waves + 3) / 4 * 4;
        const int64_t by_len = (int64_t)Mt / 256 / 4 * 4;
        if (want > by_len) want = by_len;
        if (want < 4) want = 4;
    }
    const int64_t cap = MSPLIT_MAX;
    if (tun.msplit_chunks > 0) want = (tun.msplit_chunks + 3) / 4 * 4;
    if (want > cap) want = cap;
    int chunk = (int)((Mt + want - 1) / want);
    chunk = (chunk + 7) / 8 * 8;
    if (chunk < 16) chunk = 16;
    const int ms = (Mt + chunk - 1) / chunk;
    if (ms >= 2) {
        *MS = ms;
        *m_chunk = chunk;
    }
}


// the cells x clusters x mutations launch itself (tables are resident)
template <int KW>
static int issue_ll(bnpc_ctx *c, const View &v, int64_t K, int64_t ldo,
                    double *d_out, int MS, int m_chunk)
{
    const int64_t G = (K + KW - 1) / KW;
    const int64_t nwg = ((v.nblk + 3) / 4) * G * MS;
    ARGCHK(nwg < (1ll << 31), "launch too large");
    const int xcd = 1;      // XCD-aware tile order (ll_tile_coords)
    // L2 prefetch of the mask / table streams: when they cannot all sit in
    // one XCD's 4 MiB L2, and on split launches
    const size_t stream_bytes = (size_t)v.nblk * c->Mpad * 16
        + (size_t)G * c->Mt * 2 * KW * sizeof(double) / 8;
    const int pf = stream_bytes > (3u << 20) || MS > 1;
    double *dst = d_out;
    // partial planes: the hand-placed kernel sums 4 chunks inside a
    // workgroup (ceil(MS / 4) planes, none for exactly 4), k_ll writes MS
    const int need_planes = MS > 1 ? (KW == 8 ? (MS + 3) / 4 : MS) : 1;
    if (need_planes > 1) {
        if (ensure(c->part, (size_t)need_planes * v.n * K * sizeof(double)))
            return 1;
        dst = (double *)c->part.p;
    }
    const int64_t wg2 = ((v.nblk + 7) / 8) * G * MS;
    // mutation-split form of the hand-placed kernel: a workgroup's 4 waves
    // take 4 chunks of the same blocks -> ceil(MS / 4) partial planes
    const int MSq = (MS + 3) / 4;
    const int64_t split2 = ((v.nblk + 1) / 2) * G * MSq;
    const int64_t split1 = v.nblk * G * MSq;
    int planes = MS;
#define LAUNCH_ASM(CB_, SPLIT_, GRID_)                                        \
    BNPC_LAUNCH((k_ll8_asm<CB_, SPLIT_>), dim3((unsigned)(GRID_)),     \
                       dim3(256), 0, c->stream,                              \
                       (const ulonglong2 *)v.masks.p, c->Mpad, c->Mt,        \
                       (long long)v.n, (long long)v.nblk,                    \
                       (const double *)c->tabs.p, (int)K, (long long)ldo,    \
                       dst, xcd, MS, m_chunk,                                \
                       pf ? (const ulonglong2 *)v.masks.p : nullptr,         \
                       pf ? (const double *)c->tabs.p : nullptr,             \
                       (SPLIT_) && need_planes == 1 ? sig : no_sig)
    // the completion word rides on the LAST kernel of the evaluation: the
    // combine pass, or the split sums kernel itself when a workgroup holds
    // the whole sum (4 chunks); other forms leave it unattached
    const DoneSignal sig = c->sig_next, no_sig = {nullptr, nullptr, 0};
    c->sig_next = no_sig;
    c->sig_attached = false;
    const char *combine = "";
    if (KW == 8 && wg2 >= ASM2_MIN_WGS) {
        if (MS > 1) {
            if (MSq == 1) dst = d_out;  // the workgroup already holds the sum
            LAUNCH_ASM(2, true, split2);
            planes = MSq;
            combine = MSq > 1 ? " + k_ll_combine" : "";
            snprintf(c->last_name, sizeof(c->last_name),
                     "k_ll8_asm<2, true>%s", combine);
        } else if (c->Mt >= LDS_TABLE_MIN_M && wg2 >= LDS_TABLE_MIN_WGS) {
            // long mutation streams, enough workgroups to keep every CU's
            // LDS pipeline full: the table through LDS (measured, % of the
            // two-add issue peak, scalar path -> LDS path: 1024 x 5000 x
            // 31608 68 -> 80, 50000 x 5000 x 512 73 -> 78; but 1024 x 2000
            // x 31608 76 -> 78 at 1024 rows and 89 -> 81 at 2048+, x 1000
            // 93 -> 81, and 50000 x 5000 x 100 with its 1300 workgroups
            // 57 -> 48: tools/tile_shape_bench.py, profiles/r04)
            BNPC_LAUNCH((k_ll8_lds<2>), dim3((unsigned)wg2), dim3(256),
                               0, c->stream, (const ulonglong2 *)v.masks.p,
                               c->Mpad, c->Mt, (long long)v.n,
                               (long long)v.nblk, (const double *)c->tabs.p,
                               (int)K, (long long)ldo, dst, xcd);
            snprintf(c->last_name, sizeof(c->last_name), "k_ll8_lds<2>");
        } else {
            LAUNCH_ASM(2, false, wg2);
            snprintf(c->last_name, sizeof(c->last_name),
                     "k_ll8_asm<2, false>");
        }
    } else if (KW == 8) {
        if (MS > 1) {
            if (MSq == 1) dst = d_out;
            LAUNCH_ASM(1, true, split1);
            planes = MSq;
            combine = MSq > 1 ? " + k_ll_combine" : "";
            snprintf(c->last_name, sizeof(c->last_name),
                     "k_ll8_asm<1, true>%s", combine);
        } else {
            LAUNCH_ASM(1, false, nwg);
            snprintf(c->last_name, sizeof(c->last_name),
                     "k_ll8_asm<1, false>");
        }
    }
#undef LAUNCH_ASM
    else {
        snprintf(c->last_name, sizeof(c->last_name), "k_ll<%d>%s", KW,
                 MS > 1 ? " + k_ll_combine" : "");
        BNPC_LAUNCH(k_ll<KW>, dim3((unsigned)nwg), dim3(256), 0,
                           c->stream, (const ulonglong2 *)v.masks.p, c->Mpad,
                           c->Mt, (long long)v.n, (long long)v.nblk,
                           (const double *)c->tabs.p, (int)K, (long long)ldo,
                           dst, xcd, MS, m_chunk);
    }
    if (MS > 1 && planes > 1) {
        HIPCHK(hipGetLastError());
        const long long total = (long long)v.n * K;
        BNPC_LAUNCH(k_ll_combine, dim3((unsigned)((total + 255) / 256)),
                           dim3(256), 0, c->stream, (const double *)c->part.p,
                           (long long)v.n, (int)K, planes, (long long)ldo,
                           d_out, sig);
        c->sig_attached = sig.count != nullptr;
    } else if (KW == 8 && MS > 1 && need_planes == 1) {
        c->sig_attached = sig.count != nullptr;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

template <int KW>
static int launch_ll(bnpc_ctx *c, const View &v, int64_t K, int64_t ldo,
                     bool from_theta, double FP, double FN, double *d_out,
                     int MS, int m_chunk)
{
    const int64_t G = (K + KW - 1) / KW;
    ARGCHK(G <= 65535, "too many cluster groups for one launch");
    // + slack: k_ll prefetches one stage past the last group
    if (ensure(c->tabs, ((size_t)G * c->Mt + 8) * 2 * KW * sizeof(double)))
        return 1;
    // k_tables_theta: 256 / (2 KW) mutations per workgroup; k_tables_relayout:
    // 256
    dim3 tgrid((unsigned)((c->Mt + 255) / 256), (unsigned)G);
    constexpr int TMB = 256 / (2 * KW);
    dim3 tgrid_e((unsigned)((c->Mt + TMB - 1) / TMB), (unsigned)G);
    if (from_theta && G * KW * (int64_t)c->Mt
            <= TABLES_FLAT_MAX) {
        const int64_t threads = G * KW * (int64_t)c->Mt;
        BNPC_LAUNCH(k_tables_theta_flat<KW>,
                           dim3((unsigned)((threads + 255) / 256)), dim3(256),
                           0, c->stream,
                           c->use_rows ? (const float *)c->theta_store.p
                                       : c->theta_src,
                           c->use_rows, (int)K, (int)c->M, c->Mt, (int)G, FP,
                           FN, (double *)c->tabs.p);
    } else if (from_theta)
        BNPC_LAUNCH(k_tables_theta<KW>, tgrid_e, dim3(256), 0, c->stream,
                           c->use_rows ? (const float *)c->theta_store.p
                                       : c->theta_src,
                           c->use_rows, (int)K, (int)c->M, c->Mt, FP, FN,
                           (double *)c->tabs.p);
    else
        BNPC_LAUNCH(k_tables_relayout<KW>, tgrid, dim3(256), 0,
                           c->stream, c->tab_src,
                           c->tab_src + (size_t)K * c->M,
                           (int)K, (int)c->M, c->Mt, (double *)c->tabs.p);
    HIPCHK(hipGetLastError());
    if (issue_ll<KW>(c, v, K, ldo, d_out, MS, m_chunk)) return 1;
    return 0;
}

// K2p on the caller's tables (c->tab_src: L1 [K][M] then L0 [K][M])
static int issue_seqp(bnpc_ctx *c, const View &v, int64_t K, int64_t ldo,
                      double *d_out)
{
    static bool lds_raised = false;
    if (!lds_raised) {
        HIPCHK(hipFuncSetAttribute((const void *)k_ll_seqp,
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)SEQP_LDS));
        lds_raised = true;
    }
    const double *tabs = c->tab_src;
    const size_t n2 = (size_t)K * c->M;         // double2 elements: 2 K M / 2
    if ((const void *)tabs != c->tab_in.p) {
        // staged in the pinned arena: every chain of the launch would pull
        // its table over the host link again - one copy kernel instead
        if (ensure(c->tab_in, 2 * n2 * sizeof(double))) return 1;
        BNPC_LAUNCH(k_stage_copy, dim3((unsigned)((n2 + 255) / 256)),
                           dim3(256), 0, c->stream, (const double2 *)tabs,
                           (double2 *)c->tab_in.p, (long long)n2);
        tabs = (const double *)c->tab_in.p;
    }
    dim3 grid((unsigned)v.nblk, (unsigned)K);
    snprintf(c->last_name, sizeof(c->last_name), "k_ll_seqp");
    BNPC_LAUNCH(k_ll_seqp, grid, dim3(256), SEQP_LDS, c->stream,
                       (const ulonglong2 *)v.masks.p, c->Mpad, (int)c->M,
                       (long long)v.n, (long long)v.nblk, tabs,
                       tabs + (size_t)K * c->M, (int)K, (long long)ldo, d_out,
                       c->sig_next);
    c->sig_attached = c->sig_next.count != nullptr;
    c->sig_next = DoneSignal{nullptr, nullptr, 0};
    HIPCHK(hipGetLastError());
    return 0;
}

static int ll_common(bnpc_ctx *c, int view, int64_t K, int64_t ldo,
                     bool from_theta, double FP, double FN, double *out)
{
    const View &v = c->views[view];
    if (v.n == 0 || K == 0) return 0;
    if (ldo == 0) ldo = K;
    ARGCHK(ldo >= K, "ldo smaller than K");
    c->pin_lazy_bytes = 0;      // (a matrix left on the device is given up)
    c->hint_later.state = 0;    // (... and so are hints not yet launched)
    const size_t out_bytes = (size_t)v.n * ldo * sizeof(double);
    // a small result that the caller wants on the host is written by the
    // kernels straight into pinned host memory (no copy-engine launch)
    void *zc_dev = nullptr;
    void *zc_host = out ? zc_result(c, out_bytes, &zc_dev) : nullptr;
    if (!zc_host && !c->dst_override && ensure(c->out, out_bytes)) return 1;
    int kw = pick_kw(K);
    // A launch that will be split over the mutations runs the hand-placed
    // 8-cluster kernel whatever K is: its workgroups reduce 4 chunks through
    // LDS, which beats the narrower C++ tilings from K = 2 on (measured
    // K = 2..12: 11-17 us against 11-25 us; profiles/r01/small_launch_study.md)
    if (K >= 2 && from_theta && c->tun.msplit
        && v.nblk * ((K + 7) / 8) < msplit_limit(c->tun))
        kw = 8;
    if (c->tun.force_kw) kw = c->tun.force_kw;
    // Sums over caller-built tables keep the strict mutation order (they are
    // the bit-exact path); device-built tables may split the mutations.
    int MS, m_chunk;
    pick_msplit(c->tun, v.nblk * ((K + kw - 1) / kw), c->Mt,
                from_theta && c->tun.msplit, &MS, &m_chunk);
    double *d_out = zc_host ? (double *)zc_dev : (double *)c->out.p;
    if (c->dst_override) d_out = c->dst_override;
    // a result written in place for the host is waited for through the
    // completion word of the evaluation's last kernel
    unsigned done_seq = 0;
    c->sig_attached = false;
    if (zc_host) c->sig_next = make_signal(c, 0, &done_seq);
    int rc;
    // caller-built tables: the one-chain-on-four-SIMDs pipeline; a forced
    // cluster tile (BNPC_KW, tests) takes them through k_ll<KW> instead
    if (!from_theta && !c->tun.force_kw && K <= 65535) {
        rc = issue_seqp(c, v, K, ldo, d_out);
        kw = -1;
        MS = 1;
    } else
    switch (kw) {
    case 8: rc = launch_ll<8>(c, v, K, ldo, from_theta, FP, FN, d_out, MS, m_chunk); break;
    case 4: rc = launch_ll<4>(c, v, K, ldo, from_theta, FP, FN, d_out, MS, m_chunk); break;
    case 2: rc = launch_ll<2>(c, v, K, ldo, from_theta, FP, FN, d_out, MS, m_chunk); break;
    default: rc = launch_ll<1>(c, v, K, ldo, from_theta, FP, FN, d_out, MS, m_chunk); break;
    }
    if (rc) return rc;
    c->last_ms = MS;
    c->last_mchunk = m_chunk;
    c->last_from_theta = from_theta;
    c->last_FP = FP;
    c->last_FN = FN;
    c->last_kw = kw;
    c->last_view = view;
    c->last_K = K;
    c->last_ldo = ldo;
    c->last_out = d_out;
    c->sig_next = DoneSignal{nullptr, nullptr, 0};
    if (zc_host && c->defer_next) {
        c->defer.zc_host = zc_host;
        c->defer.seq = c->sig_attached ? done_seq : 0;
        c->defer.out = out;
        c->defer.bytes = out_bytes;
        c->defer.n = v.n;
        c->defer.K = K;
        c->defer.ldo = ldo;
        c->defer_set = true;
        return 0;
    }
    if (zc_host) {
        if (int rw = wait_done(c, 0, c->sig_attached ? done_seq : 0))
            return rw;
        // columns K..ldo of the caller's rows are not ours to touch
        if (ldo == K) {
            memcpy(out, zc_host, out_bytes);
        } else {
            for (int64_t r = 0; r < v.n; r++)
                memcpy(out + r * ldo, (const double *)zc_host + r * ldo,
                       (size_t)K * sizeof(double));
        }
    } else if (out) {
        HIPCHK(hipMemcpyAsync(out, c->out.p, out_bytes, hipMemcpyDeviceToHost,
                              c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// `drain`: synchronise before returning even when nothing is fetched (the
// staged parameters must have left the arena); bnpc_ll_theta_pinned passes
// false because its own copy + synchronisation follow on the same stream.
static int ll_theta_impl(bnpc_ctx *c, int view, const float *theta, int64_t K,
                         double FP, double FN, double *out, int64_t ldo,
                         bool drain)
{
    ARGCHK(c, "ctx is NULL");
    ARGCHK(view >= 0 && view < BNPC_MAX_VIEWS, "view out of range");
    ARGCHK(K >= 0 && (K == 0 || theta), "theta is NULL");
    ARGCHK(FP > 0.0 && FP < 1.0 && FN > 0.0 && FN < 1.0,
           "error rates must lie in (0, 1)");
    HIPCHK(hipSetDevice(c->device));
    if (K == 0) return 0;
    SideLane lane(c);
    const size_t bytes = (size_t)K * c->M * sizeof(float);
    if (arena_reset(c)) return 1;
    void *slot = nullptr;
    c->theta_src = (const float *)stage_in_place(c, theta, bytes);
    if (c->theta_src) {
        slot = (void *)c->theta_src;    // staged: drain before returning
    } else {
        if (ensure(c->theta, bytes)) return 1;
        slot = stage_slot(c, bytes);
        if (slot) memcpy(slot, theta, bytes);
        HIPCHK(hipMemcpyAsync(c->theta.p, slot ? slot : (const void *)theta,
                              bytes, hipMemcpyHostToDevice, c->stream));
        c->theta_src = (const float *)c->theta.p;
    }
    int rc = ll_common(c, view, K, ldo, true, FP, FN, out);
    // staged parameters / the side lane: nothing may be in flight on return
    if (((slot && drain) || lane.on) && !out && rc == 0)
        HIPCHK(hipStreamSynchronize(c->stream));
    return rc;
}

extern "C" int bnpc_ll_theta(bnpc_ctx *c, int view, const float *theta,
                             int64_t K, double FP, double FN, double *out,
                             int64_t ldo)
{
    return ll_theta_impl(c, view, theta, K, FP, FN, out, ldo, true);
}

// bnpc_ll_theta in two halves (bnpc_internal.h): _begin queues the evaluation
// and returns where its result is small enough to be written in place for the
// host (otherwise it is complete on return, as bnpc_ll_theta); _end waits for
// the completion word and fills `out`.  No other call on the context in
// between (the parameters stay staged in the arena).
int bnpc_ll_theta_begin(bnpc_ctx *c, int view, const float *theta, int64_t K,
                        double FP, double FN, double *out, int64_t ldo)
{
    ARGCHK(c && out, "NULL argument");
    c->defer_set = false;
    c->defer_next = !c->any_tile_pending();
    const int rc = ll_theta_impl(c, view, theta, K, FP, FN, out, ldo, true);
    c->defer_next = false;
    if (rc) c->defer_set = false;
    return rc;
}

int bnpc_ll_theta_end(bnpc_ctx *c)
{
    ARGCHK(c, "ctx is NULL");
    if (!c->defer_set) return 0;
    c->defer_set = false;
    HIPCHK(hipSetDevice(c->device));
    if (int rw = wait_done(c, 0, c->defer.seq)) return rw;
    const int64_t K = c->defer.K, ldo = c->defer.ldo;
    if (ldo == K) {
        memcpy(c->defer.out, c->defer.zc_host, c->defer.bytes);
    } else {
        for (int64_t r = 0; r < c->defer.n; r++)
            memcpy(c->defer.out + r * ldo,
                   (const double *)c->defer.zc_host + r * ldo,
                   (size_t)K * sizeof(double));
    }
    return 0;
}

// Same as bnpc_ll_theta, but the result lands in a context-owned PINNED host
// buffer (DMA straight from the device, no bounce through pageable staging)
// that the caller reads - and may write - in place.
extern "C" int bnpc_ll_theta_pinned(bnpc_ctx *c, int view, const float *theta,
                                    int64_t K, double FP, double FN,
                                    int64_t ldo, double **host)
{
    ARGCHK(c && host, "NULL argument");
    ARGCHK(view >= 0 && view < BNPC_MAX_VIEWS, "view out of range");
    ARGCHK(K > 0 && theta, "theta is NULL");
    if (ldo == 0) ldo = K;
    ARGCHK(ldo >= K, "ldo smaller than K");
    *host = nullptr;
    ARGCHK(!c->any_tile_pending(),
           "not available while an issued tile is in flight");
    const size_t bytes = (size_t)c->views[view].n * ldo * sizeof(double);
    if (bytes && ensure_pin(c, bytes)) return 1;
    int rc = ll_theta_impl(c, view, theta, K, FP, FN, nullptr, ldo, false);
    if (rc) return rc;
    if (bytes == 0) {
        HIPCHK(hipStreamSynchronize(c->stream));
        return 0;
    }
    HIPCHK(hipMemcpyAsync(c->pin, c->out.p, bytes, hipMemcpyDeviceToHost,
                          c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *host = (double *)c->pin;
    return 0;
}

// the hint kernel over the n rows of the matrix at d_ll (row stride ldo): up
// to 64 columns one thread per row with the priors as kernel arguments, more
// one wave per row with the priors in c->hint_prior.  order != NULL: record r
// is made from row order[r] (the hints in visiting order).
static int hint_launch(bnpc_ctx *c, const double *d_ll, int64_t n, int64_t K,
                       int64_t ldo, const Top2Prior &pr, void *hint_dev,
                       double *rows_dev, const long long *order)
{
    if (K <= 64)
        BNPC_LAUNCH(k_row_top2, dim3((unsigned)((n + 255) / 256)),
                           dim3(256), 0, c->stream, d_ll, (long long)n,
                           (long long)ldo, (int)K, pr, (bnpc_top2 *)hint_dev,
                           rows_dev, order);
    else
        BNPC_LAUNCH(k_row_top4_wave, dim3((unsigned)((n + 3) / 4)),
                           dim3(256), 0, c->stream, d_ll, (long long)n,
                           (long long)ldo, (int)K,
                           (const double *)c->hint_prior.p,
                           (bnpc_top2 *)hint_dev, rows_dev,
                           (int)HINT_THROUGH_MAX, order);
    HIPCHK(hipGetLastError());
    return 0;
}

// what follows the hint kernel: the event the host waits for, the copy of the
// matrix where the sweep is known to need it
static int hints_queued(bnpc_ctx *c, int64_t K, size_t bytes, bool wait)
{
    // a row of thousands of columns is not written through: the sweep that
    // meets such a matrix (a first sweep: nothing is decided before its first
    // births) reads it from its first cell on - the copy is queued at once
    if (K > HINT_THROUGH_MAX) c->matrix_eager = true;
    // the caller gets the hints now; the matrix stays on the device and is
    // copied if and when the sweep first needs a row of it (bnpc_matrix_wait)
    // - a converged sweep never does
    if (!c->ev_hints)
        HIPCHK(hipEventCreateWithFlags(&c->ev_hints, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->ev_hints, c->stream));
    if (c->matrix_eager) {
        HIPCHK(hipMemcpyAsync(c->pin, c->out.p, bytes, hipMemcpyDeviceToHost,
                              c->stream));
        c->pin_copy_queued = true;
    }
    if (wait) HIPCHK(hipEventSynchronize(c->ev_hints));
    c->pin_lazy_bytes = bytes;
    return 0;
}

// later: the sums only; the hint kernel follows with bnpc_hints_in_order_issue
static int ll_top2_impl(bnpc_ctx *c, int view, const float *theta, int64_t K,
                        double FP, double FN, int64_t ldo,
                        const double *col_prior, double **host,
                        bnpc_top2 **top2, bool wait, bool later = false)
{
    ARGCHK(c && host && col_prior && (top2 || later), "NULL argument");
    ARGCHK(K > 0 && K <= HINT_COLS_MAX, "K out of range for the hint");
    ARGCHK(view >= 0 && view < BNPC_MAX_VIEWS, "view out of range");
    if (top2) *top2 = nullptr;
    c->hint_later.state = 0;
    if (ldo == 0) ldo = K;
    const int64_t n = c->views[view].n;
    // the hints of all slots, written in place into pinned host memory of
    // their own: they must outlive the calls made DURING the sweep (a column
    // for a cluster opened half-way), which use the shared result buffer
    void *zc_dev = nullptr;
    bnpc_top2 *hint = nullptr;
    if (n && c->tun.zero_copy) {
        const size_t need = (size_t)n * sizeof(bnpc_top2);
        if (need > c->hint_cap) {
            if (c->hint_pin) HIPCHK(hipHostFree(c->hint_pin));
            c->hint_pin = nullptr;
            c->hint_cap = 0;
            HIPCHK(hipHostMalloc(&c->hint_pin, need + need / 4,
                                 hipHostMallocDefault));
            c->hint_cap = need + need / 4;
        }
        if (hipHostGetDevicePointer(&zc_dev, c->hint_pin, 0) == hipSuccess)
            hint = (bnpc_top2 *)c->hint_pin;
    }
    // the matrix itself as bnpc_ll_theta_pinned, minus its final wait
    ARGCHK(theta, "theta is NULL");
    ARGCHK(ldo >= K, "ldo smaller than K");
    *host = nullptr;
    ARGCHK(!c->any_tile_pending(),
           "not available while an issued tile is in flight");
    const size_t bytes = (size_t)n * ldo * sizeof(double);
    if (bytes && ensure_pin(c, bytes)) return 1;
    // did the previous hinted sweep read its matrix?
    c->matrix_eager = c->lazy_fetched;
    c->lazy_fetched = false;
    // the host's copy of the matrix as the device sees it (rows the sweep is
    // going to scan are written through by the hint kernel)
    double *rows_dev = nullptr;
    Top2Prior pr = {};
    if (hint && bytes) {
        void *pin_dev = nullptr;
        if (hipHostGetDevicePointer(&pin_dev, c->pin, 0) == hipSuccess)
            rows_dev = (double *)pin_dev;
        else
            (void)hipGetLastError();        // not mapped: no write-through
        if (K <= 64) {
            for (int k = 0; k < 64; k++) pr.v[k] = k < K ? col_prior[k] : 0.0;
        } else {
            // more columns than fit the kernel's arguments: the priors in
            // device memory - every wave reads all of them, which the host
            // link should see once: staged in a pinned block of their own
            // (the arena is reset by the evaluation below) and moved by a
            // copy kernel ahead of everything else
            const size_t pb = (size_t)K * sizeof(double);
            const size_t pb2 = (pb + 15) & ~(size_t)15;
            if (ensure(c->hint_prior, pb2)) return 1;
            if (!c->hint_prior_pin)
                HIPCHK(hipHostMalloc(&c->hint_prior_pin,
                                     ((size_t)HINT_COLS_MAX + 1) * 8,
                                     hipHostMallocDefault));
            void *pp_dev = nullptr;
            HIPCHK(hipHostGetDevicePointer(&pp_dev, c->hint_prior_pin, 0));
            memcpy(c->hint_prior_pin, col_prior, pb);
            const long long n2 = (long long)(pb2 / 16);
            BNPC_LAUNCH(k_stage_copy, dim3((unsigned)((n2 + 255) / 256)),
                               dim3(256), 0, c->stream, (const double2 *)pp_dev,
                               (double2 *)c->hint_prior.p, n2);
            HIPCHK(hipGetLastError());
        }
    }
    int rc = ll_theta_impl(c, view, theta, K, FP, FN, nullptr, ldo, false);
    if (rc) return rc;
    if (bytes == 0) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (later) c->hint_later.state = 2;
        return 0;
    }
    *host = (double *)c->pin;
    if (hint && later) {
        c->hint_later.state = 1;
        c->hint_later.n = n;
        c->hint_later.K = K;
        c->hint_later.ldo = ldo;
        c->hint_later.bytes = bytes;
        c->hint_later.prior = pr;
        c->hint_later.hint_dev = zc_dev;
        c->hint_later.rows_dev = rows_dev;
        return 0;
    }
    if (hint) {
        if (int rh = hint_launch(c, (const double *)c->out.p, n, K, ldo, pr,
                                 zc_dev, rows_dev, nullptr))
            return rh;
        if (int rq = hints_queued(c, K, bytes, wait)) return rq;
    } else {
        HIPCHK(hipMemcpyAsync(c->pin, c->out.p, bytes, hipMemcpyDeviceToHost,
                              c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (later) c->hint_later.state = 2;
    }
    if (top2) *top2 = hint;
    return 0;
}

extern "C" int bnpc_ll_theta_pinned_top2(bnpc_ctx *c, int view,
                                         const float *theta, int64_t K,
                                         double FP, double FN, int64_t ldo,
                                         const double *col_prior,
                                         double **host, bnpc_top2 **top2)
{
    return ll_top2_impl(c, view, theta, K, FP, FN, ldo, col_prior, host, top2,
                        true);
}

// The same, returning as soon as the work is queued when the hints are
// written in place (the parameters were staged: nothing of the caller's is
// borrowed): the caller prepares the sweep - its permutation, its tables -
// under the launch and calls bnpc_sync before reading the hints.
extern "C" int bnpc_ll_theta_pinned_top2_issue(bnpc_ctx *c, int view,
                                               const float *theta, int64_t K,
                                               double FP, double FN,
                                               int64_t ldo,
                                               const double *col_prior,
                                               double **host,
                                               bnpc_top2 **top2)
{
    return ll_top2_impl(c, view, theta, K, FP, FN, ldo, col_prior, host, top2,
                        false);
}

// The matrix of the last bnpc_ll_theta_pinned_top2 is complete on return.
extern "C" int bnpc_matrix_wait(bnpc_ctx *c)
{
    ARGCHK(c, "ctx is NULL");
    if (!c->pin_lazy_bytes) return 0;
    HIPCHK(hipSetDevice(c->device));
    const size_t bytes = c->pin_lazy_bytes;
    c->pin_lazy_bytes = 0;
    c->lazy_fetched = true;
    if (!c->pin_copy_queued)
        HIPCHK(hipMemcpyAsync(c->pin, c->out.p, bytes, hipMemcpyDeviceToHost,
                              c->stream));
    c->pin_copy_queued = false;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// bnpc_ll_theta_pinned_top2_issue in two halves, for a sweep that reads its
// hints IN VISITING ORDER (include/bnpc_hip.h): the sums now ...
extern "C" int bnpc_ll_theta_pinned_sums_issue(bnpc_ctx *c, int view,
                                               const float *theta, int64_t K,
                                               double FP, double FN,
                                               int64_t ldo,
                                               const double *col_prior,
                                               double **host)
{
    return ll_top2_impl(c, view, theta, K, FP, FN, ldo, col_prior, host,
                        nullptr, false, true);
}

// ... the hints when the caller has its visiting order: record r of *top2 is
// made from row order[r] of the matrix.  The order travels through a pinned
// buffer of its own and a copy kernel (every wave of the wide hint kernel
// reads one entry: the host link should see the list once, in whole lines).
// *top2 = NULL: no hints (no zero-copy memory) - the matrix is complete on
// the host instead.
extern "C" int bnpc_hints_in_order_issue(bnpc_ctx *c, const int64_t *order,
                                         bnpc_top2 **top2)
{
    ARGCHK(c && order && top2, "NULL argument");
    *top2 = nullptr;
    HIPCHK(hipSetDevice(c->device));
    const int state = c->hint_later.state;
    c->hint_later.state = 0;
    // (another evaluation since - it reuses the matrix's device buffer - or
    // none at all: the rows the records would be made from are not there)
    ARGCHK(state != 0, "no sums issued for these hints "
                       "(bnpc_ll_theta_pinned_sums_issue comes first)");
    // (sums issued without a hint buffer: the matrix was copied there)
    if (state == 2) return 0;
    const int64_t n = c->hint_later.n;
    ARGCHK(n <= c->N, "more rows than cells");
    {
        // every entry a row of the matrix (one pass the compiler vectorises)
        uint64_t out_of_range = 0;
        for (int64_t i = 0; i < n; i++)
            out_of_range |= (uint64_t)((uint64_t)order[i] >= (uint64_t)n);
        ARGCHK(!out_of_range, "order entry out of range");
    }
    const size_t ob = (size_t)n * sizeof(long long);
    const size_t ob2 = (ob + 15) & ~(size_t)15;
    // (the caller has drawn its permutation from the stream by now: a host
    // that gives no mapped pinned memory must not fail the sweep here - the
    // order then travels by a plain copy from the caller's array)
    if (!c->order_pin
        && hipHostMalloc(&c->order_pin, ((size_t)c->N + 2) * 8,
                         hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->order_pin = nullptr;
    }
    void *op_dev = nullptr;
    if (c->order_pin
        && hipHostGetDevicePointer(&op_dev, c->order_pin, 0) != hipSuccess) {
        (void)hipGetLastError();
        op_dev = nullptr;
    }
    if (ensure(c->order_dev, ob2)) return 1;
    if (op_dev) {
        memcpy(c->order_pin, order, ob);
        const long long n2 = (long long)(ob2 / 16);
        BNPC_LAUNCH(k_stage_copy, dim3((unsigned)((n2 + 255) / 256)),
                    dim3(256), 0, c->stream, (const double2 *)op_dev,
                    (double2 *)c->order_dev.p, n2);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemcpyAsync(c->order_dev.p, order, ob,
                              hipMemcpyHostToDevice, c->stream));
    }
    if (int rh = hint_launch(c, (const double *)c->out.p, n, c->hint_later.K,
                             c->hint_later.ldo, c->hint_later.prior,
                             c->hint_later.hint_dev, c->hint_later.rows_dev,
                             (const long long *)c->order_dev.p))
        return rh;
    if (int rq = hints_queued(c, c->hint_later.K, c->hint_later.bytes, false))
        return rq;
    *top2 = (bnpc_top2 *)c->hint_pin;
    return 0;
}

// The hints of the last bnpc_ll_theta_pinned_top2_issue are complete on
// return (the matrix copy that may be queued behind them is not waited for).
extern "C" int bnpc_hints_wait(bnpc_ctx *c)
{
    ARGCHK(c, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (c->ev_hints && c->pin_lazy_bytes)
        HIPCHK(hipEventSynchronize(c->ev_hints));
    else
        HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// Resident parameter rows: row r of the store = parameters of cluster id r.
extern "C" int bnpc_theta_put(bnpc_ctx *c, int64_t row0, const float *theta,
                              int64_t R)
{
    ARGCHK(c && theta, "NULL argument");
    ARGCHK(row0 >= 0 && R > 0, "bad row range");
    HIPCHK(hipSetDevice(c->device));
    const size_t row_bytes = (size_t)c->M * sizeof(float);
    const size_t need = (size_t)(row0 + R) * row_bytes;
    if (need > c->theta_store.cap) {
        // grow, keeping the rows already stored
        DevBuf bigger;
        if (ensure(bigger, need + need / 2)) return 1;
        if (c->theta_store.p) {
            HIPCHK(hipMemcpyAsync(bigger.p, c->theta_store.p,
                                  (size_t)c->store_rows * row_bytes,
                                  hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipFree(c->theta_store.p));
        }
        c->theta_store = bigger;
    }
    SideLane lane(c);
    HIPCHK(hipMemcpyAsync((char *)c->theta_store.p + (size_t)row0 * row_bytes,
                          theta, (size_t)R * row_bytes, hipMemcpyHostToDevice,
                          c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));     // theta is only borrowed
    if (row0 + R > c->store_rows) c->store_rows = row0 + R;
    return 0;
}

// bnpc_ll_theta_pinned with the K parameter vectors taken from the resident
// store: cluster k uses store row rows[k] (no host gather, no re-upload).
extern "C" int bnpc_ll_rows_pinned(bnpc_ctx *c, int view, const int64_t *rows,
                                   int64_t K, double FP, double FN,
                                   int64_t ldo, double **host)
{
    ARGCHK(c && host && rows, "NULL argument");
    ARGCHK(view >= 0 && view < BNPC_MAX_VIEWS, "view out of range");
    ARGCHK(K > 0, "K must be positive");
    ARGCHK(FP > 0.0 && FP < 1.0 && FN > 0.0 && FN < 1.0,
           "error rates must lie in (0, 1)");
    for (int64_t k = 0; k < K; k++)
        ARGCHK(rows[k] >= 0 && rows[k] < c->store_rows,
               "row is not in the resident parameter store");
    if (ldo == 0) ldo = K;
    ARGCHK(ldo >= K, "ldo smaller than K");
    *host = nullptr;
    HIPCHK(hipSetDevice(c->device));
    if (ensure(c->row_idx, K * sizeof(long long))) return 1;
    HIPCHK(hipMemcpyAsync(c->row_idx.p, rows, K * sizeof(long long),
                          hipMemcpyHostToDevice, c->stream));
    c->use_rows = (const long long *)c->row_idx.p;
    int rc = ll_common(c, view, K, ldo, true, FP, FN, nullptr);
    c->use_rows = nullptr;
    if (rc) return rc;
    const size_t bytes = (size_t)c->views[view].n * ldo * sizeof(double);
    if (bytes == 0) return 0;
    if (ensure_pin(c, bytes)) return 1;
    HIPCHK(hipMemcpyAsync(c->pin, c->out.p, bytes, hipMemcpyDeviceToHost,
                          c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *host = (double *)c->pin;
    return 0;
}

// bnpc_ll_rows_pinned without the final wait.  Tables and sums are enqueued
// on the context's stream and write the device buffer of this issue's parity;
// the copy into the slot's own pinned buffer follows on the copy stream, so
// that it runs under the sums of the next tile; an event marks its completion.
static int ll_rows_issue_impl(bnpc_ctx *c, int view, const int64_t *rows,
                              int64_t K, double FP, double FN, int64_t ldo,
                              int slot, const double *col_prior)
{
    ARGCHK(c && rows, "NULL argument");
    ARGCHK(slot >= 0 && slot < BNPC_TILE_SLOTS, "slot out of range");
    ARGCHK(view >= 0 && view < BNPC_MAX_VIEWS, "view out of range");
    ARGCHK(K > 0, "K must be positive");
    ARGCHK(FP > 0.0 && FP < 1.0 && FN > 0.0 && FN < 1.0,
           "error rates must lie in (0, 1)");
    ARGCHK(!c->tile_pending[slot], "slot has an unconsumed tile");
    for (int64_t k = 0; k < K; k++)
        ARGCHK(rows[k] >= 0 && rows[k] < c->store_rows,
               "row is not in the resident parameter store");
    if (ldo == 0) ldo = K;
    ARGCHK(ldo >= K, "ldo smaller than K");
    HIPCHK(hipSetDevice(c->device));
    const size_t bytes = (size_t)c->views[view].n * ldo * sizeof(double);
    ARGCHK(bytes > 0, "empty view");
    const int par = (int)(c->tile_seq & 1);
    if (!c->tile_done[slot])
        HIPCHK(hipEventCreateWithFlags(&c->tile_done[slot],
                                       hipEventDisableTiming));
    if (!c->tile_summed[par]) {
        HIPCHK(hipEventCreateWithFlags(&c->tile_summed[par],
                                       hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->tile_out_free[par],
                                       hipEventDisableTiming));
        // nothing has read this buffer yet
        HIPCHK(hipEventRecord(c->tile_out_free[par], c->stream));
    }
    if (ensure_lanes(c)) return 1;
    if (!c->copy_stream)
        HIPCHK(hipStreamCreateWithFlags(&c->copy_stream,
                                        hipStreamNonBlocking));
    if (ensure_host(&c->tile_rows[slot], &c->tile_rows_cap[slot],
                    K * sizeof(long long)))
        return 1;
    memcpy(c->tile_rows[slot], rows, K * sizeof(long long));
    if (ensure(c->row_idx, K * sizeof(long long))) return 1;
    // the copy of the tile that used this device buffer two issues ago must
    // have left it (growing the buffer synchronises the device anyway)
    HIPCHK(hipStreamWaitEvent(c->stream, c->tile_out_free[par], 0));
    if (bytes > c->tile_out[par].cap)
        HIPCHK(hipStreamSynchronize(c->copy_stream));
    if (ensure(c->tile_out[par], bytes)) return 1;
    HIPCHK(hipMemcpyAsync(c->row_idx.p, c->tile_rows[slot],
                          K * sizeof(long long), hipMemcpyHostToDevice,
                          c->stream));
    c->use_rows = (const long long *)c->row_idx.p;
    c->dst_override = (double *)c->tile_out[par].p;
    int rc = ll_common(c, view, K, ldo, true, FP, FN, nullptr);
    c->dst_override = nullptr;
    c->use_rows = nullptr;
    if (rc) return rc;
    c->tile_hinted[slot] = false;
    if (col_prior) {
        // the tile's hints: one record per row, written in place into pinned
        // memory by a kernel behind the sums (the priors at issue time travel
        // through a pinned staging copy)
        const size_t rows_n = (size_t)c->views[view].n;
        if (ensure_host(&c->tile_hint[slot], &c->tile_hint_cap[slot],
                        std::max<size_t>(rows_n, (size_t)c->N)
                            * sizeof(bnpc_top2)))
            return 1;
        if (ensure_host(&c->tile_prior[slot], &c->tile_prior_cap[slot],
                        K * sizeof(double)))
            return 1;
        memcpy(c->tile_prior[slot], col_prior, K * sizeof(double));
        if (ensure(c->tile_prior_dev[par], K * sizeof(double))) return 1;
        HIPCHK(hipMemcpyAsync(c->tile_prior_dev[par].p, c->tile_prior[slot],
                              K * sizeof(double), hipMemcpyHostToDevice,
                              c->stream));
        void *hint_dev = nullptr;
        HIPCHK(hipHostGetDevicePointer(&hint_dev, c->tile_hint[slot], 0));
        BNPC_LAUNCH(k_row_top2_wide, dim3((unsigned)rows_n), dim3(256),
                           0, c->stream, (const double *)c->tile_out[par].p,
                           (long long)ldo, (int)K,
                           (const double *)c->tile_prior_dev[par].p,
                           (bnpc_top2 *)hint_dev);
        HIPCHK(hipGetLastError());
        c->tile_hinted[slot] = true;
    }
    HIPCHK(hipEventRecord(c->tile_summed[par], c->stream));
    // the slot's pinned result buffer is made (18 ms per 256 MiB the first
    // time) while the device is busy with the tile just queued - not in front
    // of it: three of them were 54 ms of a config-5 first sweep
    if (ensure_host(&c->tile_pin[slot], &c->tile_cap[slot], bytes)) return 1;
    HIPCHK(hipStreamWaitEvent(c->copy_stream, c->tile_summed[par], 0));
    HIPCHK(hipMemcpyAsync(c->tile_pin[slot], c->tile_out[par].p, bytes,
                          hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(hipEventRecord(c->tile_done[slot], c->copy_stream));
    HIPCHK(hipEventRecord(c->tile_out_free[par], c->copy_stream));
    c->tile_bytes[slot] = bytes;
    c->tile_pending[slot] = true;
    c->tile_seq++;
    return 0;
}

extern "C" int bnpc_ll_rows_issue(bnpc_ctx *c, int view, const int64_t *rows,
                                  int64_t K, double FP, double FN, int64_t ldo,
                                  int slot)
{
    return ll_rows_issue_impl(c, view, rows, K, FP, FN, ldo, slot, nullptr);
}

extern "C" int bnpc_ll_rows_issue_hint(bnpc_ctx *c, int view,
                                       const int64_t *rows, int64_t K,
                                       double FP, double FN, int64_t ldo,
                                       int slot, const double *col_prior)
{
    ARGCHK(col_prior, "col_prior is NULL");
    return ll_rows_issue_impl(c, view, rows, K, FP, FN, ldo, slot, col_prior);
}

extern "C" int bnpc_ll_rows_wait_hint(bnpc_ctx *c, int slot, double **host,
                                      bnpc_top2 **hint)
{
    ARGCHK(hint, "NULL argument");
    *hint = nullptr;
    const bool hinted = c && slot >= 0 && slot < BNPC_TILE_SLOTS
        && c->tile_hinted[slot];
    if (int rc = bnpc_ll_rows_wait(c, slot, host)) return rc;
    if (hinted) *hint = (bnpc_top2 *)c->tile_hint[slot];
    return 0;
}

extern "C" int bnpc_ll_rows_wait(bnpc_ctx *c, int slot, double **host)
{
    ARGCHK(c && host, "NULL argument");
    ARGCHK(slot >= 0 && slot < BNPC_TILE_SLOTS, "slot out of range");
    *host = nullptr;
    ARGCHK(c->tile_pending[slot], "no tile was issued on this slot");
    HIPCHK(hipSetDevice(c->device));
    c->tile_pending[slot] = false;
    HIPCHK(hipEventSynchronize(c->tile_done[slot]));
    *host = (double *)c->tile_pin[slot];
    return 0;
}

extern "C" int bnpc_ll_tables(bnpc_ctx *c, int view, const double *L1,
                              const double *L0, int64_t K, double *out,
                              int64_t ldo)
{
    ARGCHK(c, "ctx is NULL");
    ARGCHK(view >= 0 && view < BNPC_MAX_VIEWS, "view out of range");
    ARGCHK(K >= 0 && (K == 0 || (L1 && L0)), "table is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (K == 0) return 0;
    const size_t bytes = (size_t)K * c->M * sizeof(double);
    if (arena_reset(c)) return 1;
    c->tab_src = nullptr;
    if (c->tun.zero_copy && (int64_t)(2 * bytes) <= ZC_IN_MAX) {
        // L1 then L0, contiguous in the arena
        void *slot = stage_slot(c, 2 * bytes);
        if (slot && c->stage_dev) {
            memcpy(slot, L1, bytes);
            memcpy((char *)slot + bytes, L0, bytes);
            c->tab_src = (const double *)(c->stage_dev
                + ((char *)slot - (char *)c->stage));
        }
    }
    if (!c->tab_src) {
        if (ensure(c->tab_in, 2 * bytes)) return 1;
        if (h2d(c, c->tab_in.p, L1, bytes)) return 1;
        if (h2d(c, (char *)c->tab_in.p + bytes, L0, bytes)) return 1;
        c->tab_src = (const double *)c->tab_in.p;
    }
    int rc = ll_common(c, view, K, ldo, false, 0.0, 0.0, out);
    if (!out && rc == 0) HIPCHK(hipStreamSynchronize(c->stream));
    return rc;
}

// ---- column counts ---------------------------------------------------------
static int colcounts_device(bnpc_ctx *c, const int64_t *cells, int64_t n_cells,
                            const int64_t *seg_offsets, int64_t G,
                            DevBuf &cnt)
{
    if (arena_reset(c)) return 1;
    const size_t cnt_bytes = (size_t)2 * G * c->M * sizeof(int32_t);
    if (ensure(cnt, cnt_bytes)) return 1;
    HIPCHK(hipMemsetAsync(cnt.p, 0, cnt_bytes, c->stream));
    std::vector<Chunk> chunks;
    const int64_t CH = 32;
    for (int64_t g = 0; g < G; g++)
        for (int64_t b = seg_offsets[g]; b < seg_offsets[g + 1]; b += CH)
            chunks.push_back(
                {(long long)b,
                 (long long)std::min<int64_t>(seg_offsets[g + 1], b + CH),
                 (long long)g});
    if (chunks.empty()) return 0;
    if (ensure(c->cells, n_cells * sizeof(long long))) return 1;
    if (ensure(c->chunks, chunks.size() * sizeof(Chunk))) return 1;
    if (h2d(c, c->cells.p, cells, n_cells * sizeof(long long))) return 1;
    if (h2d(c, c->chunks.p, chunks.data(), chunks.size() * sizeof(Chunk)))
        return 1;
    int *n1 = (int *)cnt.p;
    int *n0 = n1 + (size_t)G * c->M;
    // the chunk list lives in a std::vector: finish the copy before it dies
    const size_t maxy = 65535;
    for (size_t off = 0; off < chunks.size(); off += maxy) {
        const size_t ny = std::min(maxy, chunks.size() - off);
        dim3 grid((unsigned)((c->M + 255) / 256), (unsigned)ny);
        BNPC_LAUNCH(k_colcounts, grid, dim3(256), 0, c->stream,
                           c->rows, c->W, (int)c->M,
                           (const long long *)c->cells.p,
                           (const Chunk *)c->chunks.p + off, n1, n0);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// Column counts of G segments of a view from its lane masks (K3b).
// label_of(s) = segment of slot s, or < 0 for none.  The counts land in `cnt`
// on the device ([n1: G x M][n0: G x M]) and in n1 / n0 on the host.
// `defer` (optional): if the results can be written in place into pinned
// memory, return right after the launch - *defer receives where they will
// appear, and the caller synchronises the stream and copies them out (it has
// more work to put on the stream first); else *defer stays NULL and the
// call completes here as usual.
template <typename LabelOf>
static int counts_from_masks(bnpc_ctx *c, int view, LabelOf label_of,
                             int64_t G, DevBuf &cnt, int32_t *n1, int32_t *n0,
                             const int **defer = nullptr)
{
    if (defer) *defer = nullptr;
    const View &v = c->views[view];
    const size_t half = (size_t)G * c->M * sizeof(int32_t);
    if (ensure(cnt, 2 * half)) return 1;
    if (arena_reset(c)) return 1;
    if (v.n == 0) {
        HIPCHK(hipMemsetAsync(cnt.p, 0, 2 * half, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (n1) memset(n1, 0, half);
        if (n0) memset(n0, 0, half);
        return 0;
    }
    // membership words, built in the arena when they fit
    const size_t mem_bytes = (size_t)G * v.nblk * sizeof(unsigned long long);
    unsigned long long *mem = nullptr;
    const unsigned long long *d_mem = nullptr;
    std::vector<unsigned long long> heap;
    if (c->tun.zero_copy && (int64_t)mem_bytes <= ZC_IN_MAX) {
        mem = (unsigned long long *)stage_slot(c, mem_bytes);
        if (mem && c->stage_dev)
            d_mem = (const unsigned long long *)(c->stage_dev
                + ((char *)mem - (char *)c->stage));
        else
            mem = nullptr;
    }
    if (!mem) {
        heap.resize((size_t)G * v.nblk);
        mem = heap.data();
    }
    memset(mem, 0, mem_bytes);
    for (int64_t s = 0; s < v.n; s++) {
        const int64_t g = label_of(s);
        if (g >= G) {
            bnpc_set_error("bad argument: segment label out of range");
            return 2;
        }
        if (g >= 0) mem[(size_t)g * v.nblk + (s >> 6)] |= 1ull << (s & 63);
    }
    if (!d_mem) {
        if (ensure(c->chunks, mem_bytes)) return 1;
        HIPCHK(hipMemcpyAsync(c->chunks.p, mem, mem_bytes,
                              hipMemcpyHostToDevice, c->stream));
        d_mem = (const unsigned long long *)c->chunks.p;
    }
    void *zc_dev = nullptr;
    int *zc_host = (n1 && n0) ? (int *)zc_result(c, 2 * half, &zc_dev)
                              : nullptr;
    int *d1 = (int *)cnt.p, *d0 = d1 + (size_t)G * c->M;
    int *h1 = zc_host ? (int *)zc_dev : nullptr;
    int *h0 = h1 ? h1 + (size_t)G * c->M : nullptr;
    dim3 grid((unsigned)(c->Mpad / 64), (unsigned)((G + CM_SEG - 1) / CM_SEG));
    // dynamic LDS: nblk x CM_SEG membership words (50 KB at 50000 cells)
    size_t lds = (size_t)v.nblk * CM_SEG * sizeof(unsigned long long);
    const size_t red_bytes = 4 * 2 * CM_SEG * 64 * sizeof(int);
    if (lds < red_bytes) lds = red_bytes;
    if (lds > 56 * 1024) {
        bnpc_set_error("view too large for the mask-count kernel");
        return 1;
    }
    unsigned done_seq = 0;
    const DoneSignal sig = (zc_host && !defer) ? make_signal(c, 0, &done_seq)
                                               : DoneSignal{nullptr, nullptr, 0};
    BNPC_LAUNCH(k_counts_masks, grid, dim3(256), lds, c->stream,
                       (const ulonglong2 *)v.masks.p, c->Mpad, (int)c->M,
                       (long long)v.nblk, d_mem, (int)G, d1, d0, h1, h0, sig);
    HIPCHK(hipGetLastError());
    if (zc_host && defer) {
        *defer = zc_host;
    } else if (zc_host) {
        if (int rc = wait_done(c, 0, done_seq)) return rc;
        memcpy(n1, zc_host, half);
        memcpy(n0, zc_host + (size_t)G * c->M, half);
    } else if (n1 && n0) {
        // n1 and n0 are contiguous on the device: one copy when the arena
        // can take it, else two into the caller's arrays
        void *slot = stage_slot(c, 2 * half);
        if (slot) {
            HIPCHK(hipMemcpyAsync(slot, cnt.p, 2 * half,
                                  hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            memcpy(n1, slot, half);
            memcpy(n0, (char *)slot + half, half);
        } else {
            HIPCHK(hipMemcpyAsync(n1, d1, half, hipMemcpyDeviceToHost,
                                  c->stream));
            HIPCHK(hipMemcpyAsync(n0, d0, half, hipMemcpyDeviceToHost,
                                  c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
    } else {
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

extern "C" int bnpc_view_counts(bnpc_ctx *c, int view, const int64_t *labels,
                                int64_t G, int32_t *n1, int32_t *n0)
{
    ARGCHK(c && n1 && n0, "NULL argument");
    ARGCHK(view >= 0 && view < BNPC_MAX_VIEWS, "view out of range");
    ARGCHK(G > 0 && G <= 4096, "G out of range");
    ARGCHK(c->views[view].n == 0 || labels, "labels is NULL");
    HIPCHK(hipSetDevice(c->device));
    SideLane lane(c);
    c->cnt_rows = 0;
    int rc = counts_from_masks(c, view,
        [=](int64_t s) -> int64_t { return labels[s]; }, G, c->cnt, n1, n0);
    if (rc == 0) c->cnt_rows = G;
    return rc;
}

extern "C" int bnpc_colcounts(bnpc_ctx *c, const int64_t *cells,
                              const int64_t *seg_offsets, int64_t G,
                              int32_t *n1, int32_t *n0)
{
    ARGCHK(c && seg_offsets && n1 && n0, "NULL argument");
    ARGCHK(G >= 0, "G negative");
    if (G == 0) return 0;
    ARGCHK(seg_offsets[0] == 0, "seg_offsets[0] must be 0");
    const int64_t n = seg_offsets[G];
    for (int64_t g = 0; g < G; g++)
        ARGCHK(seg_offsets[g] <= seg_offsets[g + 1], "seg_offsets not sorted");
    ARGCHK(n == 0 || cells, "cells is NULL");
    for (int64_t i = 0; i < n; i++)
        ARGCHK(cells[i] >= 0 && cells[i] < c->N, "cell index out of range");
    HIPCHK(hipSetDevice(c->device));
    c->cnt_rows = 0;
    if (colcounts_device(c, cells, n, seg_offsets, G, c->cnt)) return 1;
    const size_t half = (size_t)G * c->M * sizeof(int32_t);
    D2H a, b;
    if (d2h_begin(c, a, n1, c->cnt.p, half)) return 1;
    if (d2h_begin(c, b, n0, (char *)c->cnt.p + half, half)) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    d2h_finish(a);
    d2h_finish(b);
    return 0;
}

static int colcounts_by_label_impl(bnpc_ctx *c, const int64_t *assignment,
                                   const int64_t *ids, int64_t K,
                                   int32_t *n1, int32_t *n0,
                                   const int **defer)
{
    if (defer) *defer = nullptr;
    ARGCHK(c && assignment && ids, "NULL argument");
    c->lab_gen++;
    ARGCHK(K > 0, "K must be positive");
    HIPCHK(hipSetDevice(c->device));
    // counting sort of the cells by the position of their label in ids[]
    int64_t max_id = 0;
    for (int64_t g = 0; g < K; g++) {
        ARGCHK(ids[g] >= 0, "negative cluster id");
        max_id = std::max(max_id, ids[g]);
    }
    std::vector<int64_t> pos(max_id + 1, -1);
    for (int64_t g = 0; g < K; g++) {
        ARGCHK(pos[ids[g]] < 0, "duplicate cluster id");
        pos[ids[g]] = g;
    }
    std::vector<int64_t> offs(K + 1, 0);
    for (int64_t i = 0; i < c->N; i++) {
        const int64_t a = assignment[i];
        ARGCHK(a >= 0 && a <= max_id && pos[a] >= 0,
               "assignment holds an id that is not in ids");
        offs[pos[a] + 1]++;
    }
    if (K <= c->tun.mask_counts_max
        && (size_t)c->views[0].nblk * CM_SEG * 8 <= 56 * 1024) {
        // few clusters: popcounts over the lane masks of the identity view
        const int64_t *pp = pos.data();
        int rc = counts_from_masks(c, 0,
            [=](int64_t s) -> int64_t { return pp[assignment[s]]; }, K,
            c->lab_cnt, n1, n0, defer);
        if (rc == 0) c->lab_K = K;
        return rc;
    }
    for (int64_t g = 0; g < K; g++) offs[g + 1] += offs[g];
    std::vector<int64_t> cells(c->N), fill(offs.begin(), offs.end() - 1);
    for (int64_t i = 0; i < c->N; i++)
        cells[fill[pos[assignment[i]]]++] = i;
    if (colcounts_device(c, cells.data(), c->N, offs.data(), K, c->lab_cnt))
        return 1;
    c->lab_K = K;
    if (n1 && n0) {
        const size_t half = (size_t)K * c->M * sizeof(int32_t);
        D2H a, b;
        if (d2h_begin(c, a, n1, c->lab_cnt.p, half)) return 1;
        if (d2h_begin(c, b, n0, (char *)c->lab_cnt.p + half, half)) return 1;
        HIPCHK(hipStreamSynchronize(c->stream));
        d2h_finish(a);
        d2h_finish(b);
    }
    return 0;
}

uint64_t bnpc_ctx_label_counts_generation(const bnpc_ctx *c)
{
    return c ? c->lab_gen : 0;
}

extern "C" int bnpc_colcounts_by_label(bnpc_ctx *c, const int64_t *assignment,
                                       const int64_t *ids, int64_t K,
                                       int32_t *n1, int32_t *n0)
{
    return colcounts_by_label_impl(c, assignment, ids, K, n1, n0, nullptr);
}


// ---- device screen of a parameter batch -----------------------------------
// layout of the pinned block for G x M = E elements (all 16-byte aligned):
//   U[E] f64 | u[E] f64 | sd_idx[E] i32 | theta[E] f32 | new32[E] f32 |
//   flags[E] u8
// (new32: the proposals whose float32 bits the screen vouches for, flag 3)
struct MHPin {
    double *U, *u;
    int32_t *sd_idx;
    float *theta;
    float *new32;
    uint8_t *flags;
};

static size_t mh_pin_offsets(size_t E, size_t off[6])
{
    const size_t Ea = (E + 15) & ~(size_t)15;
    off[0] = 0;
    off[1] = off[0] + Ea * 8;
    off[2] = off[1] + Ea * 8;
    off[3] = off[2] + Ea * 4;
    off[4] = off[3] + Ea * 4;
    off[5] = off[4] + Ea * 4;
    return off[5] + Ea;
}

// ---- draws taken AHEAD ------------------------------
// The draws of a parameter batch - choice(sd, M), M uniforms, M uniforms per
// cluster, cluster by cluster (libs/CRP.py:328-335) - depend on nothing but
// the position of the stream, and rank 0's walk through them paces a large
// batch (0.5 ms of config 5's 0.9 ms parameter phase) while the same thread
// idles 0.9 ms in front of the sweep's kernel.  So the step posts a WALKER on
// the aside thread as soon as the stream's way to the batch is known (after a
// sweep's permutation: one uniform per cell, the alpha test; in a move: after
// its last variable draw): on a private COPY of the stream it goes that way
// (`prelude`), notes the state it arrives with (`start`) and draws rows
// straight into the pinned block, publishing row after row.  The batch adopts
// them iff the live stream stands exactly at `start` when it begins - any
// birth, any other draw, any difference makes the two states differ, and the
// walker's work is dropped (the live stream was never touched).  Same bits
// by construction: the same generator from the same state.  A batch with
// more rows than the walker took draws the rest itself, one with fewer
// continues from the state kept after its last row.
struct MhAhead {
    // set by the thread that posts the walker
    bnpc_mt19937 rng;               // the walker's private stream
    bnpc_legacy_gauss gauss;
    std::function<bool(bnpc_mt19937 *, bnpc_legacy_gauss *)> prelude;
    int64_t rows = 0, M = 0, n_sd = 0;
    MHPin h;
    bool active = false;            // posted and not yet taken or dropped
    // written by the walker
    std::atomic<int> start_known{0};    // 1: `start` holds; -1: no way known
    bnpc_mt19937 start;
    std::atomic<int64_t> rows_ready{0};
    std::vector<bnpc_mt19937> after_row;
    std::atomic<int> cancel{0};
};

static inline void mt_canonical(bnpc_mt19937 &s)
{
    if (s.pos >= 624) mt_refill(&s);
}

static bool mt_same_position(const bnpc_mt19937 &a, const bnpc_mt19937 &b)
{
    bnpc_mt19937 x = a, y = b;
    mt_canonical(x);
    mt_canonical(y);
    return x.pos == y.pos && memcmp(x.key, y.key, sizeof x.key) == 0;
}

// the walker is stopped and forgotten (its draws were not wanted)
static void mh_ahead_drop(bnpc_ctx *c)
{
    MhAhead *ah = c->ahead;
    if (!ah || !ah->active) return;
    ah->cancel.store(1, std::memory_order_release);
    bnpc_aside_wait();
    ah->active = false;
}

// A small batch (the rows of a restricted scan) takes a walker's rows whole:
// the number adopted (the live stream then stands behind them), 0 if there is
// no walker or the live stream stands elsewhere.
static int64_t mh_ahead_adopt(bnpc_ctx *c, bnpc_mt19937 *rng, int64_t G,
                              int64_t M, int64_t n_sd)
{
    MhAhead *ah = c->ahead;
    if (!ah || !ah->active) return 0;
    bool ok = rng && ah->M == M && ah->n_sd == n_sd && c->tun.mh_ahead != 3;
    if (ok) {
        int known;
        while ((known = ah->start_known.load(std::memory_order_acquire)) == 0)
            __builtin_ia32_pause();
        ok = known == 1 && mt_same_position(ah->start, *rng);
    }
    if (!ok) {
        mh_ahead_drop(c);
        return 0;
    }
    const int64_t rows = std::min<int64_t>(ah->rows, G);
    while (ah->rows_ready.load(std::memory_order_acquire) < rows)
        __builtin_ia32_pause();
    *rng = ah->after_row[(size_t)(rows - 1)];
    mh_ahead_drop(c);
    c->ahead_taken++;
    c->ahead_rows_taken += rows;
    return rows;
}

static void mh_ahead_destroy(bnpc_ctx *c)
{
    mh_ahead_drop(c);
    delete c->ahead;
    c->ahead = nullptr;
}

static int mh_pin_get(bnpc_ctx *c, size_t E, MHPin &host, MHPin &dev,
                      bool keep_ahead = false)
{
    // (whoever else wants the block ends a walker that writes into it)
    if (!keep_ahead) mh_ahead_drop(c);
    size_t off[6];
    // the block is laid out for its CAPACITY in entries, not for this batch:
    // rows the walker drew for K + 1 clusters lie where a batch of K finds them
    size_t need = mh_pin_offsets(E, off);
    if (E <= c->mh_capE) need = mh_pin_offsets(c->mh_capE, off);
    if (need > c->mh_cap || E > c->mh_capE) {
        mh_ahead_drop(c);
        // the screen of a batch in flight reads this block: nothing is in
        // flight here (every screened call ends with a synchronisation)
        if (c->mh_pin) HIPCHK(hipHostFree(c->mh_pin));
        c->mh_pin = nullptr;
        c->mh_dev = nullptr;
        c->mh_cap = 0;
        // (no head room for the largest blocks: a slice of a batch beyond the
        // budget never grows)
        const size_t capE = E + (need < c->tun.mh_pin_max ? E / 4 : 0) + 64;
        const size_t cap = mh_pin_offsets(capE, off) + 4096;
        c->mh_capE = 0;
        const hipError_t e = hipHostMalloc(&c->mh_pin, cap,
                                           hipHostMallocDefault);
        if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
            // the caller evaluates the batch without the screen
            (void)hipGetLastError();
            c->mh_pin = nullptr;
            bnpc_set_error("no pinned memory for the screen of a parameter "
                           "batch (%zu bytes)", cap);
            return MH_PIN_NO_MEMORY;
        }
        HIPCHK(e);
        void *d = nullptr;
        HIPCHK(hipHostGetDevicePointer(&d, c->mh_pin, 0));
        c->mh_dev = (char *)d;
        c->mh_cap = cap;
        c->mh_capE = capE;
    }
    char *h = (char *)c->mh_pin, *d = c->mh_dev;
    host = {(double *)(h + off[0]), (double *)(h + off[1]),
            (int32_t *)(h + off[2]), (float *)(h + off[3]),
            (float *)(h + off[4]), (uint8_t *)(h + off[5])};
    dev = {(double *)(d + off[0]), (double *)(d + off[1]),
           (int32_t *)(d + off[2]), (float *)(d + off[3]),
           (float *)(d + off[4]), (uint8_t *)(d + off[5])};
    return 0;
}

// Post the walker (bnpc_internal.h).  Returns 0 whether or not one was posted
// (*posted says): no walker is never an error.
int bnpc_mh_ahead_begin(bnpc_ctx *c, const bnpc_mt19937 *rng,
                        const bnpc_legacy_gauss *g,
                        const std::function<bool(bnpc_mt19937 *,
                                                 bnpc_legacy_gauss *)> &prelude,
                        int64_t rows, int64_t M, int64_t n_sd, bool *posted,
                        int64_t min_entries)
{
    if (posted) *posted = false;
    if (!c || !rng || !g || rows < 1 || M != c->M || n_sd < 1 || n_sd > 8)
        return 0;
    const int mode = c->tun.mh_ahead;
    const int64_t E = rows * M;
    size_t off[6];
    if (min_entries < 0) min_entries = MH_AHEAD_MIN;
    if (mode == 0 || !c->tun.mh_screen || rows > MH_AHEAD_MAX_ROWS
        || E < (mode >= 2 ? MH_SCREEN_MIN : min_entries)
        || mh_pin_offsets((size_t)E, off) > c->tun.mh_pin_max
        || c->any_tile_pending())
        return 0;
    if (hipSetDevice(c->device) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (!c->ahead) c->ahead = new MhAhead();
    MHPin h, d;
    if (mh_pin_get(c, (size_t)E, h, d)) return 0;   // (drops a walker at work)
    MhAhead *ah = c->ahead;
    ah->rng = *rng;
    ah->gauss = *g;
    ah->prelude = prelude;
    ah->rows = rows;
    ah->M = M;
    ah->n_sd = n_sd;
    ah->h = h;
    ah->start_known.store(0, std::memory_order_relaxed);
    ah->rows_ready.store(0, std::memory_order_relaxed);
    ah->cancel.store(0, std::memory_order_relaxed);
    ah->after_row.resize((size_t)rows);
    const bool ok = bnpc_aside_start([ah]() {
        if (ah->prelude && !ah->prelude(&ah->rng, &ah->gauss)) {
            ah->start_known.store(-1, std::memory_order_release);
            return;
        }
        mt_canonical(ah->rng);
        ah->start = ah->rng;
        ah->start_known.store(1, std::memory_order_release);
        const int64_t M = ah->M;
        for (int64_t r = 0; r < ah->rows; r++) {
            if (ah->cancel.load(std::memory_order_acquire)) break;
            // (non-temporal stores, fenced before the row is published)
            if (bnpc_mt_mh_draws_to(&ah->rng, 1, M, ah->n_sd,
                                    ah->h.sd_idx + r * M, ah->h.U + r * M,
                                    ah->h.u + r * M, true))
                break;
            ah->after_row[(size_t)r] = ah->rng;
            ah->rows_ready.store(r + 1, std::memory_order_release);
        }
    });
    if (!ok) return 0;
    ah->active = true;
    c->ahead_begun++;
    if (posted) *posted = true;
    return 0;
}

void bnpc_mh_ahead_drop(bnpc_ctx *c)
{
    if (c) mh_ahead_drop(c);
}

// The rows of a restricted scan's parameter batch (bnpc_rg_counts_and_batch)
// taken ahead: posted by the scan when its visiting order is drawn - exactly
// `uniforms` uniforms (one per cell) lie between there and the batch.
void bnpc_mh_ahead_scan(bnpc_ctx *c, const bnpc_mt19937 *rng, int64_t uniforms,
                        int64_t rows, int64_t M, int64_t n_sd)
{
    static const bnpc_legacy_gauss no_gauss = {};
    const auto way = [uniforms](bnpc_mt19937 *r, bnpc_legacy_gauss *) {
        for (int64_t left = 2 * uniforms; left > 0;) {
            if (r->pos >= 624) mt_refill(r);
            const int64_t take = std::min<int64_t>(624 - r->pos, left);
            r->pos += (int32_t)take;
            left -= take;
        }
        return true;
    };
    bool posted = false;
    (void)bnpc_mh_ahead_begin(c, rng, &no_gauss, way, rows, M, n_sd, &posted,
                              MH_AHEAD_SCAN_MIN);
}

extern "C" int bnpc_mh_ahead_stats(bnpc_ctx *c, int64_t *begun, int64_t *taken,
                                   int64_t *rows)
{
    ARGCHK(c && begun && taken && rows, "NULL argument");
    *begun = c->ahead_begun;
    *taken = c->ahead_taken;
    *rows = c->ahead_rows_taken;
    return 0;
}

// counts of the batch's rows on the device: src 0 = the per-cluster counts of
// the last bnpc_colcounts_by_label (G rows), src 1 = the two segments of the
// last bnpc_view_counts (rows 0, 1; a third row of the batch is their sum)
static int mh_counts(bnpc_ctx *c, int src, int64_t G, const int **n1,
                     const int **n0, int *sum_row, int64_t row0 = 0,
                     int64_t G_all = -1)
{
    *sum_row = -1;
    if (src == 0) {
        // (a slice of a batch beyond the pinned budget: rows row0 ... of the
        // G_all resident ones)
        if (G_all < 0) G_all = G;
        ARGCHK(c->lab_cnt.p && c->lab_K == G_all && row0 >= 0
               && row0 + G <= G_all,
               "the batch does not match the resident per-cluster counts");
        *n1 = (const int *)c->lab_cnt.p + (size_t)row0 * c->M;
        *n0 = *n1 + (size_t)G_all * c->M;
    } else {
        ARGCHK(c->cnt.p && c->cnt_rows == 2 && (G == 2 || G == 3),
               "the batch does not match the last view counts");
        *n1 = (const int *)c->cnt.p;
        *n0 = *n1 + (size_t)2 * c->M;
        if (G == 3) *sum_row = 2;
    }
    return 0;
}

// rows [g0, g0 + Gp) of the batch
static int mh_screen_launch(bnpc_ctx *c, int src, const bnpc_mh_args *a,
                            const MHPin &dev, int64_t g0 = 0, int64_t Gp = -1,
                            DoneSignal sig = {nullptr, nullptr, 0},
                            int64_t row0 = 0, int64_t G_all = -1)
{
    const int *n1, *n0;
    int sum_row;
    if (int rc = mh_counts(c, src, a->G, &n1, &n0, &sum_row, row0, G_all))
        return rc;
    if (Gp < 0) Gp = a->G;
    const size_t at = (size_t)g0 * a->M;
    ARGCHK(sum_row < 0 || (g0 == 0 && Gp == a->G),
           "a batch with a summed row is screened whole");
    MHScreenConst k;
    for (int i = 0; i < 8; i++) k.sd[i] = i < a->n_sd ? a->sd[i] : 1.0;
    k.FP = a->FP;
    k.FN = a->FN;
    k.p = a->p;
    k.q = a->q;
    k.tmin32 = (float)a->tmin;
    k.tmax32 = (float)a->tmax;
    k.uniform_prior = a->uniform_prior;
    const long long GM = (long long)Gp * a->M;
    BNPC_LAUNCH(k_mh_screen, dim3((unsigned)((GM + 255) / 256)),
                       dim3(256), 0, c->stream, dev.theta + at, n1 + at,
                       n0 + at, dev.sd_idx + at, dev.U + at, dev.u + at, GM,
                       (int)a->M, sum_row, k, dev.flags + at,
                       c->tun.screen_theta ? dev.new32 + at : nullptr, sig);
    HIPCHK(hipGetLastError());
    return 0;
}

static int mh_screen_argchk(const bnpc_ctx *c, const bnpc_mh_args *a)
{
    ARGCHK(c && a, "NULL argument");
    ARGCHK(a->G > 0 && a->M == c->M, "batch shape does not match the context");
    ARGCHK(a->old_theta && a->sd && a->n_sd >= 1 && a->n_sd <= 8 && a->sd_idx
           && a->U && a->u, "NULL argument");
    ARGCHK(a->FP > 0.0 && a->FP < 1.0 && a->FN > 0.0 && a->FN < 1.0,
           "error rates must lie in (0, 1)");
    return 0;
}

extern "C" int bnpc_mh_screen(bnpc_ctx *c, int counts_src,
                              const bnpc_mh_args *a, uint8_t *flags,
                              float *new32)
{
    if (int rc = mh_screen_argchk(c, a)) return rc;
    ARGCHK(flags, "flags is NULL");
    HIPCHK(hipSetDevice(c->device));
    const size_t E = (size_t)a->G * a->M;
    MHPin h, d;
    if (mh_pin_get(c, E, h, d)) return 1;
    memcpy(h.U, a->U, E * 8);
    memcpy(h.u, a->u, E * 8);
    memcpy(h.sd_idx, a->sd_idx, E * 4);
    memcpy(h.theta, a->old_theta, E * 4);
    if (int rc = mh_screen_launch(c, counts_src, a, d)) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(flags, h.flags, E);
    // (defined where the flag is 3)
    if (new32) memcpy(new32, h.new32, E * sizeof(float));
    return 0;
}

// bnpc_mh_batch with the device screen in front (include/bnpc_hip.h).
// `pending`: counts the device is still writing into pinned memory behind the
// stream (n1 rows then n0 rows, a->G x M each): copied into a->n1 / a->n0
// once the first screen has been waited for - before the host reads any.
static bool mh_screen_applies(const bnpc_ctx *c, const bnpc_mh_args *a)
{
    return !(a->trans_prob || !c->tun.mh_screen || a->screen
             || a->G * a->M < MH_SCREEN_MIN);
}

// The counts a fused launch is still writing behind the stream, taken now
// (a batch that will not wait for its first screen before the host reads them)
static int mh_take_pending(bnpc_ctx *c, const bnpc_mh_args *a,
                           const int **pending)
{
    if (!*pending) return 0;
    const size_t E = (size_t)a->G * a->M;
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy((void *)a->n1, *pending, E * sizeof(int32_t));
    memcpy((void *)a->n0, *pending + E, E * sizeof(int32_t));
    *pending = nullptr;
    return 0;
}

static int mh_batch_dev_impl(bnpc_ctx *c, const bnpc_host_kernels *k,
                             bnpc_mt19937 *rng, const bnpc_mh_args *a,
                             int counts_src, int *status, const int *pending,
                             int64_t row0 = 0, int64_t G_all = -1)
{
    ARGCHK(c && a && status, "NULL argument");
    if (!mh_screen_applies(c, a)) {
        ARGCHK(!pending, "counts still pending");
        // (the step's parameter batch without its screen: a walker's draws
        // are not wanted - the scored batch of a move, counts_src 1, runs
        // WHILE one walks and leaves it alone)
        if (counts_src == 0) mh_ahead_drop(c);
        return bnpc_mh_batch(k, rng, a, status);
    }
    if (int rc = mh_screen_argchk(c, a)) return rc;
    HIPCHK(hipSetDevice(c->device));
    {
        // The pinned block has a budget (a first step that is a
        // move at config 5 brings a batch of K0 x M = 158 M entries - 5.7 GB
        // pinned per chain without one).  A larger batch is screened in
        // slices of whole rows that reuse the block: the draws are taken row
        // by row in the reference's order either way, and nothing of a row
        // depends on another.
        size_t off[6];
        const size_t per_row = mh_pin_offsets((size_t)a->M, off);
        const int64_t rows_max = std::max<int64_t>(
            1, (int64_t)(c->tun.mh_pin_max / per_row));
        if (a->G > rows_max && counts_src == 0 && row0 == 0 && G_all < 0) {
            if (int rc = mh_take_pending(c, a, &pending)) return rc;
            *status = 0;
            for (int64_t g0 = 0; g0 < a->G; g0 += rows_max) {
                const int64_t Gs = std::min<int64_t>(rows_max, a->G - g0);
                const size_t at = (size_t)g0 * a->M;
                bnpc_mh_args b = *a;
                b.G = Gs;
                b.old_theta += at;
                b.n1 += at;
                b.n0 += at;
                if (b.known_theta) {
                    b.known_theta += at;
                    b.known_prior += at;
                }
                b.sd_idx += at;
                b.U += at;
                b.u += at;
                b.new_theta += at;
                if (b.prior_out) b.prior_out += at;
                b.A += at;
                b.log_prob += g0;
                b.declined += g0;
                int st = 0;
                if (int rc = mh_batch_dev_impl(c, k, rng, &b, 0, &st, nullptr,
                                               g0, a->G))
                    return rc;
                // (an element left to SciPy: the caller puts the stream back
                // and walks the whole batch by the binding - no point in
                // going on)
                if (st) {
                    *status = 1;
                    return 0;
                }
            }
            return 0;
        }
    }
    static const bool trace = [] {      // BNPC_TIMING=mh
        const char *e = getenv("BNPC_TIMING");
        return e && strstr(e, "mh");
    }();
    timespec ts0, ts1;
    if (trace) clock_gettime(CLOCK_MONOTONIC, &ts0);
    const int64_t G = a->G, M = a->M;
    const size_t E = (size_t)G * M;
    MHPin h, d;
    if (const int prc = mh_pin_get(c, E, h, d, true)) {
        if (prc != MH_PIN_NO_MEMORY) return 1;
        // no pinned block: the batch without its screen (the exact arithmetic
        // of every entry on the team) instead of a failed step
        if (int rc = mh_take_pending(c, a, &pending)) return rc;
        return bnpc_mh_batch(k, rng, a, status);
    }
    // Parts in a pipeline (batches of 4 rows and more): the draws of part
    // p + 1 are taken while the device screens part p, and the host evaluates
    // what a screen left while the next one runs.  Two halves for the batches
    // of a config-3 step; a LARGE batch (configs 4 and 5: 65 000 - 250 000
    // entries, 0.2 ms of draws) is cut into up to 8 parts, two of them
    // issued up front; while the team evaluates what the screen left of part
    // p, its rank 0 - this thread, the only one that touches the stream -
    // first takes the draws of part p + 2, stages them and launches their
    // screen (bnpc_mh_rank0_hook), then joins the evaluation.  (A helper
    // std::thread did the issuing earlier in round 4: a thread started per
    // batch on whatever core is free took 80-120 us for the draws of a part
    // that this thread, warm, takes in 27 - tools/draws_bench.py.)
    // Draws taken ahead (MhAhead above): adopted iff the live stream stands
    // where the walker's stood when it began to draw.
    MhAhead *ah = c->ahead;
    int64_t ahead_rows = 0;
    if (ah && ah->active) {
        bool ok = rng && row0 == 0 && G_all < 0 && counts_src == 0
            && ah->M == M && ah->n_sd == a->n_sd && c->tun.mh_ahead != 3;
        if (ok) {
            int known;
            while ((known = ah->start_known.load(std::memory_order_acquire))
                   == 0)
                __builtin_ia32_pause();
            ok = known == 1 && mt_same_position(ah->start, *rng);
        }
        if (!ok) {
            mh_ahead_drop(c);
        } else {
            ahead_rows = std::min<int64_t>(ah->rows, G);
            c->ahead_taken++;
            c->ahead_rows_taken += ahead_rows;
        }
    }
    // the walker's part of the batch is over: the live stream continues from
    // the state kept after the last row taken from it
    auto ahead_close = [&]() {
        if (!ahead_rows || !ah->active) return;
        while (ah->rows_ready.load(std::memory_order_acquire) < ahead_rows)
            __builtin_ia32_pause();
        *rng = ah->after_row[(size_t)(ahead_rows - 1)];
        mh_ahead_drop(c);
    };
    const bool threaded = rng && c->tun.done_words && G >= 4 && counts_src == 0
        && E >= MH_THREADED_MIN;     // "pipelined on rank 0"
    int parts = (G >= 4 && counts_src == 0) ? 2 : 1;
    if (threaded)
        parts = (int)std::max<int64_t>(2, std::min<int64_t>(
            std::min<int64_t>(8, G), (int64_t)(E / (MH_THREADED_MIN / 2))));
    int64_t cut[9];
    for (int p = 0; p <= parts; p++) cut[p] = G * p / parts;
    // ONE team job for the whole batch (round 6; below): its ranks wait for a
    // part's verdicts at the first row of it they touch
    const bool one_job = c->tun.done_words && parts >= 2;
    for (int p = 0; p < 2; p++)
        if (!c->mh_ev[p])
            HIPCHK(hipEventCreateWithFlags(&c->mh_ev[p],
                                           hipEventDisableTiming));
    SideLane lane(c);
    unsigned done_seq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double t_draws_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // draws + staging + launch of one part (either thread)
    auto issue_part = [&](int p) -> int {
        const int64_t g0 = cut[p], Gp = cut[p + 1] - cut[p];
        const size_t at = (size_t)g0 * M, n = (size_t)Gp * M;
        timespec tq0, tq1;
        if (trace) clock_gettime(CLOCK_MONOTONIC, &tq0);
        if (rng) {
            int64_t lo = g0;
            const int64_t hi = g0 + Gp;
            if (lo < ahead_rows) {      // rows the walker took (or is taking)
                const int64_t upto = std::min(hi, ahead_rows);
                while (ah->rows_ready.load(std::memory_order_acquire) < upto)
                    __builtin_ia32_pause();
                lo = upto;
            }
            if (lo < hi) {              // rows this thread draws
                ahead_close();
                const size_t lat = (size_t)lo * M;
                if (int rc = bnpc_mt_mh_draws_to(rng, hi - lo, M, a->n_sd,
                                                 h.sd_idx + lat, h.U + lat,
                                                 h.u + lat, threaded))
                    return rc;
            }
            if (trace) {
                clock_gettime(CLOCK_MONOTONIC, &tq1);
                t_draws_us[p] = (tq1.tv_sec - tq0.tv_sec) * 1e6
                    + (tq1.tv_nsec - tq0.tv_nsec) / 1e3;
            }
        } else {
            memcpy(h.U + at, a->U + at, n * 8);
            memcpy(h.u + at, a->u + at, n * 8);
            memcpy(h.sd_idx + at, a->sd_idx + at, n * 4);
        }
        memcpy(h.theta + at, a->old_theta + at, n * 4);
        // (one word for all parts: the launches of a stream finish in order)
        const DoneSignal sig = make_signal(c, threaded || one_job ? 0 : p,
                                           &done_seq[p]);
        if (int rc = mh_screen_launch(c, counts_src, a, d, g0, Gp, sig, row0,
                                      G_all))
            return rc;
        if (!done_seq[p]) HIPCHK(hipEventRecord(c->mh_ev[p & 1], c->stream));
        return 0;
    };
    timespec t_issued[8], t_got[8], t_waited[8], t_hosted[8];
    // issue one part, keep its time for the trace
    auto issue_timed = [&](int p) -> int {
        const int rc = issue_part(p);
        if (trace) clock_gettime(CLOCK_MONOTONIC, &t_issued[p]);
        return rc;
    };
    int next_issue = 0;
    {
        // (parts whose draws are there cost this thread a copy and a launch:
        // all of them up front)
        int up_front = threaded ? std::min(2, parts) : parts;
        while (up_front < parts && ahead_rows >= cut[up_front + 1]
               && ah->rows_ready.load(std::memory_order_acquire)
                   >= cut[up_front + 1])
            up_front++;
        for (; next_issue < up_front; next_issue++)
            if (int rc = issue_timed(next_issue)) return rc;
    }
    if (trace) clock_gettime(CLOCK_MONOTONIC, &ts1);
    int64_t kept = 0;
    double weighted_per_row = 0.0;      // flagged entries of the parts so far
    *status = 0;
    bool all_signalled = one_job;
    for (int p = 0; p < next_issue; p++)
        all_signalled = all_signalled && done_seq[p] != 0;
    if (all_signalled) {
        // ---- the batch as ONE team job ---------------------------------
        // Round 5 evaluated part after part: a team job per part, its rank 0
        // issuing the next part but one before it joined - seven wake-ups,
        // seven barriers and seven tails per config-5 batch, 60-70 us a part
        // even with the draws taken ahead.  Now the team is started once:
        // rank 0 (this thread) first issues what is left to issue - a copy
        // and a launch per part when a walker took the draws, else the draws
        // too -, copies the counts a fused launch wrote, and joins; every
        // rank, before it touches a row, waits until the part the row lies
        // in has been launched and the device's completion word has reached
        // that launch's number (the parts finish in order on one stream).
        std::atomic<int> issued(next_issue), ready(0), failed(0);
        const volatile unsigned *word = c->done_pin;    // slot 0
        const std::function<bool(int64_t)> gate = [&](int64_t g) -> bool {
            int p = 0;
            while (p + 1 < parts && g >= cut[p + 1]) p++;
            if (ready.load(std::memory_order_acquire) > p) return true;
            for (long spins = 0;; spins++) {
                if (failed.load(std::memory_order_relaxed)) return false;
                if (issued.load(std::memory_order_acquire) > p
                    && (int)(*word - done_seq[p]) >= 0)
                    break;
                if (spins < 4000) __builtin_ia32_pause();
                else std::this_thread::yield();
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            int seen = ready.load(std::memory_order_relaxed);
            while (seen < p + 1
                   && !ready.compare_exchange_weak(seen, p + 1,
                                                   std::memory_order_release))
            {}
            return true;
        };
        int hook_rc = 0;
        const int *counts_dev = pending;
        const std::function<void()> hook = [&]() {
            while (next_issue < parts && !hook_rc) {
                hook_rc = issue_timed(next_issue);
                if (!hook_rc && !done_seq[next_issue]) hook_rc = 1;
                next_issue++;
                issued.store(next_issue, std::memory_order_release);
            }
            if (hook_rc) {
                failed.store(1, std::memory_order_relaxed);
                return;
            }
            if (counts_dev) {   // behind the first screen: the counts are there
                if (!gate(0)) return;
                memcpy((void *)a->n1, counts_dev, E * sizeof(int32_t));
                memcpy((void *)a->n0, counts_dev + E, E * sizeof(int32_t));
            }
        };
        bnpc_mh_args b = *a;
        b.sd_idx = h.sd_idx;
        b.U = h.U;
        b.u = h.u;
        b.screen = h.flags;
        b.screen_theta = h.new32;
        if (pending) {          // the evaluation reads them where they are
            b.n1 = pending;
            b.n0 = pending + E;
        }
        int64_t counts[3] = {0, 0, 0};
        b.flag_counts = counts;
        b.flagged_estimate = std::max<int64_t>(
            1, (int64_t)(c->mh_flagged_share * (double)E));
        int st = 0;
        bnpc_mh_rank0_hook(&hook);
        bnpc_mh_row_gate(&gate);
        int rc = bnpc_mh_batch(k, nullptr, &b, &st);
        bnpc_mh_rank0_hook(nullptr);
        bnpc_mh_row_gate(nullptr);
        pending = nullptr;
        if (rc == 0 && hook_rc) {
            bnpc_set_error("a part of a screened parameter batch could not "
                           "be issued");
            rc = hook_rc == 2 ? 2 : 1;
        }
        if (rc) {
            (void)hipStreamSynchronize(c->stream);
            return rc;
        }
        if (st) *status = 1;
        kept = counts[0] + counts[1] + counts[2];
        c->mh_flagged_share = ((double)(counts[0] + counts[1])
            + 0.5 * (double)counts[2]) / (double)E;
        if (trace)
            for (int p = 0; p < parts; p++) t_got[p] = t_waited[p] =
                t_hosted[p] = t_issued[p];
    } else
    for (int p = 0; p < parts; p++) {
        const int64_t g0 = cut[p], Gp = cut[p + 1] - cut[p];
        const size_t at = (size_t)g0 * M;
        if (trace) clock_gettime(CLOCK_MONOTONIC, &t_got[p]);
        if (done_seq[p]) {
            if (int rc = wait_done(c, threaded || one_job ? 0 : p, done_seq[p]))
                return rc;
        } else {
            HIPCHK(hipEventSynchronize(c->mh_ev[p & 1]));
        }
        if (trace) clock_gettime(CLOCK_MONOTONIC, &t_waited[p]);
        if (pending) {          // the counts were written before the screen
            memcpy((void *)a->n1, pending, E * sizeof(int32_t));
            memcpy((void *)a->n0, pending + E, E * sizeof(int32_t));
            pending = nullptr;
        }
        bnpc_mh_args b = *a;
        b.G = Gp;
        b.old_theta += at;
        b.n1 += at;
        b.n0 += at;
        if (b.known_theta) {
            b.known_theta += at;
            b.known_prior += at;
        }
        b.sd_idx = h.sd_idx + at;
        b.U = h.U + at;
        b.u = h.u + at;
        b.new_theta += at;
        if (b.prior_out) b.prior_out += at;
        b.A += at;
        b.log_prob += g0;
        b.declined += g0;
        b.screen = h.flags + at;
        b.screen_theta = h.new32 + at;
        // (the flags sit in memory the device has just written: a pass over
        // them - to size the team, to count what was left - is 10-15 us of
        // misses per part on the calling thread; the team counts while it
        // works, and the parts of one batch leave about the same share)
        int64_t counts[3] = {0, 0, 0};
        b.flag_counts = counts;
        b.flagged_estimate = p == 0 ? 0
            : std::max<int64_t>(1, (int64_t)(weighted_per_row * (double)Gp));
        int st = 0;
        // rank 0 of this part's evaluation issues the next part but one first
        int hook_rc = 0;
        const std::function<void()> hook = [&]() {
            hook_rc = issue_timed(next_issue++);
        };
        if (next_issue < parts) bnpc_mh_rank0_hook(&hook);
        int rc = bnpc_mh_batch(k, nullptr, &b, &st);
        bnpc_mh_rank0_hook(nullptr);
        if (rc == 0) rc = hook_rc;
        if (rc) {
            (void)hipStreamSynchronize(c->stream);
            return rc;
        }
        if (st) *status = 1;
        kept += counts[0] + counts[1] + counts[2];
        weighted_per_row = ((double)(counts[0] + counts[1])
            + 0.5 * (double)counts[2]) / (double)Gp;
        if (trace) clock_gettime(CLOCK_MONOTONIC, &t_hosted[p]);
    }
    ahead_close();
    c->screened += (int64_t)E;
    c->screen_kept += kept;
    if (trace) {
        timespec ts3;
        clock_gettime(CLOCK_MONOTONIC, &ts3);
        auto us = [](const timespec &x, const timespec &y) {
            return (y.tv_sec - x.tv_sec) * 1e6 + (y.tv_nsec - x.tv_nsec) / 1e3;
        };
        fprintf(stderr, "[mh_batch_dev] %lld x %lld in %d part(s): draws + "
                "staging + launches %.1f us, waits + host %.1f us (%lld left)\n",
                (long long)G, (long long)M, parts, us(ts0, ts1), us(ts1, ts3),
                (long long)kept);
        if (threaded)
            for (int p = 0; p < parts; p++)
                fprintf(stderr, "[mh_batch_dev]   part %d: issued at %.1f "
                        "(its draws %.1f), picked up at %.1f, screened at "
                        "%.1f, evaluated at %.1f us\n", p,
                        us(ts0, t_issued[p]), t_draws_us[p], us(ts0, t_got[p]),
                        us(ts0, t_waited[p]), us(ts0, t_hosted[p]));
    }
    if (*status != 0) {
        // the caller's view of the draws (the SciPy-level twin evaluates the
        // batch from them when the library hands an element back)
        memcpy(a->sd_idx, h.sd_idx, E * 4);
        memcpy(a->U, h.U, E * 8);
        memcpy(a->u, h.u, E * 8);
    }
    return 0;
}

// The second half of a restricted-Gibbs scan (bnpc_rg_scan_step,
// bnpc_sweeps.cpp) on ONE stream synchronisation: the column counts of the two
// launch clusters for the new assignment (labels: one per slot of the view),
// the draws of the parameter batch, its device screen - queued behind the
// counts it reads - then the counts are copied out and the host evaluates what
// the screen left.  Scored batches and contexts without the screen take the
// two calls one after the other.
int bnpc_rg_counts_and_batch(bnpc_ctx *c, const bnpc_host_kernels *k,
                             bnpc_mt19937 *rng, int view,
                             const int64_t *labels, const bnpc_mh_args *a,
                             int32_t *n1, int32_t *n0, int *status)
{
    ARGCHK(c && a && labels && n1 && n0 && status && rng, "NULL argument");
    const int64_t M = a->M;
    auto finish_rows = [&]() {
        if (a->G == 3)
            for (int64_t m = 0; m < M; m++) {       // the merged cluster
                n1[2 * M + m] = n1[m] + n1[M + m];
                n0[2 * M + m] = n0[m] + n0[M + m];
            }
    };
    const bool fused = !a->trans_prob && c->tun.mh_screen
        && a->G * M >= MH_SCREEN_MIN && !c->any_tile_pending();
    if (!fused) {
        if (int rc = bnpc_view_counts(c, view, labels, 2, n1, n0)) return rc;
        finish_rows();
        return bnpc_mh_batch_dev(c, k, rng, a, 1, status);
    }
    if (int rc = mh_screen_argchk(c, a)) return rc;
    ARGCHK(view >= 0 && view < BNPC_MAX_VIEWS, "view out of range");
    HIPCHK(hipSetDevice(c->device));
    c->cnt_rows = 0;
    const int *pending = nullptr;
    if (int rc = counts_from_masks(c, view,
            [=](int64_t s) -> int64_t { return labels[s]; }, 2, c->cnt, n1, n0,
            &pending))
        return rc;
    c->cnt_rows = 2;
    const size_t E = (size_t)a->G * M;
    MHPin h, d;
    if (mh_pin_get(c, E, h, d, true)) return 1;
    {
        // the rows a walker took under the scan's sums and loop (posted when
        // the scan had drawn its visiting order: one uniform per cell lay
        // between there and here), the rest by this thread
        const int64_t got = mh_ahead_adopt(c, rng, a->G, M, a->n_sd);
        if (got < a->G) {
            const size_t at = (size_t)got * M;
            if (int rc = bnpc_mt_mh_draws(rng, a->G - got, M, a->n_sd,
                                          h.sd_idx + at, h.U + at, h.u + at))
                return rc;
        }
    }
    memcpy(h.theta, a->old_theta, E * 4);
    unsigned done_seq = 0;
    const DoneSignal sig = make_signal(c, 0, &done_seq);
    if (int rc = mh_screen_launch(c, 1, a, d, 0, -1, sig)) return rc;
    if (int rc = wait_done(c, 0, done_seq)) return rc;
    if (pending) {
        const size_t half = (size_t)2 * M * sizeof(int32_t);
        memcpy(n1, pending, half);
        memcpy(n0, pending + (size_t)2 * M, half);
    }
    finish_rows();
    bnpc_mh_args b = *a;
    b.sd_idx = h.sd_idx;
    b.U = h.U;
    b.u = h.u;
    b.screen = h.flags;
    b.screen_theta = h.new32;
    int64_t counts[3] = {0, 0, 0};
    b.flag_counts = counts;
    if (int rc = bnpc_mh_batch(k, nullptr, &b, status)) return rc;
    c->screened += (int64_t)E;
    c->screen_kept += counts[0] + counts[1] + counts[2];
    if (*status != 0) {
        memcpy(a->sd_idx, h.sd_idx, E * 4);
        memcpy(a->U, h.U, E * 8);
        memcpy(a->u, h.u, E * 8);
    }
    return 0;
}

extern "C" int bnpc_mh_batch_dev(bnpc_ctx *c, const bnpc_host_kernels *k,
                                 bnpc_mt19937 *rng, const bnpc_mh_args *a,
                                 int counts_src, int *status)
{
    return mh_batch_dev_impl(c, k, rng, a, counts_src, status, nullptr);
}

// update_parameters in one call (libs/CRP.py:302-311): the per-cluster
// column counts for `assignment` (bnpc_colcounts_by_label; they stay resident
// for bnpc_ll_total) and the screened parameter batch on them, the screens
// queued behind the counts kernel - the counts reach a->n1 / a->n0 (the
// caller's arrays, K x M each) with the first screen's results.
extern "C" int bnpc_label_counts_and_batch(bnpc_ctx *c,
                                           const bnpc_host_kernels *k,
                                           bnpc_mt19937 *rng,
                                           const int64_t *assignment,
                                           const int64_t *ids,
                                           const bnpc_mh_args *a, int *status)
{
    ARGCHK(c && a && status && a->n1 && a->n0, "NULL argument");
    const bool fused = mh_screen_applies(c, a) && !c->any_tile_pending();
    const int *pending = nullptr;
    if (int rc = colcounts_by_label_impl(c, assignment, ids, a->G,
            (int32_t *)a->n1, (int32_t *)a->n0, fused ? &pending : nullptr))
        return rc;
    return mh_batch_dev_impl(c, k, rng, a, 0, status, pending);
}

extern "C" int bnpc_mh_screen_stats(bnpc_ctx *c, int64_t *screened,
                                    int64_t *kept)
{
    ARGCHK(c && screened && kept, "NULL argument");
    *screened = c->screened;
    *kept = c->screen_kept;
    return 0;
}

extern "C" int bnpc_ll_total_issue(bnpc_ctx *c, const float *theta, int64_t K,
                                   const double *FP, const double *FN, int E)
{
    ARGCHK(c && theta && FP && FN, "NULL argument");
    ARGCHK(E >= 1 && E <= BNPC_MAX_TRIALS, "E out of range");
    ARGCHK(K == c->lab_K && K > 0,
           "K does not match the resident counts (call "
           "bnpc_colcounts_by_label first)");
    ARGCHK(!c->total_pending, "a deferred total is already pending");
    for (int e = 0; e < E; e++)
        ARGCHK(FP[e] > 0.0 && FP[e] < 1.0 && FN[e] > 0.0 && FN[e] < 1.0,
               "error rates must lie in (0, 1)");
    HIPCHK(hipSetDevice(c->device));
    const size_t bytes = (size_t)K * c->M * sizeof(float);
    if (arena_reset(c)) return 1;
    const float *d_theta = (const float *)stage_in_place(c, theta, bytes);
    if (!d_theta) {
        if (ensure(c->theta, bytes)) return 1;
        if (h2d(c, c->theta.p, theta, bytes)) return 1;
        d_theta = (const float *)c->theta.p;
    }
    // a fixed number of blocks per problem size (the sum order must not
    // depend on anything else): one element per thread up to 256 blocks -
    // the parameters may be read in place from host memory, and every
    // further trip of a thread's loop would pay that latency again
    const long long KM = (long long)K * c->M;
    int blocks = (int)((KM + 255) / 256);
    if (blocks > TOTAL_BLOCKS) blocks = TOTAL_BLOCKS;
    if (blocks < 1) blocks = 1;
    // the partial sums land in a small pinned buffer of their own (a
    // deferred total must survive the calls made before it is picked up)
    if (!c->pin_small)
        HIPCHK(hipHostMalloc(&c->pin_small, TOTAL_BLOCKS * 4 * sizeof(double),
                             hipHostMallocDefault));
    void *d_part = nullptr;
    const bool in_place = c->tun.zero_copy
        && hipHostGetDevicePointer(&d_part, c->pin_small, 0) == hipSuccess;
    if (!in_place) {
        if (ensure(c->partial, TOTAL_BLOCKS * 4 * sizeof(double))) return 1;
        d_part = c->partial.p;
    }
    double fp[4] = {0.5, 0.5, 0.5, 0.5}, fn[4] = {0.5, 0.5, 0.5, 0.5};
    for (int e = 0; e < E; e++) {
        fp[e] = FP[e];
        fn[e] = FN[e];
    }
    const int *n1 = (const int *)c->lab_cnt.p;
    const int *n0 = n1 + (size_t)K * c->M;
    BNPC_LAUNCH(k_ll_total, dim3(blocks), dim3(256), 0, c->stream,
                       d_theta, n1, n0, KM, E, fp[0], fn[0], fp[1], fn[1],
                       fp[2], fn[2], fp[3], fn[3], (double *)d_part,
                       in_place ? make_signal(c, 2, &c->total_seq)
                                : DoneSignal{nullptr, nullptr, 0});
    if (!in_place) c->total_seq = 0;
    HIPCHK(hipGetLastError());
    if (!in_place)
        HIPCHK(hipMemcpyAsync(c->pin_small, c->partial.p,
                              (size_t)blocks * 4 * sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
    c->total_pending = true;
    c->total_blocks = blocks;
    c->total_E = E;
    return 0;
}

extern "C" int bnpc_ll_total_wait(bnpc_ctx *c, double *out)
{
    ARGCHK(c && out, "NULL argument");
    ARGCHK(c->total_pending, "no deferred total is pending");
    HIPCHK(hipSetDevice(c->device));
    c->total_pending = false;
    if (int rc = wait_done(c, 2, c->total_seq)) return rc;
    const double *p = (const double *)c->pin_small;
    for (int e = 0; e < c->total_E; e++) {
        double sum = 0.0;
        for (int b = 0; b < c->total_blocks; b++) sum += p[b * 4 + e];
        out[e] = sum;
    }
    return 0;
}

extern "C" int bnpc_ll_total(bnpc_ctx *c, const float *theta, int64_t K,
                             const double *FP, const double *FN, int E,
                             double *out)
{
    ARGCHK(out, "NULL argument");
    int rc = bnpc_ll_total_issue(c, theta, K, FP, FN, E);
    if (rc) return rc;
    return bnpc_ll_total_wait(c, out);
}

extern "C" int bnpc_last_launch(const bnpc_ctx *c, char *name, int len,
                                int64_t *K, int *mutation_chunks)
{
    ARGCHK(c && name && len > 0, "NULL argument");
    ARGCHK(c->last_kw != 0, "no previous bnpc_ll_theta / bnpc_ll_tables call");
    strncpy(name, c->last_name, len - 1);
    name[len - 1] = 0;
    if (K) *K = c->last_K;
    if (mutation_chunks) *mutation_chunks = c->last_ms;
    return 0;
}

// the whole evaluation of the last call again - element tables, sums, combine -
// `reps` times between two HIP events
extern "C" int bnpc_bench_ll_full(bnpc_ctx *c, int reps, float *ms_per_call)
{
    ARGCHK(c && ms_per_call, "NULL argument");
    ARGCHK(reps >= 1, "reps must be positive");
    ARGCHK(c->last_kw != 0, "no previous bnpc_ll_theta / bnpc_ll_tables call");
    HIPCHK(hipSetDevice(c->device));
    const View &v = c->views[c->last_view];
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; r++) {
        int rc;
        double *o = c->last_out;
        const int64_t K = c->last_K, ldo = c->last_ldo;
        const bool ft = c->last_from_theta;
        const double FP = c->last_FP, FN = c->last_FN;
        const int MS = c->last_ms, mc = c->last_mchunk;
        if (c->last_kw == -1)
            rc = issue_seqp(c, v, K, ldo, o);
        else
        switch (c->last_kw) {
        case 8: rc = launch_ll<8>(c, v, K, ldo, ft, FP, FN, o, MS, mc); break;
        case 4: rc = launch_ll<4>(c, v, K, ldo, ft, FP, FN, o, MS, mc); break;
        case 2: rc = launch_ll<2>(c, v, K, ldo, ft, FP, FN, o, MS, mc); break;
        default: rc = launch_ll<1>(c, v, K, ldo, ft, FP, FN, o, MS, mc); break;
        }
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_per_call = ms / reps;
    return 0;
}

extern "C" int bnpc_launch_timers(bnpc_ctx *c, int on, double *device_ms,
                                  int64_t *launches)
{
    ARGCHK(c, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    if (on) {
        g_timers.used = 0;
        g_timers.on = true;
        return 0;
    }
    g_timers.on = false;
    HIPCHK(hipDeviceSynchronize());
    double sum = 0.0;
    for (size_t i = 0; i + 1 < g_timers.used; i += 2) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, g_timers.ev[i], g_timers.ev[i + 1]));
        sum += ms;
    }
    if (device_ms) *device_ms = sum;
    if (launches) *launches = (int64_t)(g_timers.used / 2);
    g_timers.used = 0;
    return 0;
}

extern "C" int bnpc_bench_ll(bnpc_ctx *c, int reps, float *ms_per_launch)
{
    ARGCHK(c && ms_per_launch, "NULL argument");
    ARGCHK(reps >= 1, "reps must be positive");
    ARGCHK(c->last_kw != 0, "no previous bnpc_ll_theta / bnpc_ll_tables call");
    HIPCHK(hipSetDevice(c->device));
    const View &v = c->views[c->last_view];
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; r++) {
        int rc;
        double *o = c->last_out;
        if (c->last_kw == -1)
            rc = issue_seqp(c, v, c->last_K, c->last_ldo, o);
        else
        switch (c->last_kw) {
        case 8: rc = issue_ll<8>(c, v, c->last_K, c->last_ldo, o, c->last_ms, c->last_mchunk); break;
        case 4: rc = issue_ll<4>(c, v, c->last_K, c->last_ldo, o, c->last_ms, c->last_mchunk); break;
        case 2: rc = issue_ll<2>(c, v, c->last_K, c->last_ldo, o, c->last_ms, c->last_mchunk); break;
        default: rc = issue_ll<1>(c, v, c->last_K, c->last_ldo, o, c->last_ms, c->last_mchunk); break;
        }
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_per_launch = ms / reps;
    return 0;
}

extern "C" int bnpc_timer_start(bnpc_ctx *c)
{
    ARGCHK(c, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    return 0;
}

extern "C" int bnpc_timer_stop(bnpc_ctx *c, float *ms)
{
    ARGCHK(c && ms, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return 0;
}

extern "C" int bnpc_sync(bnpc_ctx *c)
{
    ARGCHK(c, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
