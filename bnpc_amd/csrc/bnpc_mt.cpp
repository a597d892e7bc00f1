// bnpc_mt.cpp - bulk draws from NumPy's legacy MT19937 stream.
//
// The draws of the parameter moves are the one sequential stretch of a step
// that nothing can be overlapped with beyond one cluster's worth (the stream
// is a single sequence): 5.3 generator words per matrix element.  The words
// are the same as mt_next32's; they are produced block-wise - the state
// refill and the tempering are loops without dependences, compiled for
// AVX-512, AVX2 and the x86-64 baseline (the loader picks one).

#include <stdint.h>

#include "bnpc_hip.h"
#include "bnpc_internal.h"

#if defined(__x86_64__) && defined(__clang__) && !defined(__HIP_DEVICE_COMPILE__)
#define BNPC_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define BNPC_CLONES
#endif

namespace {

inline uint32_t temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

BNPC_CLONES void refill(uint32_t *__restrict__ k)
{
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, A = 0x9908b0dfu;
    // k[i] needs k[i + 1] and k[i + 397] as they were: both lie ahead
    for (int i = 0; i < 624 - 397; i++) {
        const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER);
        k[i] = k[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
    }
    // k[i - 227] is already new: 227 words behind, beyond any vector width
    for (int i = 624 - 397; i < 623; i++) {
        const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER);
        k[i] = k[i - 227] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
    }
    const uint32_t y = (k[623] & UPPER) | (k[0] & LOWER);
    k[623] = k[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
}

}  // namespace

#if defined(__x86_64__) && defined(__clang__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
// The state refill with explicit 512-bit operations (the compiler's own
// vectorisation of the loops above manages 0.30 ns per word, this 0.06): the
// same recurrence in blocks of 16 words - a block reads itself, the word after
// it and a block 397 ahead (first stretch) or 227 behind (second stretch, all
// of it already new), none of which it has written yet.
__attribute__((target("avx512f"))) static inline __m512i
twist_512(const uint32_t *cur, const uint32_t *far)
{
    const __m512i upper = _mm512_set1_epi32((int)0x80000000u);
    const __m512i lower = _mm512_set1_epi32(0x7fffffff);
    const __m512i matrix = _mm512_set1_epi32((int)0x9908b0dfu);
    const __m512i one = _mm512_set1_epi32(1), zero = _mm512_setzero_si512();
    const __m512i y = _mm512_or_si512(
        _mm512_and_si512(_mm512_loadu_si512((const void *)cur), upper),
        _mm512_and_si512(_mm512_loadu_si512((const void *)(cur + 1)), lower));
    const __m512i mag = _mm512_and_si512(
        _mm512_sub_epi32(zero, _mm512_and_si512(y, one)), matrix);
    return _mm512_xor_si512(
        _mm512_xor_si512(_mm512_loadu_si512((const void *)far),
                         _mm512_srli_epi32(y, 1)), mag);
}

__attribute__((target("avx512f"))) static void refill_512(uint32_t *k)
{
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, A = 0x9908b0dfu;
    int i = 0;
    for (; i + 16 <= 624 - 397; i += 16)
        _mm512_storeu_si512((void *)(k + i), twist_512(k + i, k + i + 397));
    for (; i < 624 - 397; i++) {
        const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER);
        k[i] = k[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
    }
    for (; i + 16 <= 623; i += 16)
        _mm512_storeu_si512((void *)(k + i), twist_512(k + i, k + i - 227));
    for (; i < 623; i++) {
        const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER);
        k[i] = k[i - 227] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
    }
    const uint32_t y = (k[623] & UPPER) | (k[0] & LOWER);
    k[623] = k[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
}
#define BNPC_HAVE_REFILL_512 1
#endif

// the refill every consumer of the stream uses (mt_next32 included)
void mt_refill_block(uint32_t *key)
{
#ifdef BNPC_HAVE_REFILL_512
    static const bool wide = __builtin_cpu_supports("avx512f");
    if (wide) {
        refill_512(key);
        return;
    }
#endif
    refill(key);
}

namespace {

BNPC_CLONES void doubles_from(const uint32_t *__restrict__ k,
                              double *__restrict__ o, int64_t pairs)
{
    for (int64_t j = 0; j < pairs; j++) {
        const int32_t a = (int32_t)(temper(k[2 * j]) >> 5);
        const int32_t b = (int32_t)(temper(k[2 * j + 1]) >> 6);
        o[j] = (a * 67108864.0 + b) / 9007199254740992.0;
    }
}

BNPC_CLONES void masked_from(const uint32_t *__restrict__ k,
                             uint32_t *__restrict__ o, int n, uint32_t mask)
{
    for (int j = 0; j < n; j++) o[j] = temper(k[j]) & mask;
}

}  // namespace

#if defined(__x86_64__) && defined(__clang__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define BNPC_HAVE_AVX512_DRAWS 1
// doubles_from for whole blocks of 8 doubles (16 state words) with explicit
// 512-bit operations: the words of a pair share a 64-bit lane (low half the
// first, high half the second), so a = low >> 5 and b = high >> 6 fall out of
// two 64-bit shifts without any shuffle; both are below 2^27 and become
// doubles through the 2^52 bit pattern; a * 2^26 + b < 2^53 and the division
// by 2^53 are exact, so the result has the bits of the scalar expression.
// STREAM: non-temporal stores (o + j 64-byte aligned): a large destination in
// pinned memory that the device reads next - an ordinary store first fetches
// the line for ownership, all the way from DRAM once the device's reads have
// had it written back (measured: 80-120 us per 35 000-entry part of a
// config-5 batch against 27 us into a cache-resident array).
template <bool STREAM>
__attribute__((target("avx512f"))) static int64_t
doubles_from_512(const uint32_t *k, double *o, int64_t pairs)
{
    const __m512i c1 = _mm512_set1_epi32((int)0x9d2c5680u);
    const __m512i c2 = _mm512_set1_epi32((int)0xefc60000u);
    const __m512i low = _mm512_set1_epi64(0xffffffffll);
    const __m512i magic = _mm512_set1_epi64(0x4330000000000000ll);
    const __m512d magic_d = _mm512_castsi512_pd(magic);
    const __m512d two26 = _mm512_set1_pd(67108864.0);
    const __m512d inv53 = _mm512_set1_pd(1.0 / 9007199254740992.0);
    int64_t j = 0;
    for (; j + 8 <= pairs; j += 8) {
        __m512i y = _mm512_loadu_si512((const void *)(k + 2 * j));
        y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 11));
        y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 7), c1));
        y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 15), c2));
        y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 18));
        const __m512i a = _mm512_srli_epi64(_mm512_and_si512(y, low), 5);
        const __m512i b = _mm512_srli_epi64(y, 32 + 6);
        const __m512d da = _mm512_sub_pd(
            _mm512_castsi512_pd(_mm512_or_si512(a, magic)), magic_d);
        const __m512d db = _mm512_sub_pd(
            _mm512_castsi512_pd(_mm512_or_si512(b, magic)), magic_d);
        const __m512d v = _mm512_mul_pd(
            _mm512_add_pd(_mm512_mul_pd(da, two26), db), inv53);
        if (STREAM)
            _mm512_stream_pd(o + j, v);
        else
            _mm512_storeu_pd(o + j, v);
    }
    return j;
}
#endif

// n x random_sample(); stream: see doubles_from_512 (the caller fences)
void mt_fill_double(bnpc_mt19937 *s, double *out, int64_t n, bool stream)
{
    int64_t i = 0;
    while (i < n) {
        if (s->pos >= 624) {
            mt_refill_block(s->key);
            s->pos = 0;
        }
        int64_t pairs = (624 - s->pos) / 2;
        if (pairs > n - i) pairs = n - i;
        if (pairs == 0) {               // a double straddles two blocks
            out[i++] = mt_double(s);
            continue;
        }
        int64_t done = 0;
#ifdef BNPC_HAVE_AVX512_DRAWS
        static const bool wide = __builtin_cpu_supports("avx512f");
        if (wide && stream) {
            // up to 7 doubles the plain way, until the destination is aligned
            int64_t peel = (int64_t)((64 - ((uintptr_t)(out + i) & 63)) & 63)
                / (int64_t)sizeof(double);
            if (peel > pairs) peel = pairs;
            doubles_from(s->key + s->pos, out + i, peel);
            done = peel + doubles_from_512<true>(s->key + s->pos + 2 * peel,
                                                 out + i + peel, pairs - peel);
        } else if (wide) {
            done = doubles_from_512<false>(s->key + s->pos, out + i, pairs);
        }
#endif
        doubles_from(s->key + s->pos + 2 * done, out + i + done, pairs - done);
        s->pos += (int32_t)(2 * pairs);
        i += pairs;
    }
}

#if defined(__x86_64__) && defined(__clang__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define BNPC_HAVE_COMPRESS 1
// The accepted candidates of whole blocks of 16, in order, by a register
// compress (the scalar slot-advance loop costs ~1.2 ns per candidate, as much
// as producing it): stops a block short of `need` outputs, so it never takes
// a candidate the one-by-one loop would not have taken.  Returns the outputs
// written; *used = candidates consumed.
__attribute__((target("avx512f"))) static int64_t
accept_blocks(const uint32_t *cand, int avail, uint32_t max, int32_t *out,
              int64_t need, int *used)
{
    const __m512i vmax = _mm512_set1_epi32((int)max);
    int j = 0;
    int64_t cnt = 0;
    while (j + 16 <= avail && cnt + 16 <= need) {
        const __m512i v = _mm512_loadu_si512((const void *)(cand + j));
        const __mmask16 m = _mm512_cmple_epu32_mask(v, vmax);
        _mm512_storeu_si512((void *)(out + cnt),
                            _mm512_maskz_compress_epi32(m, v));
        cnt += __builtin_popcount((unsigned)m);
        j += 16;
    }
    *used = j;
    return cnt;
}
#endif

// n x random_interval(max) for max < 2^32 (masked rejection on 32-bit draws)
void mt_fill_interval32(bnpc_mt19937 *s, uint32_t max, int32_t *out, int64_t n)
{
    if (max == 0) {
        for (int64_t i = 0; i < n; i++) out[i] = 0;
        return;
    }
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t cand[624];
    int64_t cnt = 0;
    while (cnt < n) {
        if (s->pos >= 624) {
            mt_refill_block(s->key);
            s->pos = 0;
        }
        const int avail = 624 - s->pos;
        masked_from(s->key + s->pos, cand, avail, mask);
        int j = 0;
#ifdef BNPC_HAVE_COMPRESS
        static const bool wide = __builtin_cpu_supports("avx512f");
        if (wide) cnt += accept_blocks(cand, avail, max, out + cnt, n - cnt, &j);
#endif
        // every candidate is written to the next free slot; the slot only
        // advances when it is accepted (no data-dependent branch)
        for (; j < avail && cnt < n; j++) {
            out[cnt] = (int32_t)cand[j];
            cnt += (cand[j] <= max);
        }
        s->pos += j;
    }
}

// ---------------------------------------------------------------------------
// NumPy's LEGACY Beta / Gamma samplers on the same stream: the draws behind
// CRP._init_cl_params_new (/root/reference/libs/CRP.py:183-188: one Beta per
// mutation with shapes p + #ones, q + #zeros of the cells), _init_cl_params
// (:155-180) and the restricted-Gibbs launch states (:563-567).  NumPy
// (unpinned dependency of the reference) implements np.random.beta with
// numpy/random/src/legacy/legacy-distributions.c; what is restated here is
// that file's published algorithm:
//   legacy_beta            Johnk's algorithm when both shapes are <= 1, else
//                          Ga / (Ga + Gb) from two standard gammas;
//   legacy_standard_gamma  shape == 1: -log(1 - U); shape < 1: the
//                          Ahrens-Dieter style rejection with U, an
//                          exponential and pow(); shape > 1: Marsaglia-Tsang
//                          with the squeeze 1 - 0.0331 x^4;
//   legacy_gauss           polar Box-Muller, the second variate cached in
//                          (has_gauss, gauss) - state that lives in NumPy's
//                          RandomState next to the bit generator and is
//                          shared with every other normal / gamma draw of the
//                          stream, so it is read and written here in place;
// with libm's pow / log / exp / sqrt, as NumPy calls them.  Pinned by
// tests/golden/rng_beta.npz (vectors drawn by NumPy itself on the reference's
// stack) and against the NumPy this process runs (tests/test_native_sweeps.py).
// ---------------------------------------------------------------------------
#include <math.h>

namespace {

// The stream as the samplers below consume it: the rest of the current
// 624-word block tempered AT ONCE (the vectorised loop of masked_from /
// doubles_from) into a local buffer, the words handed out from there - the
// same words in the same order as mt_next32, without a tempering chain and a
// refill test per word (a Beta draw takes ~9 words: 63 -> 50 ns per draw).
// The state block itself stays untempered; pos is written back on exit.
BNPC_CLONES void temper_block(const uint32_t *__restrict__ k,
                              uint32_t *__restrict__ o, int from)
{
    for (int j = from; j < 624; j++) o[j] = temper(k[j]);
}

struct Words {
    bnpc_mt19937 *s;
    uint32_t buf[624];
    int pos;
    explicit Words(bnpc_mt19937 *state) : s(state)
    {
        // (an exhausted block is refilled by the first draw, not here: a
        // call that draws nothing leaves the state as it found it)
        pos = (s->pos >= 624 || s->pos < 0) ? 624 : s->pos;
        if (pos < 624) temper_block(s->key, buf, pos);
    }
    ~Words() { s->pos = pos; }
    inline uint32_t next32()
    {
        if (pos >= 624) {
            mt_refill_block(s->key);
            temper_block(s->key, buf, 0);
            pos = 0;
        }
        return buf[pos++];
    }
    inline double next_double()
    {
        const int32_t a = (int32_t)(next32() >> 5);
        const int32_t b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

inline double lg_gauss(Words &w, bnpc_legacy_gauss *g)
{
    if (g->has_gauss) {
        const double t = g->gauss;
        g->has_gauss = 0;
        g->gauss = 0.0;
        return t;
    }
    double x1, x2, r2;
    do {
        x1 = 2.0 * w.next_double() - 1.0;
        x2 = 2.0 * w.next_double() - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    const double f = sqrt(-2.0 * log(r2) / r2);
    g->gauss = f * x1;
    g->has_gauss = 1;
    return f * x2;
}

inline double lg_exponential(Words &w)
{
    return -log(1.0 - w.next_double());
}

}  // namespace

// np.random.permutation(n) for n < 2^32: the legacy shuffle (one masked
// rejection draw per element, from the back) on the pre-tempered words
void mt_fill_permutation(bnpc_mt19937 *s, int64_t n, int64_t *out)
{
    for (int64_t i = 0; i < n; i++) out[i] = i;
    Words w(s);
    uint32_t mask = 0;
    for (int64_t i = n - 1; i >= 1; i--) {
        // smallest all-ones mask covering i (i only shrinks: recompute when
        // it drops below half the mask)
        if (mask == 0 || (uint32_t)i <= (mask >> 1)) {
            mask = (uint32_t)i;
            mask |= mask >> 1;
            mask |= mask >> 2;
            mask |= mask >> 4;
            mask |= mask >> 8;
            mask |= mask >> 16;
        }
        uint32_t v;
        while ((v = (w.next32() & mask)) > (uint32_t)i) {}
        const int64_t t = out[i];
        out[i] = out[v];
        out[v] = t;
    }
}

namespace {

double lg_standard_gamma(Words &w, bnpc_legacy_gauss *g, double shape)
{
    if (shape == 1.0) return lg_exponential(w);
    if (shape == 0.0) return 0.0;
    if (shape < 1.0) {
        for (;;) {
            const double U = w.next_double();
            const double V = lg_exponential(w);
            if (U <= 1.0 - shape) {
                const double X = pow(U, 1. / shape);
                if (X <= V) return X;
            } else {
                const double Y = -log((1 - U) / shape);
                const double X = pow(1.0 - shape + shape * Y, 1. / shape);
                if (X <= (V + Y)) return X;
            }
        }
    }
    const double b = shape - 1. / 3.;
    const double c = 1. / sqrt(9 * b);
    for (;;) {
        double X, V;
        do {
            X = lg_gauss(w, g);
            V = 1.0 + c * X;
        } while (V <= 0.0);
        V = V * V * V;
        const double U = w.next_double();
        if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return (b * V);
        if (log(U) < 0.5 * X * X + b * (1. - V + log(V))) return (b * V);
    }
}

inline double lg_beta(Words &w, bnpc_legacy_gauss *g, double a, double b)
{
    if ((a <= 1.0) && (b <= 1.0)) {
        for (;;) {                              // Johnk
            const double U = w.next_double();
            const double V = w.next_double();
            const double X = pow(U, 1.0 / a);
            const double Y = pow(V, 1.0 / b);
            const double XpY = X + Y;
            // (U + V == 0 has probability 2^-106: NumPy rejects it too)
            if ((XpY <= 1.0) && (U + V > 0.0)) {
                if (XpY > 0) return X / XpY;
                double logX = log(U) / a;
                double logY = log(V) / b;
                const double logM = logX > logY ? logX : logY;
                logX -= logM;
                logY -= logM;
                return exp(logX - log(exp(logX) + exp(logY)));
            }
        }
    }
    const double Ga = lg_standard_gamma(w, g, a);
    const double Gb = lg_standard_gamma(w, g, b);
    return Ga / (Ga + Gb);
}

}  // namespace

// np.random.gamma(shape, scale) of the legacy stream
// (numpy/random/src/legacy/legacy-distributions.c: legacy_gamma = scale *
// legacy_standard_gamma): the two draws of CRP.update_DP_alpha
// (libs/CRP.py:386-410)
double bnpc_legacy_gamma(bnpc_mt19937 *rng, bnpc_legacy_gauss *g, double shape,
                         double scale)
{
    Words w(rng);
    return scale * lg_standard_gamma(w, g, shape);
}

extern "C" int bnpc_mt_gamma(bnpc_mt19937 *rng, bnpc_legacy_gauss *g,
                             double shape, double scale, double *out)
{
    if (!rng || !g || !out || !(shape >= 0.0) || !(scale >= 0.0)) {
        bnpc_set_error("bad argument: mt_gamma");
        return 2;
    }
    *out = bnpc_legacy_gamma(rng, g, shape, scale);
    return 0;
}

// One profile row from the {ones, zeros} words of ONE cell's observations
// (a cluster it opens, libs/CRP.py:183-188): theta[m] = float32(clip(Beta(p +
// [x_m = 1], q + [x_m = 0]))).
void bnpc_legacy_beta_row(bnpc_mt19937 *rng, bnpc_legacy_gauss *g, int64_t M,
                          const unsigned long long *row, double p, double q,
                          double tmin, double tmax, float *theta)
{
    Words w(rng);
    for (int64_t m = 0; m < M; m++) {
        const unsigned long long one = (row[2 * (m >> 6)] >> (m & 63)) & 1ull;
        const unsigned long long zero =
            (row[2 * (m >> 6) + 1] >> (m & 63)) & 1ull;
        double v = lg_beta(w, g, p + (double)one, q + (double)zero);
        v = v < tmin ? tmin : (v > tmax ? tmax : v);
        theta[m] = (float)v;
    }
}

extern "C" int bnpc_mt_beta(bnpc_mt19937 *rng, bnpc_legacy_gauss *g, int64_t n,
                            const double *a, const double *b, double *out)
{
    if (!rng || !g || n < 0 || (n > 0 && (!a || !b || !out))) {
        bnpc_set_error("bad argument: mt_beta");
        return 2;
    }
    for (int64_t i = 0; i < n; i++) {
        if (!(a[i] > 0.0) || !(b[i] > 0.0)) {       // NumPy: ValueError
            bnpc_set_error("bad argument: Beta shape %lld is not positive",
                           (long long)i);
            return 2;
        }
    }
    Words w(rng);
    for (int64_t i = 0; i < n; i++) out[i] = lg_beta(w, g, a[i], b[i]);
    return 0;
}

// One profile row: theta[m] = float32(clip(Beta(p + n1[m] * fkt,
// q + n0[m] * fkt), tmin, tmax)) - np.clip(np.random.beta(...), TMIN, TMAX)
// .astype(np.float32) of libs/CRP.py:183-188 from integer column counts.
extern "C" int bnpc_mt_beta_theta(bnpc_mt19937 *rng, bnpc_legacy_gauss *g,
                                  int64_t M, double p, double q,
                                  const int32_t *n1, const int32_t *n0,
                                  double fkt, double tmin, double tmax,
                                  float *theta)
{
    if (!rng || !g || M < 0 || (M > 0 && (!n1 || !n0 || !theta))
        || !(p > 0.0) || !(q > 0.0) || !(fkt >= 0.0)) {
        bnpc_set_error("bad argument: mt_beta_theta");
        return 2;
    }
    for (int64_t m = 0; m < M; m++) {
        if (n1[m] < 0 || n0[m] < 0) {
            bnpc_set_error("bad argument: negative count");
            return 2;
        }
    }
    Words w(rng);
    for (int64_t m = 0; m < M; m++) {
        double v = lg_beta(w, g, p + (double)n1[m] * fkt,
                           q + (double)n0[m] * fkt);
        v = v < tmin ? tmin : (v > tmax ? tmax : v);
        theta[m] = (float)v;
    }
    return 0;
}
