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

// n x random_sample()
void mt_fill_double(bnpc_mt19937 *s, double *out, int64_t n)
{
    int64_t i = 0;
    while (i < n) {
        if (s->pos >= 624) {
            refill(s->key);
            s->pos = 0;
        }
        int64_t pairs = (624 - s->pos) / 2;
        if (pairs > n - i) pairs = n - i;
        if (pairs == 0) {               // a double straddles two blocks
            out[i++] = mt_double(s);
            continue;
        }
        doubles_from(s->key + s->pos, out + i, pairs);
        s->pos += (int32_t)(2 * pairs);
        i += pairs;
    }
}

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
            refill(s->key);
            s->pos = 0;
        }
        const int avail = 624 - s->pos;
        masked_from(s->key + s->pos, cand, avail, mask);
        int j = 0;
        // every candidate is written to the next free slot; the slot only
        // advances when it is accepted (no data-dependent branch)
        for (; j < avail && cnt < n; j++) {
            out[cnt] = (int32_t)cand[j];
            cnt += (cand[j] <= max);
        }
        s->pos += j;
    }
}
