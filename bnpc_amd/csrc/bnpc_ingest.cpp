// bnpc_ingest.cpp - native reader for the reference's text matrix format
// (dpmmIO.load_data parses 250 M entries at
// config 5 through pandas; this is a two-pass byte scanner).
//
// Semantics restated from /root/reference/libs/dpmmIO.py:27-98: entries are
// 0 | 1 | 2 | 3 (possibly written as floats), separated by ONE separator
// character (space, tab or comma); an empty field or 3 is "missing"; 2
// (homozygous) counts as 1.  Separator / header-row / index-column sniffing
// stays in Python (bnpc_amd/io.py); this file only scans the body.
// Output codes: 0, 1, 3 (missing) as int8, row-major in FILE orientation.

#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "bnpc_hip.h"
#include "bnpc_internal.h"

static int read_all(const char *path, std::vector<char> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f) {
        bnpc_set_error("cannot open %s: %s", path, strerror(errno));
        return 1;
    }
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize((size_t)size + 1);
    const size_t got = fread(buf.data(), 1, (size_t)size, f);
    fclose(f);
    if (got != (size_t)size) {
        bnpc_set_error("short read on %s", path);
        return 1;
    }
    buf[size] = '\n';       // sentinel: every line ends with a newline
    return 0;
}

static inline int token_code(const char *b, const char *e, int *code)
{
    while (b < e && (*b == ' ' || *b == '\r')) b++;
    while (e > b && (e[-1] == ' ' || e[-1] == '\r')) e--;
    if (b == e) {
        *code = 3;
        return 0;
    }
    if (e - b == 1 && *b >= '0' && *b <= '3') {
        *code = *b - '0';
    } else {
        char tmp[64];
        const size_t n = (size_t)(e - b);
        if (n >= sizeof(tmp)) return 1;
        memcpy(tmp, b, n);
        tmp[n] = 0;
        char *end = nullptr;
        const double v = strtod(tmp, &end);
        if (end == tmp || *end != 0) return 1;
        if (v == 0.0) *code = 0;
        else if (v == 1.0) *code = 1;
        else if (v == 2.0) *code = 2;
        else if (v == 3.0) *code = 3;
        else return 1;
    }
    if (*code == 2) *code = 1;
    return 0;
}

// rows/cols of the body (after skipping `skip_rows` lines and, if
// skip_index, the first field of every line); out == NULL: count only.
extern "C" int bnpc_parse_matrix(const char *path, char sep, int skip_rows,
                                 int skip_index, int8_t *out,
                                 int64_t *rows, int64_t *cols)
{
    if (!path || !rows || !cols) {
        bnpc_set_error("bad argument: NULL");
        return 2;
    }
    std::vector<char> buf;
    if (read_all(path, buf)) return 1;
    const char *p = buf.data();
    const char *end = p + buf.size();
    // drop trailing blank lines
    const char *last = end - 1;
    while (last > p && (last[-1] == '\n' || last[-1] == '\r' ||
                        last[-1] == ' ')) last--;
    for (int s = 0; s < skip_rows && p < last; s++) {
        while (p < last && *p != '\n') p++;
        p++;
    }
    int64_t r = 0, width = 0;
    const int64_t want_cols = out ? *cols : 0;
    while (p < last) {
        const char *eol = p;
        while (*eol != '\n') eol++;
        const char *le = eol;
        if (le > p && le[-1] == '\r') le--;
        if (sep == ' ') {           // the Python reader strips the line
            while (p < le && *p == ' ') p++;
            while (le > p && le[-1] == ' ') le--;
        }
        int64_t c = 0;
        const char *tok = p;
        bool first = true;
        for (const char *q = p;; q++) {
            if (q == le || *q == sep) {
                if (!(first && skip_index)) {
                    if (out) {
                        int code;
                        if (token_code(tok, q, &code)) {
                            bnpc_set_error("row %lld, field %lld is not "
                                           "0|1|2|3 or empty",
                                           (long long)r, (long long)c);
                            return 3;
                        }
                        if (c < want_cols) out[r * want_cols + c] = (int8_t)code;
                    }
                    c++;
                }
                first = false;
                tok = q + 1;
                if (q == le) break;
            }
        }
        if (out)
            for (int64_t j = c; j < want_cols; j++)
                out[r * want_cols + j] = 3;      // ragged row: missing
        if (c > width) width = c;
        r++;
        p = eol + 1;
    }
    if (out && (r != *rows || width > want_cols)) {
        bnpc_set_error("matrix shape changed between passes");
        return 3;
    }
    *rows = r;
    *cols = width;
    return 0;
}

// ---- bit planes <-> codes (host) -------------------------------------------
// The packed form of the data matrix, on disk (bnpc_amd/bitplanes.py) and in
// HBM alike: per cell W = ceil(M / 64) pairs of 64-bit words {ones, zeros},
// bit b of word w = mutation 64 w + b; an entry with neither bit is missing.
extern "C" int bnpc_pack_codes(const int8_t *codes, int64_t N, int64_t M,
                               int64_t row_stride, int64_t col_stride,
                               uint64_t *planes)
{
    if (!codes || !planes || N < 0 || M < 1) {
        bnpc_set_error("bad argument: pack_codes");
        return 2;
    }
    const int64_t W = (M + 63) / 64;
    if (row_stride == 1 && col_stride != 1) {
        // a transposed view (the file is mutations x cells): walk it in
        // 64 x 64 tiles along ITS rows, so that every cache line of the
        // source is used whole
        for (int64_t i0 = 0; i0 < N; i0 += 64) {
            const int64_t ni = i0 + 64 < N ? 64 : N - i0;
            for (int64_t w = 0; w < W; w++) {
                uint64_t o[64] = {0}, z[64] = {0};
                const int64_t m0 = w * 64, m1 = m0 + 64 < M ? m0 + 64 : M;
                for (int64_t m = m0; m < m1; m++) {
                    const int8_t *src = codes + m * col_stride + i0;
                    const uint64_t bit = 1ull << (m - m0);
                    for (int64_t j = 0; j < ni; j++) {
                        const int v = src[j];
                        if (v == 1 || v == 2) o[j] |= bit;
                        else if (v == 0) z[j] |= bit;
                        else if (v != 3) {
                            bnpc_set_error(
                                "codes[%lld,%lld] = %d is not 0|1|2|3",
                                (long long)(i0 + j), (long long)m, v);
                            return 2;
                        }
                    }
                }
                for (int64_t j = 0; j < ni; j++) {
                    uint64_t *out = planes + ((size_t)(i0 + j) * W + w) * 2;
                    out[0] = o[j];
                    out[1] = z[j];
                }
            }
        }
        return 0;
    }
    for (int64_t i = 0; i < N; i++) {
        const int8_t *row = codes + i * row_stride;
        uint64_t *out = planes + (size_t)i * W * 2;
        for (int64_t w = 0; w < W; w++) {
            uint64_t o = 0, z = 0;
            const int64_t m0 = w * 64, m1 = m0 + 64 < M ? m0 + 64 : M;
            for (int64_t m = m0; m < m1; m++) {
                const int v = row[m * col_stride];
                if (v == 1 || v == 2) o |= 1ull << (m - m0);    // 2 -> 1
                else if (v == 0) z |= 1ull << (m - m0);
                else if (v != 3) {
                    bnpc_set_error("codes[%lld,%lld] = %d is not 0|1|2|3",
                                   (long long)i, (long long)m, v);
                    return 2;
                }
            }
            out[2 * w] = o;
            out[2 * w + 1] = z;
        }
    }
    return 0;
}

// rows `cells[0..n)` of the planes as int8 codes 0 | 1 | 3 (n x M)
extern "C" int bnpc_unpack_codes(const uint64_t *planes, int64_t N, int64_t M,
                                 const int64_t *cells, int64_t n, int8_t *codes)
{
    if (!planes || !codes || (n > 0 && !cells && n != N)) {
        bnpc_set_error("bad argument: unpack_codes");
        return 2;
    }
    const int64_t W = (M + 63) / 64;
    for (int64_t r = 0; r < n; r++) {
        const int64_t i = cells ? cells[r] : r;
        if (i < 0 || i >= N) {
            bnpc_set_error("cell index out of range");
            return 2;
        }
        const uint64_t *in = planes + (size_t)i * W * 2;
        int8_t *out = codes + (size_t)r * M;
        for (int64_t m = 0; m < M; m++) {
            const uint64_t bit = 1ull << (m & 63);
            const uint64_t o = in[2 * (m >> 6)], z = in[2 * (m >> 6) + 1];
            out[m] = (o & bit) ? 1 : ((z & bit) ? 0 : 3);
        }
    }
    return 0;
}
