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
