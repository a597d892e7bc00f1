/*
 * TEST INFRASTRUCTURE - part of the CPU oracle, never of the product path.
 *
 * Strictly sequential, NaN-skipping float64 sums: the arithmetic of
 * bottleneck.nansum (Bottleneck==1.3.5, requirements.txt:1 of the reference),
 * which the reference uses for every likelihood reduction:
 *   libs/CRP.py:202,204,234 (per-cell / flat log-likelihood sums)
 *   libs/CRP.py:363,366     (per-mutation sums over a cell subset)
 *   libs/CRP_learning_errors.py:63
 * bottleneck is a third-party dependency absent from /root/reference and from
 * the GPU box; its published algorithm for nansum is "asum = 0; for each
 * element in index order: if (ai == ai) asum += ai".  Verified bit-for-bit
 * against bottleneck 1.3.2 through the golden vectors in tests/golden/.
 *
 * Built by oracle/Makefile into oracle/_build/liboracle_seqsum.so.
 */
#include <stddef.h>

double bnpc_oracle_nansum(const double *v, long n)
{
    double s = 0.0;
    for (long i = 0; i < n; i++) {
        double a = v[i];
        if (a == a) s += a;
    }
    return s;
}

/* v is (r, c) C-contiguous; out[i] = sum_j v[i, j] */
void bnpc_oracle_nansum_axis1(const double *v, long r, long c, double *out)
{
    for (long i = 0; i < r; i++) {
        const double *row = v + (size_t)i * c;
        double s = 0.0;
        for (long j = 0; j < c; j++) {
            double a = row[j];
            if (a == a) s += a;
        }
        out[i] = s;
    }
}

/* v is (r, c) C-contiguous; out[j] = sum_i v[i, j], rows added in order */
void bnpc_oracle_nansum_axis0(const double *v, long r, long c, double *out)
{
    for (long j = 0; j < c; j++) out[j] = 0.0;
    for (long i = 0; i < r; i++) {
        const double *row = v + (size_t)i * c;
        for (long j = 0; j < c; j++) {
            double a = row[j];
            if (a == a) out[j] += a;
        }
    }
}
