#!/usr/bin/env python3
"""ns per cell of the native Gibbs loop (bnpc_gibbs_sweep) with hint records
in visiting order, as the native step runs it, on a synthetic converged
matrix: K separated clusters, a fraction of the cells torn between two columns
(a rival within two nats) and a fraction among three.  Host only - the
records come from _lib.hints_from_matrix.
usage: hinted_loop_bench.py [N] [K] [reps] [pairs] [triples]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bnpc_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 14
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
pairs = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
triples = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
rng = np.random.RandomState(0)
ld = K + 4
truth = rng.randint(0, K, N)
ll = -900.0 - 100 * rng.uniform(size=(N, ld))
rows = np.arange(N)
ll[rows, truth] = -300.0 - 10 * rng.uniform(size=N)
kind = rng.uniform(size=N)
two = kind < pairs + triples
three = kind < triples
rival = (truth + 1 + rng.randint(0, K - 1, N)) % K
ll[two, rival[two]] = ll[two, truth[two]] - rng.uniform(0, 2, two.sum())
third = (truth + K - 1 - (rival == (truth + K - 1) % K)) % K
third[third == truth] = (truth[third == truth] + 2) % K
ok3 = three & (third != rival) & (third != truth)
ll[ok3, third[ok3]] = ll[ok3, truth[ok3]] - rng.uniform(0, 3, ok3.sum())
post_new = np.full(N, -700.0)
crp = np.append(0, np.log(np.append(np.arange(1, N + 1), 70.0))
    - np.log(N - 1 + 70.0))
lib = _lib.load()
i64, f64 = C.c_int64, C.c_double
np.random.seed(1)
times = []
for r in range(reps):
    perm = _lib.as_i64(np.random.permutation(N))
    assignment = truth.astype(np.int64).copy()
    sizes = np.bincount(truth, minlength=K).astype(np.int64)
    col_of_id = np.full(N, -1, dtype=np.int64)
    col_of_id[:K] = np.arange(K)
    col_id = np.full(ld, -1, dtype=np.int64)
    col_id[:K] = np.arange(K)
    col_size = np.zeros(ld, dtype=np.int64)
    col_size[:K] = sizes
    order = np.zeros(ld, dtype=np.int64)
    order[:K] = np.arange(K)
    scratch = np.empty(2 * (ld + 1))
    col_prior = np.ascontiguousarray(crp[sizes])
    if r == 0:
        by_cell = _lib.hints_from_matrix(ll[:, :K], col_prior)
    hint = np.ascontiguousarray(by_cell[perm])
    st = _lib.GibbsState(N, ld, K, K, 0, -1, N, -1)
    st.hint = hint.ctypes.data
    st.hint_prior = col_prior.ctypes.data
    st.hint_cols = K
    st.hint_in_order = 1
    t0 = time.perf_counter()
    with _lib.NumpyStream() as s:
        _lib.check(lib.bnpc_gibbs_sweep(C.byref(st), s, _lib.ptr(perm, i64),
            _lib.ptr(ll, f64), _lib.ptr(post_new, f64), _lib.ptr(crp, f64),
            _lib.ptr(assignment, i64), _lib.ptr(col_of_id, i64),
            _lib.ptr(col_id, i64), _lib.ptr(col_size, i64),
            _lib.ptr(order, i64), _lib.ptr(scratch, f64)), 'sweep')
    times.append(time.perf_counter() - t0)
times = np.sort(times)
print(f'N={N} K={K} pairs={pairs} triples={triples}: min '
    f'{1e9 * times[0] / N:.1f} median {1e9 * times[reps // 2] / N:.1f} '
    f'ns/cell ({1e6 * times[reps // 2]:.1f} us/sweep); from the records '
    f'{st.hint_used}, between two {st.pair_used}, among three '
    f'{st.triple_used}, in the lane {st.lane_used} (in runs {st.stride_used}), moved '
    f'{int((assignment != truth).sum())}')
