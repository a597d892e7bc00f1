#!/usr/bin/env python3
"""ns per cell of the native Gibbs loop (bnpc_gibbs_sweep) on a synthetic
converged-regime matrix: K well separated clusters, N cells.  Host only."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bnpc_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rng = np.random.RandomState(0)
ld = K + 4
truth = rng.randint(0, K, N)
ll = -900.0 - 100 * rng.uniform(size=(N, ld))
ll[np.arange(N), truth] = -300.0 - 10 * rng.uniform(size=N)
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
    st = _lib.GibbsState(N, ld, K, K, 0, -1, N, -1)
    t0 = time.perf_counter()
    with _lib.NumpyStream() as s:
        _lib.check(lib.bnpc_gibbs_sweep(C.byref(st), s, _lib.ptr(perm, i64),
            _lib.ptr(ll, f64), _lib.ptr(post_new, f64), _lib.ptr(crp, f64),
            _lib.ptr(assignment, i64), _lib.ptr(col_of_id, i64),
            _lib.ptr(col_id, i64), _lib.ptr(col_size, i64),
            _lib.ptr(order, i64), _lib.ptr(scratch, f64)), 'sweep')
    times.append(time.perf_counter() - t0)
    assert K > 64 or (st.new_cell < 0 and np.array_equal(assignment, truth))
times = np.sort(times)
print(f'N={N} K={K}: min {1e9 * times[0] / N:.1f} ns/cell, median '
    f'{1e9 * times[reps // 2] / N:.1f} ns/cell '
    f'({1e3 * times[reps // 2]:.3f} ms/sweep)')
