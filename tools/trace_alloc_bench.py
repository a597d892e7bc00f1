"""First write of a large sample trace (bnpc_rows_copy_zero) against
NumPy's np.zeros + one write per page, by team size:
    python3 tools/trace_alloc_bench.py [slots K M]      (default: config 5's
    74 x 67 x 5000 float32 = 99 MB)
Every measurement allocates fresh memory (the previous array is freed)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bnpc_amd import _lib  # noqa: E402

shape = tuple(int(x) for x in sys.argv[1:4]) or (74, 67, 5000)
lib = _lib.load()


def numpy_way():
    a = np.zeros(shape, dtype=np.float32)
    a.reshape(-1).view(np.uint8)[::4096] = 0
    return a


def team_way(threads):
    a = np.empty(shape, dtype=np.float32)
    width = a.nbytes // shape[0]
    _lib.check(lib.bnpc_rows_copy_zero(_lib.ptr(a), width, None, 0, shape[0],
        0, width, threads))
    return a


def ms(fn, *args, reps=5):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        a = fn(*args)
        out.append(round(1e3 * (time.perf_counter() - t0), 2))
        del a
    return out


print('shape', shape, round(np.prod(shape) * 4 / 1e6, 1), 'MB; default team',
    _lib.threads_for(int(np.prod(shape)) * 4))
print('np.zeros + a write per page:', ms(numpy_way))
for threads in (1, 2, 4, 8, 16, 32, 64):
    print('team of', threads, ':', ms(team_way, threads))
print('np.zeros + a write per page:', ms(numpy_way))
