#!/usr/bin/env python3
"""CLI start-up at a large shape: text matrix -> device context, by route.
  round-1 route : native scan -> int8 codes -> float64 matrix -> bnpc_create
  first run     : native scan -> bit planes (+ bit-plane file written)
  later runs    : bit-plane file memory-mapped -> bnpc_create_planes
usage: ingest_bench.py [N M]   (default 50000 5000: config 5, ~500 MB of text)"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from bnpc_amd import _lib, bitplanes, io as bio  # noqa: E402

N, M = (int(a) for a in sys.argv[1:3]) if len(sys.argv) > 2 else (50000, 5000)
data = bench.synth(0, N, M, 50, 0.2)
codes = np.where(np.isnan(data), 3, data).astype(np.uint8).T    # muts x cells
del data
tmp = tempfile.mkdtemp(prefix='bnpc_ingest_')
path = os.path.join(tmp, 'data.csv')
text = np.empty((codes.shape[0], 2 * codes.shape[1]), dtype=np.uint8)
text[:, 0::2] = codes + ord('0')
text[:, 1::2] = ord(' ')
text[:, -1] = ord('\n')
text.tofile(path)
del text, codes
print(f'{N} cells x {M} mutations, text file {os.path.getsize(path) / 1e6:.0f} MB')


def timed(label, fn):
    t0 = time.perf_counter()
    out = fn()
    print(f'  {label:58s} {time.perf_counter() - t0:7.2f} s', flush=True)
    return out


print('round-1 route (every run):')
c = timed('native scan -> int8 codes (transposed)',
    lambda: bio.load_codes_native(path))
d = timed('codes -> float64 matrix with NaN', lambda: bio.codes_to_data(c))
ctx = timed('bnpc_create (packs on the host, uploads)',
    lambda: _lib.Context(data=d))
ctx.close()
del c, d
print('first run with the bit-plane file:')
os.environ['BNPC_BITPLANE_CACHE'] = '1'
p = timed('native scan -> planes, file written',
    lambda: bitplanes.load_matrix(path))
ctx = timed('bnpc_create_planes', lambda: _lib.Context(data=p))
ctx.close()
side = path + bitplanes.SUFFIX
print(f'  bit-plane file: {os.path.getsize(side) / 1e6:.0f} MB')
print('later runs:')
p = timed('memory-map the bit-plane file', lambda: bitplanes.load_matrix(path))
ctx = timed('bnpc_create_planes (pages in, uploads)',
    lambda: _lib.Context(data=p))
ctx.close()
for f in os.listdir(tmp):
    os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
