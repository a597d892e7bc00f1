#!/usr/bin/env python3
"""Calibrate rocprofv3's FETCH_SIZE on THIS kernel's access pattern (scalar
s_load_dwordx4 / x16 streams): one cluster group (K = 8, KW = 8), so every
mask row is read exactly once: known bytes = nblk * Mpad * 16 (+ the 640 KB
table once per XCD).  Run under `rocprofv3 --pmc FETCH_SIZE`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bnpc_amd import _lib  # noqa: E402
import bench  # noqa: E402

os.environ['BNPC_KW'] = '8'
N, M = 50000, 5000
data = bench.synth(0, N, M, 50, 0.2)
theta = np.clip(np.random.RandomState(1).uniform(size=(8, M)), 1e-5,
    1 - 1e-5).astype(np.float32)
ctx = _lib.Context(data=data)
for _ in range(3):
    ctx.ll_theta(0, theta, 0.01, 0.2, fetch=False)
    ctx.sync()
nblk = (N + 63) // 64
Mpad = (M + 63) // 64 * 64
print('known mask bytes per launch', nblk * Mpad * 16,
    'table bytes', (M + 7) // 8 * 8 * 128, 'out bytes', N * 8 * 8)
