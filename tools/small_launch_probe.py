#!/usr/bin/env python3
"""Per-kernel time of one small likelihood launch (dev tool; run under
rocprofv3 --kernel-trace --stats):  small_launch_probe.py K [N M reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from bnpc_amd import _lib  # noqa: E402

K = int(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
M = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
data = bench.synth(0, N, M, 10, 0.2)
ctx = _lib.Context(data)
rng = np.random.RandomState(1)
theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5).astype(np.float32)
for _ in range(reps):
    ctx.ll_theta(0, theta, 0.01, 0.2, fetch=False)
ctx.sync()
ctx.bench_ll(3)
print(f'K={K}: bench_ll {1e3 * ctx.bench_ll(20):.2f} us per launch')
ctx.close()
