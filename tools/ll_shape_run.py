#!/usr/bin/env python3
"""The cells x clusters x mutations evaluation at ONE shape, `reps` times (for
rocprofv3 --pmc / --kernel-trace passes on a shape of its own: the bench run
mixes the converged sweeps' launches with the small ones of the moves).
usage: ll_shape_run.py N M K [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bnpc_amd import _lib  # noqa: E402
import bench  # noqa: E402

N, M, K = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
data = bench.synth(0, N, M, 10, 0.2)
ctx = _lib.Context(data=data)
theta = np.clip(np.random.RandomState(1).uniform(size=(K, M)), 1e-5,
    1 - 1e-5).astype(np.float32)
for _ in range(reps):
    ctx.ll_theta(0, theta, 0.01, 0.2, fetch=False)
    ctx.sync()
print(ctx.last_launch())
ctx.close()
