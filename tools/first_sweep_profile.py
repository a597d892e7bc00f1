#!/usr/bin/env python3
"""The first Gibbs sweep of a large config alone (from K0 ~ 0.63 N clusters),
for rocprofv3 --kernel-trace --memory-copy-trace --stats (dev tool).
usage: first_sweep_profile.py [config]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c5'
N, M, C, miss, learned = bench.CONFIGS[cfg]
data = bench.synth(0, N, M, C, miss)
np.random.seed(42)
model = bench.make_model(None, dev_learn, data, True)
model.init()
model._dev()
os.environ['BNPC_TIMING'] = '2'
t0 = time.perf_counter()
model.update_assignments_Gibbs()
print(f'first sweep: {time.perf_counter() - t0:.3f} s, '
    f'K -> {len(model.cells_per_cluster)}', flush=True)
