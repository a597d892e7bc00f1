#!/usr/bin/env python3
"""Phase times of the screened parameter batches in a running config-3 chain
(BNPC_TIMING=mh): draws + staging / device screen / exact host arithmetic."""
import os
import sys
os.environ['BNPC_TIMING'] = 'mh'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import libs.CRP as dev_fixed  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N, M, C, miss, learned = bench.CONFIGS[cfg]
data = bench.synth(0, N, M, C, miss)
np.random.seed(42)
model = bench.make_model(dev_fixed, dev_learn, data, learned)
model.init()
chain = bench.new_chain(model, learned, steps + 5, cfg)
for i in range(1, steps + 6):
    if i == 6:
        print('---- converged steps ----', file=sys.stderr, flush=True)
    bench.step(chain, i, 0)
