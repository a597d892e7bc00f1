#!/usr/bin/env python3
"""The first cells of config 5's FIRST sweep at full size from the CPU ORACLE
(missing #4; BASELINE.md section 3.4 allows "a bounded number of
cells of sweep 1"):

    python tests/golden/make_c5_first_cells.py [cells=96]    # ~5 min, ~6 GB

50000 x 5000, the reference's random initial state (K0 ~ 31 600 clusters,
libs/CRP.py:119-152), then the first `cells` iterations of
update_assignments_Gibbs (libs/CRP.py:254-288) exactly as oracle/gibbs.py walks
them: the cell, the cluster it drew (the lowest free id for a new one, whose
profile is drawn from the stream), the number of clusters after it, and a
uniform peeked off the stream at the end.  A cell is visited once per sweep, so
the device's WHOLE first sweep must leave these cells with these labels
(tests/test_gpu_parity.py) - tiles, wide hints, births and the stream included.
Stored in c5_first_cells.npz."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle import crp_numpy as O  # noqa: E402

SEED = 42
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 96
cfg = sys.argv[2] if len(sys.argv) > 2 else 'c5'
N, M, C, miss, learned = bench.CONFIGS[cfg]
data = bench.synth(0, N, M, C, miss)
t0 = time.time()
np.random.seed(SEED)
model = bench.make_model(O, O, data, learned)
model.init()
K0 = len(model.cells_per_cluster)
print(f'[{time.time() - t0:5.0f}s] initial state: K0 = {K0}', flush=True)
lpost_new = model.get_lpost_single_new_cluster()
perm = np.random.permutation(N)
cells, drawn, K_after, born = [], [], [], []
for cell in perm[:n_cells]:
    model._take_out(cell)
    ids = model._cluster_ids()
    lpost = np.append(model.get_lpost_single(cell, ids), lpost_new[cell])
    pick = np.random.choice(np.append(ids, -1),
        p=model._normalize_log_probs(lpost))
    is_new = pick == -1
    if is_new:
        pick = model.init_new_cluster(cell)
    model._put_in(cell, pick)
    cells.append(int(cell))
    drawn.append(int(pick))
    K_after.append(len(model.cells_per_cluster))
    born.append(bool(is_new))
    print(f'[{time.time() - t0:5.0f}s] cell {cell}: -> {pick}'
        f'{" (new)" if is_new else ""}, K = {K_after[-1]}', flush=True)
peek = np.random.random()
name = 'c5_first_cells.npz' if cfg == 'c5' else f'{cfg}_first_cells.npz'
np.savez_compressed(os.path.join(HERE, name), seed=SEED, K0=K0,
    cells=np.array(cells), drawn=np.array(drawn), K_after=np.array(K_after),
    born=np.array(born), peek=peek, numpy=np.__version__)
print(f'{time.time() - t0:.0f}s: {sum(born)} clusters opened among '
    f'{n_cells} cells')
