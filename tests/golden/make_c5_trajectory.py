#!/usr/bin/env python3
"""Config 5's steady-state trajectory at FULL size from the CPU ORACLE:

    python tests/golden/make_c5_trajectory.py      # ~40 min on 1 core, ~25 GB

walks tests/golden/c5_schedule.py (near-truth start, 6 scheduled steps, a
forced split, a forced merge) with oracle.crp_numpy on the system Python (the
stack the GPU box runs) and stores assignments (int16), traces, parameter
digests, a sample of the last parameter rows and the restricted-Gibbs log in
c5_trajectory.npz."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import c5_schedule as S  # noqa: E402
import bench  # noqa: E402
from oracle import crp_numpy as O  # noqa: E402

N, M, C, miss, learned = bench.CONFIGS['c5']
data = bench.synth(0, N, M, C, miss)
t0 = time.time()


def progress(tag, trace, moves):
    print(f'[{time.time() - t0:7.0f}s] {tag}: K={trace["K"][-1]} '
        f'ML={trace["ML"][-1]:.4f} moves={moves}', flush=True)


res = S.drive(O, data, progress=progress)
sample = np.random.RandomState(3).choice(res['last_params'].size, 4096,
    replace=False)
np.savez_compressed(os.path.join(HERE, 'c5_trajectory.npz'),
    assignments=res['assignments'].astype(np.int16),
    ML=res['ML'], MAP=res['MAP'], DP_alpha=res['DP_alpha'], FN=res['FN'],
    FP=res['FP'], K=res['K'], digest=res['digest'], moves=res['moves'],
    last_params_sample=res['last_params'].ravel()[sample],
    last_params_index=sample, stream_check=res['stream_check'],
    numpy=np.__version__)
print(f'{time.time() - t0:.0f}s; moves (type, cells, accepted):\n',
    res['moves'])
