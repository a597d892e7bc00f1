#!/usr/bin/env python3
"""A chain with MORE THAN 64 CLUSTERS walked by the CPU ORACLE (system python:
the stack the GPU box also runs), as a fixture of the driver-run GPU suite:

    python tests/golden/make_k150_trajectory.py      # ~1 min, 1 core

bench.py's `k150` workload: 2000 cells x 500 mutations, 150 true clusters,
20 % missing, learned errors, data seed 0, CLI-default moves; 20 steps from the
initial state (K0 ~ 1260 clusters: step 1 is the sweep from there, the chain
then settles around 150-170), stepped exactly as bench.py steps its chain
(Chain.step = do_step + update_results, burn-in = a third).  Stored: the
assignments after every step (int16), the ML / MAP / alpha / FN / FP traces
-> k150_trajectory.npz."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle import crp_numpy as O  # noqa: E402

STEPS, SEED, CFG = 20, 42, 'k150'


def walk(mod_fixed, mod_learn):
    """The chain of bench.py's rank 0 for this workload (also what the GPU
    test runs, on the device model)."""
    N, M, C, miss, learned = bench.CONFIGS[CFG]
    data = bench.synth(0, N, M, C, miss)
    np.random.seed(SEED)
    np.random.seed(np.random.randint(0, 2 ** 32 - 1, 1)[0])
    model = bench.make_model(mod_fixed, mod_learn, data, learned)
    model.init()
    chain = bench.new_chain(model, learned, STEPS, CFG)
    burn = int(STEPS * .33)
    for i in range(1, STEPS + 1):
        bench.step(chain, i, burn)
    return model, chain.results


if __name__ == '__main__':
    t0 = time.time()
    model, res = walk(O, O)
    print(f'{time.time() - t0:.0f}s; K per step',
        [len(np.unique(a)) for a in res['assignments']])
    np.savez_compressed(os.path.join(HERE, f'{CFG}_trajectory.npz'),
        assignments=res['assignments'].astype(np.int16),
        ML=res['ML'], MAP=res['MAP'], DP_alpha=res['DP_alpha'], FN=res['FN'],
        FP=res['FP'], numpy=np.__version__)
