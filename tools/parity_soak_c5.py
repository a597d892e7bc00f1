#!/usr/bin/env python3
"""Config 5 at full size, steady state: the schedule of
tests/golden/c5_schedule.py (near-truth start, scheduled steps of the
sampler's own move schedule, a forced split, a forced merge) walked by the
device chain and by the CPU oracle with ANOTHER seed and more steps than the
committed fixture, on the same host.  ~45 CPU-minutes for the oracle.
usage: parity_soak_c5.py [seed] [scheduled steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import c5_schedule as S  # noqa: E402
import libs.CRP_learning_errors as dev  # noqa: E402
from oracle import crp_numpy as O  # noqa: E402

S.SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 7
scheduled = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, M, C, miss, learned = bench.CONFIGS['c5']
data = bench.synth(0, N, M, C, miss)
res = {}
for name, mod in (('gpu', dev), ('oracle', O)):
    t0 = time.time()
    res[name] = S.drive(mod, data, scheduled=scheduled)
    print(f'{name}: {scheduled} scheduled steps + forced split + forced '
        f'merge in {time.time() - t0:.1f}s; moves (type, cells, accepted): '
        f'{res[name]["moves"].tolist()}', flush=True)
g, o = res['gpu'], res['oracle']
same = (g['assignments'] == o['assignments']).all(axis=1)
first = -1 if same.all() else int(np.argmin(same))
rel = np.max(np.abs(g['ML'] / o['ML'] - 1))
print(f'c5 seed {S.SEED}: first diverging recorded state: {first} (of '
    f'{same.size}); moves equal: {np.array_equal(g["moves"], o["moves"])}; '
    f'parameter digests equal: {list(g["digest"]) == list(o["digest"])}; '
    f'stream position equal: {g["stream_check"] == o["stream_check"]}; '
    f'max rel ML diff: {rel:.2e}; K: {g["K"].tolist()}')
