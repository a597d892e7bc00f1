#!/usr/bin/env python3
"""Chains sharing ONE GPU: aggregate steps/s for 1, 2, 4, 8 chains run the way
the CLI runs them (bnpc_amd.mcmc.MCMC.run: one forked process per chain, all on
device 0, host threads divided among them).  A chain keeps the GPU busy for
under a tenth of its step, so the device is far from the limit; what the curve
shows is the host side of the sharding path under
contention.  Dev tool; usage: multichain_bench.py [config] [steps] [n ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import libs.CRP as dev_fixed  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402
from bnpc_amd import _lib  # noqa: E402
from bnpc_amd.mcmc import MCMC  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
counts = [int(a) for a in sys.argv[3:]] or [1, 2, 4, 8]
N, M, C, miss, learned = bench.CONFIGS[cfg]
data = bench.synth(0, N, M, C, miss)
os.environ['BNPC_NUM_DEVICES'] = '1'
print(f'{cfg}: {N} x {M}, {steps} steps per chain, all chains on one GPU; '
    f'host cores {len(os.sched_getaffinity(0))}', flush=True)
base = None
for n in counts:
    model = bench.make_model(dev_fixed, dev_learn, data, learned)
    mcmc = MCMC(model, sm_prob=.33, dpa_prob=.25,
        error_prob=.25 if learned else 0., sm_ratios=[.75, .25], sm_steps=3)
    t0 = time.perf_counter()
    mcmc.run((steps, steps // 3), 42, n=n, verbosity=0)
    wall = time.perf_counter() - t0
    res = mcmc.get_results()
    assert len(res) == n and all(r['ML'].size == steps + 1 for r in res)
    rate = n * steps / wall
    base = base or rate
    print(f'  {n} chain(s): wall {wall:6.2f} s (start-up and first sweep '
        f'included)  aggregate {rate:8.1f} steps/s  x{rate / base:4.2f}',
        flush=True)
    assert not _lib.gpu_touched()       # the parent stays fork-safe
