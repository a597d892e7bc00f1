#!/usr/bin/env python3
"""End-to-end CLI run at a large shape on one GPU box (dev tool): writes a
synthetic matrix in the reference's text format, runs run_BnpC.py with several
chains through the fork pool, all three estimators, and reports timings.
usage: e2e_cli_big.py N M C chains steps [estimators...]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402

N, M, C, chains, steps = (int(a) for a in sys.argv[1:6])
data = bench.synth(0, N, M, C, 0.2)
tmp = tempfile.mkdtemp(prefix='bnpc_e2e_')
path = os.path.join(tmp, 'data.csv')
t0 = time.time()
codes = np.where(np.isnan(data), 3, data).astype(np.int8).T     # muts x cells
np.savetxt(path, codes, fmt='%d', delimiter=' ')
print(f'wrote {path}: {os.path.getsize(path) / 1e6:.1f} MB in '
    f'{time.time() - t0:.1f}s', flush=True)
out = os.path.join(tmp, 'out')
t0 = time.time()
r = subprocess.run([sys.executable, os.path.join(ROOT, 'run_BnpC.py'), path,
    '-n', str(chains), '-s', str(steps), '--seed', '1', '-np', '-o', out,
    '-e', *(sys.argv[6:] or ['posterior', 'ML', 'MAP'])],
    capture_output=True, text=True)
print(r.stdout[-3000:])
print(r.stderr[-2000:])
print(f'CLI exit {r.returncode} in {time.time() - t0:.1f}s')
for f in sorted(os.listdir(out)):
    print(f, os.path.getsize(os.path.join(out, f)))
