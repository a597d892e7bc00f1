#!/usr/bin/env python3
"""End-to-end CLI run at a large shape on one GPU box (dev tool): writes a
synthetic matrix in the reference's text format, runs run_BnpC.py with several
chains through the fork pool, all three estimators, and reports timings.
usage: [E2E_FLAGS="-smp 0.5 -sms 5"] e2e_cli_big.py N M C chains steps
       [estimators...]   (E2E_FLAGS: further CLI flags, e.g. config 5's moves)"""
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
codes = np.where(np.isnan(data), 3, data).astype(np.uint8).T    # muts x cells
text = np.empty((codes.shape[0], 2 * codes.shape[1]), dtype=np.uint8)
text[:, 0::2] = codes + ord('0')
text[:, 1::2] = ord(' ')
text[:, -1] = ord('\n')
text.tofile(path)
del text
print(f'wrote {path}: {os.path.getsize(path) / 1e6:.1f} MB in '
    f'{time.time() - t0:.1f}s', flush=True)
# twice: the first run scans the text and writes the bit-plane file next to
# it, the second one memory-maps that file
for attempt in ('first run (text scanned, bit-plane file written)',
        'second run (bit-plane file memory-mapped)'):
    out = os.path.join(tmp, 'out')
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run_BnpC.py'),
        path, '-n', str(chains), '-s', str(steps), '--seed', '1', '-np', '-o',
        out, *os.environ.get('E2E_FLAGS', '').split(), '-e',
        *(sys.argv[6:] or ['posterior', 'ML', 'MAP'])],
        capture_output=True, text=True)
    print(f'== {attempt}')
    print(r.stdout[-1500:])
    print(r.stderr[-1500:])
    print(f'CLI exit {r.returncode} in {time.time() - t0:.1f}s')
    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)))
print(sorted(os.listdir(tmp)))
