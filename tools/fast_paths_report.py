#!/usr/bin/env python3
"""Which fast paths come up on THIS Python stack (no GPU needed): the private
interfaces and start-up comparisons the ~2000 steps/s of config 3 rest on
(bnpc_amd.model.fast_paths; README "requirements").  With all of them False the
same chain runs at ~350 steps/s (profiles/r04/bench_fallbacks.json).
usage: [LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6] python tools/fast_paths_report.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import scipy  # noqa: E402

try:
    from bnpc_amd import model as P
except OSError as exc:      # the .so needs a newer libstdc++ than conda ships
    sys.exit(f'cannot load libbnpc_hip.so: {exc}\nrun with LD_PRELOAD='
        '/usr/lib/x86_64-linux-gnu/libstdc++.so.6')
print(f'python {sys.version.split()[0]}, numpy {np.__version__}, '
    f'scipy {scipy.__version__}')
paths = P.fast_paths()
for key, val in paths.items():
    print(f'  {key:13s} {val}')
print('all fast paths' if all(paths.values())
    else 'fallbacks in use: ' + ', '.join(k for k, v in paths.items() if not v))
