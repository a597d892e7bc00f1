#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc passes.
usage: pmc_collect.py out.json dir_or_csv [dir_or_csv ...]
Every *counter_collection.csv below the given paths is read; the result maps
kernel name -> counter -> {launches, mean, max}.  FETCH_SIZE / WRITE_SIZE are
in KiB as rocprofv3 reports them (MI355X_MICROARCH.md, HBM section; the fetch
counter is calibrated on the scalar-load streams of the likelihood kernel in
profiles/r01/fetch_calibration.md)."""
import csv
import glob
import json
import os
import sys

out, paths = sys.argv[1], sys.argv[2:]
files = []
for p in paths:
    if os.path.isdir(p):
        files += glob.glob(os.path.join(p, '**', '*counter_collection.csv'),
            recursive=True)
    else:
        files.append(p)
acc = {}
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row['Kernel_Name']
            ctr = row['Counter_Name']
            val = float(row['Counter_Value'])
            acc.setdefault(name, {}).setdefault(ctr, []).append(val)
summary = {k: {c: {'launches': len(v), 'mean': sum(v) / len(v), 'max': max(v)}
    for c, v in sorted(ctrs.items())} for k, ctrs in sorted(acc.items())}
# which build the counters belong to (bench.py refuses another one's)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bnpc_amd import build as _build  # noqa: E402
summary['_meta'] = {'source_digest': _build.source_digest(),
    'command': os.environ.get('PMC_COMMAND', 'rocprofv3 --pmc <counter> -- '
        'python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 (one pass per '
        'counter set)')}
with open(out, 'w') as fh:
    json.dump(summary, fh, indent=1)
print(f'{len(files)} file(s), {len(summary)} kernels -> {out}')
