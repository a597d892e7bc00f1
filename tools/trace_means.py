#!/usr/bin/env python3
"""Means of the [gibbs] / [params] phase traces (BNPC_TIMING=gibbs,params on
stderr) of a bench run next to the line's own phase clocks.
usage: trace_means.py <trace log> [bench json]"""
import json
import re
import sys

import numpy as np

rows = {}
heads = {}
for line in open(sys.argv[1]):
    m = re.match(r'\[(gibbs|params)\] ([^:]*): (.*) us', line)
    if not m:
        continue
    kind, body = m.group(1), m.group(3)
    parts = re.findall(r'([^,0-9][^,]*?) (-?[0-9.]+)(?:,|$)', body)
    heads[kind] = [p[0].strip() for p in parts]
    rows.setdefault(kind, []).append([float(p[1]) for p in parts])
for kind, r in rows.items():
    width = min(len(x) for x in r)
    a = np.array([x[:width] for x in r[10:]])
    print(f'[{kind}] {len(a)} calls after the first 10; mean us per call:')
    for h, v in zip(heads[kind], a.mean(axis=0)):
        print(f'    {h:40s} {v:8.1f}')
    print(f'    {"sum":40s} {a.sum(axis=1).mean():8.1f}')
if len(sys.argv) > 2:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    w = j['window']
    print('bench clocks (ms per call):', {k: w[k]['ms_per_call'] for k in
        ('gibbs', 'parameters', 'record', 'gibbs_waits_for_device') if k in w},
        'value', j['value'])
