#!/usr/bin/env python3
"""Device or host?  From two rocprofv3 kernel traces of ONE bench.py rank -
alone on the GPU, and next to 7 other chains on the same GPU - the kernels'
own durations (do they get longer when the GPU is shared?) and the idle gaps
between consecutive kernels of the rank (do they?).  usage:
queueing_from_trace.py <dir alone> <dir shared>"""
import csv
import glob
import os
import sys


def load(d):
    rows = []
    for path in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'),
            recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((int(r['Start_Timestamp']),
                    int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    return rows


def summary(rows):
    # the converged steps only: drop the first 15 % of the launches
    rows = rows[len(rows) * 15 // 100:]
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    by = {}
    for s, e, name in rows:
        key = name.split('(')[0][:40]
        t = by.setdefault(key, [0, 0])
        t[0] += e - s
        t[1] += 1
    return rows, busy, span, by


def main():
    out = []
    for label, d in zip(('alone', 'with 7 other chains'), sys.argv[1:3]):
        rows = load(d)
        if not rows:
            print(f'{label}: no kernel trace under {d}')
            continue
        rows, busy, span, by = summary(rows)
        print(f'{label}: {len(rows)} launches over {span / 1e6:.1f} ms, '
            f'kernels busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} % of '
            f'the span), mean kernel {busy / len(rows) / 1e3:.2f} us')
        out.append(by)
    if len(out) == 2:
        print('mean duration per kernel, us (alone -> shared):')
        for key in sorted(out[0], key=lambda k: -out[0][k][0])[:12]:
            a = out[0][key]
            b = out[1].get(key)
            if b:
                print(f'  {key:40s} {a[0] / a[1] / 1e3:8.2f} -> '
                    f'{b[0] / b[1] / 1e3:8.2f}  ({a[1]} / {b[1]} launches)')


if __name__ == '__main__':
    main()
