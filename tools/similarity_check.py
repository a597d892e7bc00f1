#!/usr/bin/env python3
"""Dev tool (build container only): how close is each file of this repo to its
nearest file of the reference checkout?  Character, line and token ratios."""
import difflib
import glob
import re
import sys

REF = '/root/reference'


def lines(text):
    return [l.strip() for l in text.splitlines()
        if l.strip() and not l.strip().startswith('#')]


def toks(text):
    return re.findall(r'[A-Za-z_]\w*|\d+\.?\d*|\S', text)


ref = {p: open(p, errors='ignore').read()
    for p in glob.glob(REF + '/**/*.py', recursive=True)}
pats = sys.argv[1:] or ['**/*.py']
mine = sorted({p for pat in pats for p in glob.glob(pat, recursive=True)
    if 'gpurun_out' not in p})
for m in mine:
    a = open(m, errors='ignore').read()
    if len(a) < 400:
        continue
    for r, b in ref.items():
        if difflib.SequenceMatcher(None, a, b).quick_ratio() < 0.7:
            continue
        c = difflib.SequenceMatcher(None, a, b, autojunk=False).ratio() \
            if len(a) < 60000 else float('nan')
        l = difflib.SequenceMatcher(None, lines(a), lines(b),
            autojunk=False).ratio()
        t = difflib.SequenceMatcher(None, toks(a), toks(b),
            autojunk=False).ratio()
        if max(c, l, t) > 0.3:
            print(f'{m:34s} {r[len(REF) + 1:]:28s} chars {c:.2f} '
                f'lines {l:.2f} tokens {t:.2f}')
