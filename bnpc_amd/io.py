"""Minimal input / reporting helpers the driver and the CLI need.

Restates, as far as the CLI requires, the reference's
/root/reference/libs/dpmmIO.py: ``load_data`` :27-98 (separator / header /
index sniffing, ``3`` or blank -> NaN, ``2`` -> 1, transpose by default),
``load_txt`` :101-112, ``show_MH_acceptance`` :343-348, run-time summary
:310-315, ``_get_mcmc_termination`` :157-169.  Plotting, tree colouring and
the simulation-folder conventions are out of scope.
"""
from datetime import timedelta
import os

import numpy as np

_CODES = (0.0, 1.0, 2.0, 3.0)


def _sniff_separator(first_line):
    tabs, spaces, commas = (first_line.count(c) for c in '\t ,')
    if tabs > spaces and tabs > commas:
        return '\t'
    if commas > spaces:
        return ','
    return ' '


def _is_code(token):
    """True if the token is one of the matrix codes 0|1|2|3."""
    try:
        return float(token) in _CODES
    except ValueError:
        return False


def _sniff_layout(head):
    """(separator, header_row, index_col) from the first lines of the file."""
    sep = _sniff_separator(head[0])
    header_row = any(not _is_code(tok) for tok in head[0].split(sep)
        if tok not in ('', ' '))
    body_probe = head[1:] if header_row else head
    index_col = any(not _is_code(ln.split(sep)[0]) for ln in body_probe
        if ln.split(sep)[0] not in ('', ' '))
    return sep, header_row, index_col


def load_codes_native(in_file, transpose=True):
    """The same matrix as `load_data`, as int8 codes 0 | 1 | 3 (missing),
    cells x mutations, scanned by libbnpc_hip.so's byte scanner
    (bnpc_parse_matrix).  No names."""
    import ctypes as C
    from bnpc_amd import _lib
    head = []
    with open(in_file, 'r') as f:
        for _ in range(5):
            ln = f.readline()
            if ln.strip():
                head.append(ln.strip())
    if not head:
        raise ValueError(f'Could not read data from file: {in_file}')
    sep, header_row, index_col = _sniff_layout(head)
    lib = _lib.load()
    rows, cols = C.c_int64(0), C.c_int64(0)
    path = os.fsencode(in_file)
    bsep = sep.encode()
    _lib.check(lib.bnpc_parse_matrix(path, bsep, int(header_row),
        int(index_col), None, C.byref(rows), C.byref(cols)), 'parse_matrix')
    codes = np.empty((rows.value, cols.value), dtype=np.int8)
    _lib.check(lib.bnpc_parse_matrix(path, bsep, int(header_row),
        int(index_col), _lib.ptr(codes, C.c_int8), C.byref(rows),
        C.byref(cols)), 'parse_matrix')
    if transpose:
        codes = np.ascontiguousarray(codes.T)
    return codes


def codes_to_data(codes):
    """int8 codes 0 | 1 | 3 -> the reference's float64 matrix with NaN."""
    data = codes.astype(np.float64)
    data[codes == 3] = np.nan
    return data


def data_to_codes(data):
    codes = np.where(np.isnan(data), 3, data).astype(np.int8)
    return codes


def load_data(in_file, transpose=True, get_names=False):
    """Read a 0|1|2|3 matrix; returns cells x mutations float64 with NaN.

    On disk the reference format is mutations x cells, hence the default
    transpose (dpmmIO.py:27-98; run_BnpC.py:50-53 passes ``-t`` as
    store_false).
    """
    with open(in_file, 'r') as f:
        lines = [ln.rstrip('\r\n') for ln in f]
    while lines and lines[-1].strip() == '':
        lines.pop()
    if not lines:
        raise ValueError(f'Could not read data from file: {in_file}')

    sep, header_row, index_col = _sniff_layout(
        [ln.strip() for ln in lines[:5]])

    col_names = None
    if header_row:
        col_names = lines[0].strip().split(sep)
        lines = lines[1:]

    rows, row_names = [], []
    for ln in lines:
        toks = ln.strip('\r\n').split(sep)
        if sep == ' ':
            toks = ln.strip().split(sep)
        if index_col:
            row_names.append(toks[0])
            toks = toks[1:]
        rows.append([np.nan if t.strip() == '' else float(t) for t in toks])
    width = max(len(r) for r in rows)
    mat = np.full((len(rows), width), np.nan)
    for i, r in enumerate(rows):
        mat[i, :len(r)] = r

    if col_names is not None:
        if index_col and len(col_names) == width + 1:
            col_names = col_names[1:]
        col_names = np.array(col_names[:width], dtype=object)
    else:
        col_names = np.arange(1 if index_col else 0,
            width + (1 if index_col else 0))
    row_names = np.array(row_names, dtype=object) if index_col \
        else np.arange(len(rows))

    if transpose:
        mat = mat.T
        row_names, col_names = col_names, row_names

    mat = np.ascontiguousarray(mat, dtype=np.float64)
    mat[mat == 3] = np.nan
    mat[mat == 2] = 1
    if get_names:
        return mat, (row_names, col_names)
    return mat


def load_txt(path):
    """Cluster assignment file: space separated ints, optionally in a
    tab-separated table with an ``Assignment`` column (dpmmIO.py:101-112)."""
    with open(path, 'r') as f:
        text = f.read()
    lines = [ln for ln in text.splitlines() if ln.strip()]
    if lines and 'Assignment' in lines[0].split('\t'):
        col = lines[0].split('\t').index('Assignment')
        text = lines[1].split('\t')[col]
    return [int(tok) for tok in text.split()]


def get_mcmc_termination(args):
    """(run_var, description) from CLI arguments (dpmmIO.py:157-169)."""
    if args.runtime > 0:
        span = timedelta(minutes=args.runtime)
        return (args.time[0] + span, args.time[0] + args.burn_in * span), \
            f'for {args.runtime} mins'
    if args.lugsail > 0:
        return (args.lugsail, 0), f'until PSRF < {args.lugsail:.4f}'
    return (args.steps, int(args.steps * args.burn_in)), \
        f'for {args.steps} steps'


def show_MH_acceptance(counter, name, tab_no=2):
    """dpmmIO.py:343-348"""
    try:
        rate = counter[0] / counter.sum()
    except (ZeroDivisionError, FloatingPointError):
        rate = np.nan
    print('\t\t{}:{}{:.2f}'.format(name, '\t' * tab_no, rate))


def show_MCMC_summary(start, end, results):
    """Run-time line of the reference (dpmmIO.py:310-315): total wall time
    over the number of recorded steps of the first chain."""
    total = sum(r['ML'].size for r in results)
    step_time = (end - start) / results[0]['ML'].size
    print(f'\nClustering time:\t{end - start}\t'
        f'({step_time.total_seconds():.2f} secs. per MCMC step)')
    print(f'Steps recorded (all chains):\t{total}')


def get_out_dir(args, prefix=''):
    """dpmmIO.py:172-192"""
    if args.output:
        if any(args.output.endswith(s) for s in ('.txt', '.gv', '.csv')):
            out_dir = os.path.dirname(args.output)
        else:
            out_dir = args.output
    else:
        stamp = f'BnpC_{args.time[0]:%Y%m%d_%H:%M:%S}{prefix}'
        base = os.path.join(os.path.dirname(args.input), stamp)
        out_dir, i = base, 1
        while os.path.exists(out_dir):
            out_dir = f'{base}_{i}'
            i += 1
    os.makedirs(out_dir, exist_ok=True)
    return out_dir
