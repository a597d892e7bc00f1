"""CLI surface (flags and defaults of the reference's run_BnpC.py:13-196) and
the data loader (libs/dpmmIO.py:27-112) - CPU only."""
import os

import numpy as np
import pytest

import run_BnpC
from bnpc_amd import io as bio, postproc

# flag -> (dest, default) as in the reference, incl. its quirks
REFERENCE_DEFAULTS = {
    'transpose': True, 'debug': False, 'falseNegative': -1,
    'falsePositive': -1, 'falseNegative_mean': 0.2, 'falseNegative_std': 0.1,
    'falsePositive_mean': 0.01, 'falsePositive_std': 0.01,
    'DPa_prior': [-1, -1], 'param_prior': [.25, .25], 'fixed_assignment': '',
    'chains': 1, 'steps': 5000, 'runtime': -1, 'lugsail': -1, 'burn_in': 0.33,
    'conc_update_prob': 0.25, 'error_update_prob': 0.25,
    'split_merge_prob': 0.33, 'split_merge_steps': 3,
    'split_merge_ratios': [0.75, 0.25], 'estimator': 'posterior',
    'single_chains': False, 'seed': -1, 'output': '', 'verbosity': 1,
    'no_plots': False, 'tree': '', 'true_clusters': '', 'true_data': '',
}


def test_cli_defaults_match_reference():
    args = run_BnpC.parse_args(['data.csv'])
    got = vars(args)
    assert got.pop('input') == 'data.csv'
    assert got == REFERENCE_DEFAULTS


def test_cli_reference_command_lines_parse():
    a = run_BnpC.parse_args('example_data/data.csv -n 1 -s 200 --seed 42 -np'
        .split())
    assert (a.chains, a.steps, a.seed, a.no_plots) == (1, 200, 42, True)
    a = run_BnpC.parse_args('d.csv -t -FP 0.001 -FN 0.1 -n 8 -smp 0.5 -sms 5 '
        '-smr 0.6 0.4 -ap 2 1 -pp 1 1 -e ML MAP -ls 1.05 -b 0.5 -v 2'.split())
    assert a.transpose is False and a.falsePositive == 0.001
    assert a.split_merge_ratios == [0.6, 0.4] and a.estimator == ['ML', 'MAP']
    assert a.lugsail == 1.05 and a.DPa_prior == [2.0, 1.0]
    for bad in ('-FN_m 0', '-FN_m 1', '-b 1.5', '-ls 2', '-v 3'):
        with pytest.raises(SystemExit):
            run_BnpC.parse_args(['d.csv'] + bad.split())
    with pytest.raises(SystemExit):
        run_BnpC.parse_args(['--version'])


def test_load_example_data(golden_dir):
    data, names = bio.load_data(os.path.join(golden_dir, 'example_data.csv'),
        get_names=True)
    assert data.shape == (100, 100)
    assert np.isnan(data).sum() == 948
    assert (data == 1).sum() == 2445 and (data == 0).sum() == 6607
    raw = bio.load_data(os.path.join(golden_dir, 'example_data.csv'),
        transpose=False)
    assert np.array_equal(np.isnan(raw.T), np.isnan(data))
    assert np.array_equal(np.nan_to_num(raw.T, nan=3),
        np.nan_to_num(data, nan=3))


@pytest.mark.parametrize('sep', [' ', '\t', ','])
def test_load_data_formats(tmp_path, sep):
    mat = np.array([[0, 1, 3, 2], [1, 1, 0, 3], [3, 0, 0, 1]])
    want = mat.astype(float)
    want[mat == 3] = np.nan
    want[mat == 2] = 1
    plain = tmp_path / 'plain.txt'
    plain.write_text('\n'.join(sep.join(str(v) for v in r) for r in mat))
    got = bio.load_data(str(plain), transpose=False)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.nan_to_num(got, nan=9), np.nan_to_num(want, nan=9))
    assert bio.load_data(str(plain)).shape == (4, 3)

    named = tmp_path / 'named.txt'
    lines = [sep.join(['id'] + [f'c{j}' for j in range(4)])]
    lines += [sep.join([f'm{i}'] + [str(v) for v in r])
        for i, r in enumerate(mat)]
    named.write_text('\n'.join(lines))
    got, (rows, cols) = bio.load_data(str(named), transpose=False,
        get_names=True)
    assert np.array_equal(np.nan_to_num(got, nan=9), np.nan_to_num(want, nan=9))
    assert list(rows) == ['m0', 'm1', 'm2'] and list(cols)[-1] == 'c3'


def test_load_txt(tmp_path):
    f = tmp_path / 'a.txt'
    f.write_text('0 0 1 2 1')
    assert bio.load_txt(str(f)) == [0, 0, 1, 2, 1]
    g = tmp_path / 'b.txt'
    g.write_text('chain\testimator\tAssignment\nmean\tMAP\t3 3 1\n')
    assert bio.load_txt(str(g)) == [3, 3, 1]


def test_lugsail_psrf():
    rng = np.random.RandomState(0)
    same = [(rng.normal(size=400), 100) for _ in range(3)]
    assert 0.95 < postproc.get_lugsail_batch_means_est(same) < 1.1
    sticky = [(np.repeat(rng.normal(size=40), 10), 100) for _ in range(3)]
    assert postproc.get_lugsail_batch_means_est(sticky) > \
        postproc.get_lugsail_batch_means_est(same)
    assert postproc.get_lugsail_batch_means_est([(np.zeros(5), 0)]) == np.inf


# ------------------------------------------------- native ingest (rank 3)
def _write(path, mat, sep, header=False, index=False, as_float=False):
    fmt = (lambda v: f'{float(v):.1f}') if as_float else str
    lines = []
    if header:
        lines.append(sep.join((['id'] if index else [])
            + [f'c{j}' for j in range(mat.shape[1])]))
    for i, r in enumerate(mat):
        lines.append(sep.join(([f'm{i}'] if index else [])
            + [fmt(v) for v in r]))
    path.write_text('\n'.join(lines) + '\n')


@pytest.mark.parametrize('sep', [' ', '\t', ','])
@pytest.mark.parametrize('header,index', [(False, False), (True, True),
    (True, False)])
def test_native_parser_equals_python_loader(tmp_path, sep, header, index):
    rng = np.random.RandomState(3)
    mat = rng.choice([0, 1, 2, 3], size=(37, 23), p=[.5, .3, .05, .15])
    f = tmp_path / 'm.txt'
    _write(f, mat, sep, header, index)
    for transpose in (True, False):
        want = bio.data_to_codes(bio.load_data(str(f), transpose=transpose))
        got = bio.load_codes_native(str(f), transpose=transpose)
        assert got.dtype == np.int8 and np.array_equal(got, want)
    data = bio.codes_to_data(got)
    assert np.array_equal(np.isnan(data), got == 3)


def test_native_parser_floats_blanks_crlf_and_errors(tmp_path, golden_dir):
    f = tmp_path / 'f.csv'
    f.write_text('1.0,0.0,,3\r\n0,2.0,1,\r\n\r\n')
    got = bio.load_codes_native(str(f), transpose=False)
    assert got.tolist() == [[1, 0, 3, 3], [0, 1, 1, 3]]
    assert np.array_equal(got,
        bio.data_to_codes(bio.load_data(str(f), transpose=False)))
    ex = os.path.join(golden_dir, 'example_data.csv')
    assert np.array_equal(bio.load_codes_native(ex),
        bio.data_to_codes(bio.load_data(ex)))
    bad = tmp_path / 'bad.txt'
    bad.write_text('0 1 1\n' * 6 + '0 7 1\n')     # beyond the sniffed lines
    with pytest.raises(RuntimeError, match='not 0\\|1\\|2\\|3'):
        bio.load_codes_native(str(bad))
    with pytest.raises((RuntimeError, OSError)):
        bio.load_codes_native(str(tmp_path / 'missing.txt'))


def test_native_parser_throughput(tmp_path):
    """2000 x 1500 entries: the byte scanner is far faster than the Python
    loader (typically 30-60x; 3x asserted to stay robust on a loaded host)."""
    import time
    rng = np.random.RandomState(0)
    mat = rng.choice([0, 1, 3], size=(1500, 2000), p=[.6, .2, .2])
    f = tmp_path / 'big.txt'
    f.write_text('\n'.join(' '.join(map(str, r)) for r in mat))
    t0 = time.perf_counter()
    py = bio.load_data(str(f))
    t_py = time.perf_counter() - t0
    t0 = time.perf_counter()
    nat = bio.load_codes_native(str(f))
    t_nat = time.perf_counter() - t0
    assert np.array_equal(bio.data_to_codes(py), nat)
    assert t_nat * 3 < t_py, (t_nat, t_py)


# ---- bit-plane file -------------------------
@pytest.mark.parametrize('sep', [' ', '\t', ','])
@pytest.mark.parametrize('header,index', [(False, False), (True, True),
    (True, False)])
def test_bitplane_file_round_trip_equals_load_data(tmp_path, sep, header,
            index):
    """Text -> native scan -> packed planes -> file -> memory map gives the
    matrix of the reference's loader (dpmmIO.load_data) on every layout, in
    both orientations, for sizes that are not multiples of 64."""
    from bnpc_amd import bitplanes as B
    rng = np.random.RandomState(4)
    mat = rng.choice([0, 1, 2, 3], size=(70, 131), p=[.5, .3, .05, .15])
    f = tmp_path / 'm.txt'
    _write(f, mat, sep, header, index)
    for transpose in (True, False):
        want = bio.load_data(str(f), transpose=transpose)
        side = str(f) + B.SUFFIX
        if os.path.exists(side):
            os.remove(side)
        first = B.load_matrix(str(f), transpose=transpose)
        assert os.path.exists(side) and not isinstance(first.planes, np.memmap)
        again = B.load_matrix(str(f), transpose=transpose)
        assert isinstance(again.planes, np.memmap)      # read, not re-parsed
        for got in (first, again):
            assert got.shape == want.shape
            assert np.array_equal(np.asarray(got), want, equal_nan=True)
            rows = [3, 0, want.shape[0] - 1, 3]
            assert np.array_equal(got[rows], want[rows], equal_nan=True)
            assert np.array_equal(got[5], want[5], equal_nan=True)
            assert np.array_equal(got.codes(), bio.data_to_codes(want))
        # the other orientation does not accept this file: rebuilt
        other = B.load_matrix(str(f), transpose=not transpose)
        assert other.shape == want.shape[::-1]
        os.remove(side)


def test_bitplane_cache_is_tied_to_its_source(tmp_path, monkeypatch):
    from bnpc_amd import bitplanes as B
    f = tmp_path / 'm.txt'
    f.write_text('0 1 3\n1 1 0\n')
    a = B.load_matrix(str(f), transpose=False)
    assert np.asarray(a).tolist()[0][:2] == [0.0, 1.0]
    side = str(f) + B.SUFFIX
    # the text changes (same size, later mtime): the cache is not trusted
    f.write_text('1 0 3\n1 1 0\n')
    st = os.stat(str(f))
    os.utime(str(f), ns=(st.st_atime_ns, st.st_mtime_ns + 10 ** 9))
    b = B.load_matrix(str(f), transpose=False)
    assert np.asarray(b).tolist()[0][:2] == [1.0, 0.0]
    assert isinstance(B.load_matrix(str(f), transpose=False).planes, np.memmap)
    # garbage in place of the cache is ignored and replaced
    with open(side, 'wb') as g:
        g.write(b'nonsense')
    assert np.array_equal(np.asarray(B.load_matrix(str(f), transpose=False)), np.asarray(b),
        equal_nan=True)
    with pytest.raises(ValueError):
        B.load(str(f))
    # switched off: nothing is written
    os.remove(side)
    monkeypatch.setenv('BNPC_BITPLANE_CACHE', '0')
    B.load_matrix(str(f), transpose=False)
    assert not os.path.exists(side)
    # inconsistent planes are refused by the packer's counterpart
    with pytest.raises(RuntimeError, match='not 0\\|1\\|2\\|3'):
        B.BitPlanes.from_codes(np.array([[0, 5]], dtype=np.int8))


def test_model_on_bit_planes_walks_the_same_chain(monkeypatch):
    """A model given BitPlanes instead of the float64 matrix: same chain
    (row gathers come from the planes), picklable, deep-copyable."""
    import copy
    import pickle
    from bnpc_amd import _lib, bitplanes as B, model as P
    from fake_device import FakeContext
    import test_host_logic as H
    monkeypatch.setattr(_lib, 'Context', FakeContext)
    data = H.synth(12, 60, 70, 3, 0.2)
    planes = B.BitPlanes.from_data(data)
    assert np.array_equal(np.asarray(planes), data, equal_nan=True)
    res = []
    for d in (data, planes):
        res.append(H.run_chain(H.make(P, 'learn', d), 25, 7, sm_prob=.4,
            eup=.25))
    assert np.array_equal(res[0]['assignments'], res[1]['assignments'])
    assert np.array_equal(res[0]['ML'], res[1]['ML'])
    m = H.make(P, 'fixed', planes)
    np.random.seed(1)
    m.init()
    for clone in (pickle.loads(pickle.dumps(m)), copy.deepcopy(m)):
        assert isinstance(clone.data, B.BitPlanes)
        assert np.array_equal(clone.data.planes, planes.planes)
        assert np.array_equal(clone.parameters, m.parameters)


def test_bit_planes_index_like_an_ndarray():
    """BitPlanes stands in for the float64 matrix, so bp[idx] must
    be np.asarray(bp)[idx] for integers (negative ones too), integer arrays,
    slices and boolean masks - and raise IndexError otherwise."""
    from bnpc_amd.bitplanes import BitPlanes
    rng = np.random.RandomState(2)
    data = (rng.random_sample((7, 70)) < 0.4).astype(np.float64)
    data[rng.random_sample(data.shape) < 0.2] = np.nan
    bp = BitPlanes.from_data(data)
    mask = np.array([True, False, True, False, False, True, False])
    for idx in (3, -1, -7, np.int64(2), [0, 6, 3], np.array([-1, 0, -7]),
            mask, slice(1, 6, 2), slice(None), np.array([], dtype=int),
            ~mask):
        np.testing.assert_array_equal(bp[idx], data[idx])
    assert bp[mask].shape == (3, 70)
    for bad in (7, -8, [0, 9], np.array([True, False]), 1.5, (1, 2)):
        with pytest.raises(IndexError):
            bp[bad]
