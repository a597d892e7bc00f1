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
