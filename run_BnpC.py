#!/usr/bin/env python3
"""Command line of the reference (/root/reference/run_BnpC.py:13-196), same
flags and defaults, driving the MI355X model classes (libs/CRP.py,
libs/CRP_learning_errors.py of this repo).

    python run_BnpC.py <DATA> [options]

Kept verbatim from the reference: every flag, its destination name, type,
default and choices - including the defaults that differ from their help text
(-FP_m 0.01, -sms 3) and `-t` being store_false.  Outputs are restated
minimally (file I/O and reporting are out of scope):
args.txt, assignment.txt and errors.txt for the posterior (MPEAR, chains
pooled), ML and MAP estimators; plots, genotype tables and metrics are not
part of this build.
"""
import argparse
from datetime import datetime
import os

VERSION = '0.2.1'


def _ratio(val):
    val = float(val)
    if val <= 0 or val >= 1:
        raise argparse.ArgumentTypeError(
            f'Invalid value: {val}. Values need to be 0 < x < 1')
    return val


def _percent(val):
    val = float(val)
    if val < 0 or val > 1:
        raise argparse.ArgumentTypeError(
            f'Invalid value: {val}. Values need to be 0 <= x <= 1')
    return val


def _psrf_cutoff(val):
    val = float(val)
    if val < 1 or val > 1.5:
        raise argparse.ArgumentTypeError(
            f'Invalid value: {val}. Values need to be 1 <= x <= 1.5')
    return val


# (group, short, long, keyword arguments) - one row per reference flag
FLAGS = [
    (None, '-t', '--transpose', dict(action='store_false',
        help='Transpose the input matrix. Default = True.')),
    (None, None, '--debug', dict(action='store_true', default=False,
        help='Run single chain in main python thread.')),
    ('model', '-FN', '--falseNegative', dict(type=float, default=-1,
        help='Fixed error rate for false negatives.')),
    ('model', '-FP', '--falsePositive', dict(type=float, default=-1,
        help='Fixed error rate for false positives.')),
    ('model', '-FN_m', '--falseNegative_mean', dict(type=_ratio, default=0.2,
        help='Prior mean of the false negative rate. Default = 0.2.')),
    ('model', '-FN_sd', '--falseNegative_std', dict(type=_ratio, default=0.1,
        help='Prior standard dev. of the false negative rate.')),
    ('model', '-FP_m', '--falsePositive_mean', dict(type=_ratio, default=0.01,
        help='Prior mean of the false positive rate.')),
    ('model', '-FP_sd', '--falsePositive_std', dict(type=_ratio, default=0.01,
        help='Prior standard dev. of the false positive rate.')),
    ('model', '-ap', '--DPa_prior', dict(type=float, nargs=2,
        default=[-1, -1], help='Gamma(a, b) prior of the CRP concentration. '
        'Default = (sqrt(#cells), 1).')),
    ('model', '-pp', '--param_prior', dict(type=float, nargs=2,
        default=[.25, .25], help='Beta(a, b) parameter prior.')),
    ('model', '-fa', '--fixed_assignment', dict(type=str, default='',
        help='File with a cluster assignment that is used and not updated.')),
    ('MCMC', '-n', '--chains', dict(type=int, default=1,
        help='Number of chains (one per GPU). Default = 1.')),
    ('MCMC', '-s', '--steps', dict(type=int, default=5000,
        help='Number of MCMC steps. Default = 5000.')),
    ('MCMC', '-r', '--runtime', dict(type=int, default=-1,
        help='Runtime in minutes; overrides steps. Default = -1.')),
    ('MCMC', '-ls', '--lugsail', dict(type=_psrf_cutoff, default=-1,
        help='Lugsail batch means PSRF threshold (e.g. 1.05).')),
    ('MCMC', '-b', '--burn_in', dict(type=_percent, default=0.33,
        help='Ratio of MCMC steps treated as burn-in. Default = 0.33.')),
    ('MCMC', '-cup', '--conc_update_prob', dict(type=_percent, default=0.25,
        help='Probability of updating the CRP concentration parameter.')),
    ('MCMC', '-eup', '--error_update_prob', dict(type=_percent, default=0.25,
        help='Probability of updating the error rates.')),
    ('MCMC', '-smp', '--split_merge_prob', dict(type=_percent, default=0.33,
        help='Probability of a split/merge step instead of Gibbs.')),
    ('MCMC', '-sms', '--split_merge_steps', dict(type=int, default=3,
        help='Restricted Gibbs scans during a split-merge move.')),
    ('MCMC', '-smr', '--split_merge_ratios', dict(type=_percent, nargs=2,
        default=[0.75, 0.25], help='Ratio of splits/merges.')),
    ('MCMC', '-e', '--estimator', dict(type=str, default='posterior',
        nargs='+', choices=['posterior', 'ML', 'MAP'],
        help='Estimator(s) used for inference.')),
    ('MCMC', '-sc', '--single_chains', dict(action='store_true',
        default=False, help='Infer a result for each chain individually.')),
    ('MCMC', None, '--seed', dict(type=int, default=-1,
        help='Seed for random number generation. Default = random.')),
    ('output', '-o', '--output', dict(type=str, default='',
        help='Output directory. Default = "<DATA_DIR>/<TIMESTAMP>".')),
    ('output', '-v', '--verbosity', dict(type=int, default=1,
        choices=[0, 1, 2], help='Print status messages. Default = 1.')),
    ('output', '-np', '--no_plots', dict(action='store_true', default=False,
        help='Accepted for compatibility (this build never plots).')),
    ('output', '-tr', '--tree', dict(type=str, default='',
        help='Accepted for compatibility (tree colouring is out of scope).')),
    ('output', '-tc', '--true_clusters', dict(type=str, default='',
        help='Accepted for compatibility (metrics are out of scope).')),
    ('output', '-td', '--true_data', dict(type=str, default='',
        help='Accepted for compatibility (metrics are out of scope).')),
]


def build_parser():
    parser = argparse.ArgumentParser(prog='BnpC',
        usage='python3 run_BnpC.py <DATA> [options]',
        description='*** Clustering of single cell data based on a '
            'Dirichlet process (MI355X build). ***')
    parser.add_argument('--version', action='version', version=VERSION)
    parser.add_argument('input', help='Path to the input matrix (mutations x '
        'cells by default, entries 0|1, 3 or empty for missing).')
    groups = {None: parser}
    for group, short, long_, kw in FLAGS:
        if group not in groups:
            groups[group] = parser.add_argument_group(group)
        names = [n for n in (short, long_) if n]
        groups[group].add_argument(*names, **kw)
    return parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def save_outputs(args, results, data, out_dir):
    from bnpc_amd import postproc
    ests = [args.estimator] if isinstance(args.estimator, str) \
        else list(args.estimator)
    chains = list(enumerate(results)) if args.single_chains else [('mean', None)]
    rows_a, rows_e = [], []
    for est in ests:
        if est == 'posterior':
            # the reference's per-chain posterior (-sc) indexes its parameter
            # trace inconsistently (utils.py:228-229); chains are pooled here
            inf = postproc.posterior_estimate(results, data)
            rows_a.append(('mean', est,
                ' '.join(str(i) for i in inf['assignment'])))
            rows_e.append(('mean', est,
                f'{inf["FN"][0]:.4f}+-{inf["FN"][1]:.4f}',
                round(float(inf['FN_geno']), 4),
                f'{inf["FP"][0]:.8f}+-{inf["FP"][1]:.8f}',
                round(float(inf['FP_geno']), 8)))
            if args.verbosity > 0:
                print(f'posterior: {len(set(inf["assignment"]))} clusters, '
                    f'FN {inf["FN"][0]:.4f}, FP {inf["FP"][0]:.6f}')
            continue
        for chain, res in chains:
            res = res if res is not None else postproc.best_chain(results, est)
            inf = postproc.point_estimate(res, est, data)
            rows_a.append((chain, est,
                ' '.join(str(i) for i in inf['assignment'])))
            rows_e.append((chain, est, round(float(inf['FN']), 4),
                round(float(inf['FN_geno']), 4), round(float(inf['FP']), 8),
                round(float(inf['FP_geno']), 8)))
            if args.verbosity > 0:
                print(f'{est} (chain {chain}): step {inf["step"]}, '
                    f'{len(set(inf["assignment"]))} clusters, '
                    f'FN {inf["FN"]:.4f}, FP {inf["FP"]:.6f}')
    with open(os.path.join(out_dir, 'assignment.txt'), 'w') as f:
        f.write('chain\testimator\tAssignment\n')
        for row in rows_a:
            f.write('\t'.join(str(x) for x in row) + '\n')
    with open(os.path.join(out_dir, 'errors.txt'), 'w') as f:
        f.write('chain\testimator\tFN_model\tFN_data\tFP_model\tFP_data\n')
        for row in rows_e:
            f.write('\t'.join(str(x) for x in row) + '\n')
    with open(os.path.join(out_dir, 'args.txt'), 'w') as f:
        for key, val in vars(args).items():
            if key == 'time':
                val = [f'{t:%Y%m%d_%H:%M:%S}' for t in val]
            f.write(f'{key}: {val}\n')


def main(args):
    from bnpc_amd import io as bio
    from bnpc_amd import postproc
    from libs.MCMC import MCMC

    if os.path.getsize(args.input) > (4 << 20):
        # large matrices (no row / column names): the packed bit planes, read
        # from the file next to the input when it is current, else scanned
        # natively and written there for the next run; the model and the
        # device take them as they are (no float64 matrix)
        from bnpc_amd import bitplanes
        data = bitplanes.load_matrix(args.input, transpose=args.transpose)
    else:
        data, _ = bio.load_data(args.input, transpose=args.transpose,
            get_names=True)
    assert data.size > 0, f'Could not read data from file: {args.input}'

    # fixed error rates only if BOTH are given (run_BnpC.py:249-262)
    if args.falsePositive > 0 and args.falseNegative > 0:
        args.error_update_prob = 0
        import libs.CRP as mod
        model = mod.CRP(data, DP_alpha=args.DPa_prior,
            param_beta=args.param_prior, FN_error=args.falseNegative,
            FP_error=args.falsePositive)
    else:
        import libs.CRP_learning_errors as mod
        model = mod.CRP_errors_learning(data, DP_alpha=args.DPa_prior,
            param_beta=args.param_prior, FP_mean=args.falsePositive_mean,
            FP_sd=args.falsePositive_std, FN_mean=args.falseNegative_mean,
            FN_sd=args.falseNegative_std)

    args.time = [datetime.now()]
    run_var, run_str = bio.get_mcmc_termination(args)
    mcmc = MCMC(model, sm_prob=args.split_merge_prob,
        dpa_prob=args.conc_update_prob, error_prob=args.error_update_prob,
        sm_ratios=args.split_merge_ratios, sm_steps=args.split_merge_steps)
    if args.verbosity > 0:
        print(model)
        print(mcmc)
        print(f'Run MCMC with ({args.chains} chains {run_str}):')
    if args.debug:
        args.chains = 1

    mcmc.run(run_var, args.seed, args.chains, args.verbosity,
        args.fixed_assignment, args.debug)
    args.chain_seeds = [int(s) for s in mcmc.get_seeds()]
    results = mcmc.get_results()
    args.time.append(datetime.now())

    args.PSRF = float(postproc.get_lugsail_batch_means_est(
        [(r['ML'], r['burn_in']) for r in results]))
    args.steps = [int(r['ML'].size) for r in results]
    out_dir = bio.get_out_dir(args)
    if args.verbosity > 0:
        bio.show_MCMC_summary(args.time[0], args.time[1], results)
        print(f'Lugsail PSRF:\t\t{args.PSRF:.5f}')
        print(f'\nWriting output to: {out_dir}\n')
    if hasattr(data, 'planes'):
        data = data.codes()     # 0 | 1 | 3: all the estimators compare with
    save_outputs(args, results, data, out_dir)
    return results


if __name__ == '__main__':
    main(parse_args())
