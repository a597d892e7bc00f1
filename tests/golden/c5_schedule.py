"""Config 5 at FULL size (50000 cells x 5000 mutations, 20 % missing, learned
error rates, -smp 0.5 -sms 5) in its STEADY state: the schedule both the CPU
oracle (make_c5_trajectory.py, ~40 CPU-minutes) and the device chain
(tests/test_gpu_parity.py) walk, step for step.

The chain does not start from the reference's random state (K0 = 31 600
clusters: the CPU oracle would need days for that first sweep) but from
`init(assign=...)` (libs/CRP.py:119-152) with a deterministic near-truth
labelling - the generator's own cluster of every cell, 5 % of the cells moved
to a random other cluster - so that every move handles thousands of cells x
5000 mutations: Gibbs sweeps over ~50 clusters, split / merge moves whose
views hold 10^3 cells, parameter batches of 50 x 5000 entries.

Schedule: SCHEDULED steps of the sampler's own move schedule
(bnpc_amd.mcmc.advance = libs/MCMC.py:320-342), then one forced split and one
forced merge (do_split_move / do_merge_move, libs/CRP.py:434-524), each
followed by update_parameters, so that both move types are covered whatever
the stream decides.  Recorded after every step: assignment, ML, MAP, alpha,
FN, FP, a digest of the populated parameter rows, and the (move, cells,
accepted) log of every restricted-Gibbs run.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCHEDULED = 6
SEED = 42
RELABELLED = 0.05


def near_truth_labels(N, M, C, seed=0, frac=RELABELLED):
    """The cluster the section-8(d) generator drew for every cell (bench.synth
    with the same seed), `frac` of the cells moved to another cluster."""
    rng = np.random.RandomState(seed)
    rng.random_sample((C, M))               # the genotypes (stream position)
    z = rng.randint(0, C, N)
    mix = np.random.RandomState(seed + 1)
    moved = mix.choice(N, int(N * frac), replace=False)
    z[moved] = (z[moved] + mix.randint(1, C, moved.size)) % C
    return z


def params_digest(model):
    live = np.sort(np.fromiter(model.cells_per_cluster.keys(), dtype=np.int64))
    rows = np.ascontiguousarray(model.parameters[live], dtype=np.float32)
    return hashlib.sha256(rows.tobytes()).hexdigest()


def drive(mod, data, config='c5', scheduled=SCHEDULED, progress=None):
    """Walk the schedule with model classes from `mod` (the oracle or the
    device-backed libs.CRP_learning_errors).  Returns a dict of traces."""
    import bench
    from bnpc_amd.mcmc import advance, Tally

    N, M, C, miss, learned = bench.CONFIGS[config]
    labels = near_truth_labels(N, M, C)
    np.random.seed(SEED)
    model = bench.make_model(mod, mod, data, learned)
    model.init(assign=[int(x) for x in labels])
    knobs = dict(bench.MCMC_PARAMS, error_prob=.25,
        param_proposal_sd=np.array([0.1, 0.25, 0.5]))
    knobs.update(bench.MOVE_OVERRIDES.get(config, {}))
    sm_steps = knobs['sm_steps']

    moves = []
    if hasattr(model, '_note_move'):
        # the device-backed classes report their moves themselves (a move may
        # run as one native call, bnpc_sm_move)
        model._move_hook = lambda move, n_cells, accepted: moves.append(
            (0 if move == 'split' else 1, n_cells, int(accepted)))
    else:
        inner = model.run_rg_nc

        def logged(move, cells, size_data, scan_no):
            out = inner(move, cells, size_data, scan_no)
            moves.append((0 if move == 'split' else 1, int(len(cells)),
                int(bool(out[0]))))
            return out
        model.run_rg_nc = logged

    trace = dict(assignments=[], ML=[], MAP=[], DP_alpha=[], FN=[], FP=[],
        digest=[], K=[])

    def record(tag):
        ll = model.get_ll_full()
        trace['assignments'].append(np.array(model.assignment, dtype=np.int64))
        trace['ML'].append(ll)
        trace['MAP'].append(ll + model.get_lprior_full())
        trace['DP_alpha'].append(model.DP_a)
        trace['FN'].append(model.FN)
        trace['FP'].append(model.FP)
        trace['digest'].append(params_digest(model))
        trace['K'].append(len(model.cells_per_cluster))
        if progress:
            progress(tag, trace, moves)

    record('init')
    tally = Tally()
    for i in range(scheduled):
        advance(model, knobs, tally, False, True)
        record(f'step {i + 1}')
    model.do_split_move(sm_steps)
    model.update_parameters()
    record('forced split')
    model.do_merge_move(sm_steps)
    model.update_parameters()
    record('forced merge')

    live = np.sort(np.fromiter(model.cells_per_cluster.keys(), dtype=np.int64))
    out = {key: np.array(val) for key, val in trace.items()}
    out['moves'] = np.array(moves, dtype=np.int64).reshape(-1, 3)
    out['last_params'] = model.parameters[live].copy()
    out['stream_check'] = np.random.random()
    if hasattr(model, 'close'):
        model.close()
    return out
