"""Helper of test_eight_chains_in_workers_equal_in_process_chains (run as a
script in a fresh interpreter, which has not touched the GPU and therefore
FORKS its chain workers, as the CLI does): `-n W` chains through MCMC.run with
the worker pool against the same chains run one by one, each alone in a
process of its own.

    python tests/eight_chains_check.py [workers] [cells] [muts] [steps]

Checks: W results, W distinct seeds and worker processes, worker i on device
ordinal device_for_chain(i) (as set in its environment AND as the context it
created reports), every trace bit-equal to the solo chain's, nothing of ours
left in /dev/shm."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

KEYS = ('assignments', 'ML', 'MAP', 'FN', 'FP', 'DP_alpha', 'params')


def make_model(data):
    import libs.CRP_learning_errors as dev
    return dev.CRP_errors_learning(data, DP_alpha=[-1, -1],
        param_beta=[.25, .25], FP_mean=0.01, FP_sd=0.01, FN_mean=0.2,
        FN_sd=0.1)


def solo_chain(seed, data, knobs, steps, burn):
    from bnpc_amd.mcmc import Chain_steps
    np.random.seed(seed)
    model = make_model(data)
    model.init()
    chain = Chain_steps(model, 1, steps, burn, knobs, 0, False)
    chain.run()
    return {key: np.array(chain.results[key]) for key in KEYS}


def main(workers=8, N=2000, M=400, steps=30):
    from bnpc_amd import _lib, handoff, mcmc as drv
    import test_host_logic as H

    class Told(drv.MCMC):
        def run_chain(self, chain_type, run_var, assign, i, verbosity):
            chain = super().run_chain(chain_type, run_var, assign, i,
                verbosity)
            chain.results['_where'] = np.array([i,
                int(os.environ['BNPC_DEVICE']), chain.model._ctx.device,
                os.getpid()])
            return chain

    data = H.synth(0, N, M, 20, 0.20)
    sampler = Told(make_model(data), sm_prob=.33, dpa_prob=.25,
        error_prob=.25, sm_ratios=[.75, .25], sm_steps=3)
    before = set(os.listdir('/dev/shm'))
    burn = steps // 3
    sampler.run((steps, burn), 42, workers, 0, '', False)
    assert not _lib.gpu_touched(), 'the parent must stay off the GPU'
    res = sampler.get_results()
    seeds = [int(s) for s in sampler.get_seeds()]
    assert len(res) == workers and len(set(seeds)) == workers
    n_dev = drv._visible_gpus()
    pids = set()
    for i, r in enumerate(res):
        chain_no, env_dev, ctx_dev, pid = (int(v) for v in r['_where'])
        want = drv.device_for_chain(i, n_dev)
        assert (chain_no, env_dev, ctx_dev) == (i, want, want), r['_where']
        pids.add(pid)
    assert len(pids) == workers and os.getpid() not in pids
    for i, seed in enumerate(seeds):
        got_solo = []       # one fresh forked process per solo chain
        failures = handoff.run_jobs(solo_chain,
            [(seed, data, sampler.params, steps, burn)], got_solo.append)
        assert not failures, failures
        solo = got_solo[0]
        for key in KEYS:
            got = np.asarray(res[i][key])
            assert got.dtype == solo[key].dtype, (i, key, got.dtype)
            assert got.shape == solo[key].shape, (i, key, got.shape)
            assert np.array_equal(got, solo[key]), (i, key)
    if workers > 1:
        assert not np.array_equal(res[0]['assignments'],
            res[1]['assignments'])
    left = set(os.listdir('/dev/shm')) - before
    ours = [name for name in left if name.startswith(handoff.SHM_PREFIX)]
    assert not ours, ours
    print(f'{workers} workers on {n_dev} visible device(s), {steps} steps of '
        f'{N} x {M}: every chain bit-equal to its solo run; '
        f'devices {[int(r["_where"][2]) for r in res]}')
    print('EIGHT CHAINS OK')


if __name__ == '__main__':
    main(*(int(a) for a in sys.argv[1:]))
