"""N > 1 paths on CPU: (a) the fork pool of the driver (one chain per worker,
worker i -> GPU i), (b) bench.py's rank harness under gloo, world_size 2."""
import os
import sys
import time

import numpy as np
import pytest

from bnpc_amd import _lib, model as P
from bnpc_amd.mcmc import MCMC, Chain_steps
from fake_device import FakeContext
import test_host_logic as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pool_chains_equal_in_process_chains(monkeypatch):
    monkeypatch.setattr(_lib, 'Context', FakeContext)   # inherited by fork
    monkeypatch.delenv('BNPC_DEVICE', raising=False)
    monkeypatch.setenv('BNPC_NUM_DEVICES', '2')
    data = H.synth(7, 50, 30, 3, 0.1)
    model = H.make(P, 'learn', data)
    mcmc = MCMC(model, sm_prob=.33, dpa_prob=.25, error_prob=.25,
        sm_ratios=[.75, .25], sm_steps=2)
    mcmc.run((20, 6), 42, 3, 0, '', False)
    res = mcmc.get_results()
    assert len(res) == 3 and len(mcmc.get_seeds()) == 3
    # each pooled chain equals the same chain run alone in this process
    for i, seed in enumerate(mcmc.get_seeds()):
        np.random.seed(seed)
        solo = H.make(P, 'learn', data)
        solo.init()
        chain = Chain_steps(solo, i + 1, 20, 6, mcmc.params, 0, False)
        chain.run()
        assert np.array_equal(chain.results['assignments'],
            res[i]['assignments'])
        np.testing.assert_allclose(chain.results['ML'], res[i]['ML'],
            rtol=1e-12)
    # chains differ from each other (different seeds)
    assert not np.array_equal(res[0]['assignments'], res[1]['assignments'])


def test_worker_exception_is_reported(monkeypatch):
    class Broken(FakeContext):
        def ll_theta(self, *a, **k):
            raise RuntimeError('device lost')
    monkeypatch.setattr(_lib, 'Context', Broken)
    data = H.synth(7, 30, 20, 2, 0.1)
    mcmc = MCMC(H.make(P, 'fixed', data), sm_prob=0., dpa_prob=0.,
        error_prob=0., sm_ratios=[.75, .25], sm_steps=2)
    with pytest.raises(RuntimeError, match='chain worker failed'):
        mcmc.run((12, 3), 1, 2, 0, '', False)


def _rank_main(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank),
        WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
        MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    ranks = bench.Ranks().init()
    delay = 0.01 * (rank + 1)           # rank 1 is the slow one
    el = bench.timed_steps(ranks, lambda i: time.sleep(delay), 1, 5)
    out.put((rank, el))
    ranks.close()


def test_bench_rank_harness_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + os.getpid() % 400
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out))
        for r in range(2)]
    for p in procs:
        p.start()
    got = dict(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # both ranks report the MAX over ranks: the slow rank's 5 x 20 ms
    assert abs(got[0] - got[1]) < 1e-9
    assert 0.09 < got[0] < 0.5


def test_lugsail_mode_extends_chains_until_psrf_cutoff(monkeypatch):
    """-ls: chains run max(10, 1/(c^2-1)) steps, then are extended by 200
    steps through the pool until the lugsail PSRF drops under the cutoff
    (libs/MCMC.py:85-90, 138-189)."""
    monkeypatch.setattr(_lib, 'Context', FakeContext)
    data = H.synth(9, 40, 25, 2, 0.1)
    mcmc = MCMC(H.make(P, 'fixed', data), sm_prob=.2, dpa_prob=.25,
        error_prob=0., sm_ratios=[.75, .25], sm_steps=2)
    mcmc.run((1.2, 0), 3, 2, 0, '', False)
    res = mcmc.get_results()
    assert len(res) == 2
    for r in res:
        steps = r['ML'].size
        assert steps >= 11 and (steps - 11) % 200 == 0
        assert r['PSRF_cutoff'] == 1.2 and r['PSRF'][-1][1] <= 1.2
        assert r['burn_in'] == (r['PSRF'][-1][0] // 2) + 1
        assert r['params'].shape[0] == steps - r['burn_in'] or \
            r['params'].shape[0] <= steps
        assert np.all(r['ML'] != 0)


def test_runtime_mode_and_fixed_assignment(monkeypatch, tmp_path):
    """-r (wall-clock termination, Chain_time, MCMC.py:395-440) and -fa
    (a fixed assignment that is used and never updated, MCMC.py:96, 321)."""
    from datetime import datetime, timedelta
    from oracle import crp_numpy as O
    monkeypatch.setattr(_lib, 'Context', FakeContext)
    data = H.synth(10, 40, 25, 3, 0.1)

    # runtime mode: traces are trimmed to the steps actually run
    mcmc = MCMC(H.make(P, 'fixed', data), sm_prob=.2, dpa_prob=.25,
        error_prob=0., sm_ratios=[.75, .25], sm_steps=2)
    t0 = datetime.now()
    run_var = (t0 + timedelta(seconds=1.5), t0 + timedelta(seconds=0.5))
    mcmc.run(run_var, 5, 1, 0, '', True)
    res = mcmc.get_results()[0]
    n = res['ML'].size
    assert 3 < n < 100000 and np.all(res['MAP'] != 0)
    assert res['assignments'].shape == (n, 40)
    assert res['burn_in'] == n - res['params'].shape[0] and res['burn_in'] > 0

    # fixed assignment: product and oracle walk the same parameter trajectory
    truth = list(np.random.RandomState(1).randint(0, 3, 40))
    f = tmp_path / 'assign.txt'
    f.write_text(' '.join(str(i) for i in truth))
    outs = []
    for mod in (O, P):
        mcmc = MCMC(H.make(mod, 'learn', data), sm_prob=.33, dpa_prob=.25,
            error_prob=.25, sm_ratios=[.75, .25], sm_steps=2)
        mcmc.run((30, 10), 9, 1, 0, str(f), True)
        outs.append(mcmc.get_results()[0])
    o, p = outs
    want = np.unique(truth, return_inverse=True)[1]
    assert np.all(p['assignments'] == want) and np.all(o['assignments'] == want)
    np.testing.assert_allclose(p['ML'], o['ML'], rtol=1e-9)
    np.testing.assert_allclose(p['FN'], o['FN'], rtol=1e-9)
    assert np.array_equal(p['params'], o['params'])
