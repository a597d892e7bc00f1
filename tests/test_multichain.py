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
    devices, distinct = bench.devices_of(ranks, 5)
    out.put((rank, (el, devices, distinct)))
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
    assert abs(got[0][0] - got[1][0]) < 1e-9
    assert 0.09 < got[0][0] < 0.5
    # ... and every rank's identity (device ordinal, PCI bus id, own rate) is
    # gathered on all of them: what the line's `devices`, `per_rank_steps_s`
    # and `n_gpus` (distinct bus ids) are made of.  No GPU here: no bus id,
    # no count of distinct GPUs.
    for r in (0, 1):
        devices, distinct = got[r][1], got[r][2]
        assert [d['rank'] for d in devices] == [0, 1]
        assert all(d['pci_bus_id'] is None for d in devices)
        assert distinct is None
        # own rates: rank 0 sleeps 10 ms a step, rank 1 20 ms
        assert 60 < devices[0]['steps_s'] < 101
        assert 30 < devices[1]['steps_s'] < 51
    assert got[0][1] == got[1][1]


def _bench_cmd(args, env=None, timeout=180):
    import subprocess
    e = dict(os.environ)
    for name in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE',
            'MASTER_ADDR', 'MASTER_PORT', 'BNPC_HOST_SHARE',
            'BNPC_HOST_SPIN_US'):
        e.pop(name, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')]
        + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_bench_command_with_gpus_2():
    """`bench.py --gpus N` as a COMMAND, no launcher around it,
    runs N ranks (it starts them itself before any GPU call; --dry: the rank
    harness alone, no chain, no device) and rank 0's line says so."""
    import json
    res = _bench_cmd(['--gpus', '2', '--steps', '5', '--dry'])
    assert res.returncode == 0, res.stderr
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line['ranks'] == 2 and line['gpus_asked'] == 2
    assert [d['rank'] for d in line['devices']] == [0, 1]
    assert [d['local_rank'] for d in line['devices']] == [0, 1]
    assert all(d['steps_s'] > 0 for d in line['devices'])
    # exactly one line: the other rank prints nothing
    assert len([ln for ln in res.stdout.splitlines() if ln.startswith('{')]) \
        == 1
    # one rank needs no launcher and no torch
    res = _bench_cmd(['--gpus', '1', '--steps', '3', '--dry'])
    assert res.returncode == 0, res.stderr
    assert json.loads(res.stdout.strip().splitlines()[-1])['ranks'] == 1


def test_bench_command_refuses_a_launcher_that_disagrees_with_gpus():
    res = _bench_cmd(['--gpus', '2', '--dry'], env={'WORLD_SIZE': '3',
        'RANK': '0', 'LOCAL_RANK': '0'})
    assert res.returncode == 2
    assert '--gpus 2 but WORLD_SIZE=3' in res.stderr and not res.stdout


def test_bench_command_reports_a_rank_that_fails():
    """A rank that dies takes the command down with its exit code instead of
    leaving the others in a barrier."""
    res = _bench_cmd(['--gpus', '2', '--steps', '2', '--dry',
        '--dry-fail-rank', '1'], timeout=120)
    assert res.returncode == 7
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith('{')]


def test_record_targets_follow_the_trace_arrays():
    """TraceStore.record_target (where a natively made step writes what it
    records) caches the arrays' base addresses per SET of arrays: a trace
    that grows or whose parameter block is re-padded gets new arrays, and the
    targets must point into those - slot by slot, row by row."""
    from bnpc_amd.mcmc import TraceStore
    from bnpc_amd._lib import ptr

    class Model:
        cells_per_cluster = {3: 2, 0: 1, 7: 4}
        parameters = np.arange(8 * 5, dtype=np.float32).reshape(8, 5)
    tr = TraceStore(6, 7, 5)

    def check(slot, with_params):
        scalars, labels, block, cap = tr.record_target(slot, with_params)
        d = tr.data
        assert scalars == [ptr(d[k]) + 8 * slot for k in tr.SCALARS]
        assert labels == ptr(d['assignments']) + 8 * slot * 7
        if with_params and 'params' in d:
            row = slot - (tr.slots - d['params'].shape[0])
            if 0 <= row < d['params'].shape[0]:
                assert cap == d['params'].shape[1]
                assert block == ptr(d['params']) + 4 * row * cap * 5
                return
        assert block == 0 and cap == 0
    check(2, True)                  # no parameter block yet
    tr.put_params(3, Model)         # ... now there is (slots 3..5)
    check(2, True)
    check(4, True)
    check(4, False)
    tr.grow(4, with_params=True)    # new arrays all round
    check(4, True)
    check(8, True)
    Model.cells_per_cluster = {i: 1 for i in range(8)}
    big = np.arange(8 * 5, dtype=np.float32).reshape(8, 5)
    Model.parameters = big
    tr.data['params'] = np.pad(tr.data['params'], [(0, 0), (0, 9), (0, 0)])
    check(5, True)                  # a re-padded (replaced) parameter block
    import pickle
    again = pickle.loads(pickle.dumps(tr))
    assert '_targets' not in again.__dict__
    s2, l2, b2, c2 = again.record_target(5, True)
    assert l2 == ptr(again.data['assignments']) + 8 * 5 * 7


@pytest.mark.parametrize('threads', ['1', '3', '8'])
def test_large_traces_are_first_written_by_the_team(threads, monkeypatch):
    """The large sample traces (np.zeros / np.pad / np.append of
    libs/MCMC.py:267-294) are written for the first time by the host team
    (bnpc_rows_copy_zero): the arrays must be what NumPy's calls give."""
    from bnpc_amd.mcmc import TraceStore
    from bnpc_amd import _lib
    monkeypatch.setenv('BNPC_HOST_THREADS', threads)
    rng = np.random.RandomState(5)
    z = TraceStore._zeros((37, 11, 701), dtype=np.float32)     # > 1 MiB
    assert z.dtype == np.float32 and z.shape == (37, 11, 701) \
        and z.flags['C_CONTIGUOUS'] and not z.any()
    lab = TraceStore._zeros((300, 1000), dtype=int)
    assert lab.dtype == np.int64 and not lab.any()
    assert not TraceStore._zeros((3, 5)).any()                  # small: NumPy
    old = rng.rand(37, 11, 701).astype(np.float32)
    wide = TraceStore._widened(old, (37, 19, 701))
    assert np.array_equal(wide, np.pad(old, [(0, 0), (0, 8), (0, 0)]))
    long = TraceStore._widened(old, (45, 11, 701))
    assert np.array_equal(long, np.append(old,
        np.zeros((8, 11, 701), np.float32), axis=0))
    same = TraceStore._widened(old, old.shape)
    assert np.array_equal(same, old) and same is not old
    # one row, a row narrower than a page, nothing at all
    one = _lib.rows_copy_zero(np.full((1, 9), 7.0), np.ones((1, 4)))
    assert one.tolist() == [[1.0] * 4 + [0.0] * 5]
    assert _lib.rows_copy_zero(np.empty((0, 4))).shape == (0, 4)
    with pytest.raises(AssertionError):
        _lib.rows_copy_zero(np.empty((4, 4)), np.ones((3, 4)))
    # a trace that records, re-pads and grows keeps its samples
    class Model:
        cells_per_cluster = {3: 2, 0: 1, 7: 4}
        parameters = rng.rand(40, 300).astype(np.float32)
    tr = TraceStore(1200, 7, 300)
    tr.put_params(5, Model)
    first = tr.data['params'][0, :3].copy()
    assert np.array_equal(first, Model.parameters[[0, 3, 7]])
    Model.cells_per_cluster = {i: 1 for i in range(3 + tr.PARAMS_SPARE + 1)}
    tr.put_params(6, Model)
    assert tr.data['params'].shape[1] == 3 + 2 * tr.PARAMS_SPARE + 1
    assert np.array_equal(tr.data['params'][0, :3], first) \
        and not tr.data['params'][0, 3:].any() \
        and not tr.data['params'][2:].any()
    tr.grow(50, with_params=True)
    assert tr.data['params'].shape[0] == 1200 - 5 + 50
    assert np.array_equal(tr.data['params'][0, :3], first) \
        and not tr.data['params'][2:].any()


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


# ---- chain -> device placement (libs/MCMC.py:100-131 has no GPUs to place) ---
def _report_device(i):
    from bnpc_amd import mcmc
    mcmc._bind_worker_to_gpu(i)
    return i, int(os.environ['BNPC_DEVICE'])


def _placements(n_chains):
    from bnpc_amd import handoff
    got = {}
    fails = handoff.run_jobs(_report_device, [(i,) for i in range(n_chains)],
        lambda r: got.__setitem__(*r))
    assert not fails
    return [got[i] for i in range(n_chains)]


def test_eight_workers_get_eight_distinct_devices(monkeypatch):
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES',
            'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv('BNPC_NUM_DEVICES', '8')
    # a BNPC_DEVICE inherited from the parent must not pin every chain
    monkeypatch.setenv('BNPC_DEVICE', '5')
    assert _placements(8) == list(range(8))
    assert _placements(10) == list(range(8)) + [0, 1]


def test_placement_honours_the_visible_device_lists(monkeypatch):
    from bnpc_amd import mcmc
    monkeypatch.delenv('BNPC_NUM_DEVICES', raising=False)
    monkeypatch.delenv('CUDA_VISIBLE_DEVICES', raising=False)
    monkeypatch.setattr(mcmc, 'KFD_NODES', '/nonexistent')
    # a 4-entry list: ordinals 0..3, whatever the physical ids are
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '2,3,5,7')
    monkeypatch.delenv('ROCR_VISIBLE_DEVICES', raising=False)
    assert mcmc._visible_gpus() == 4
    assert _placements(8) == [0, 1, 2, 3, 0, 1, 2, 3]
    # ROCR filters first, HIP indexes into what is left
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '0,1,2,3,4,5')
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '4,5')
    assert mcmc._visible_gpus() == 2
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    assert mcmc._visible_gpus() == 6
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '3')
    assert _placements(3) == [0, 0, 0]


def test_placement_counts_kfd_gpu_nodes(monkeypatch, tmp_path):
    from bnpc_amd import mcmc
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES',
            'CUDA_VISIBLE_DEVICES', 'BNPC_NUM_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    for node, simds in enumerate((0, 0, 1024, 1024, 1024)):    # 2 CPUs, 3 GPUs
        d = tmp_path / str(node)
        d.mkdir()
        (d / 'properties').write_text(f'cpu_cores_count 0\nsimd_count {simds}\n')
    monkeypatch.setattr(mcmc, 'KFD_NODES', str(tmp_path))
    assert mcmc._visible_gpus() == 3
    assert [mcmc.device_for_chain(i) for i in range(4)] == [0, 1, 2, 0]


# ---- hand-off through shared memory ---------------------------------------
def _big_result(i, n):
    rng = np.random.RandomState(i)
    return {'no': i, 'labels': rng.randint(0, 9, size=(n, 50)),
        'trace': rng.uniform(size=n), 'tag': 'x' * 10}


def _dies_quietly(i):
    if i == 1:
        os._exit(3)
    return i


def test_results_return_through_shared_memory_and_it_is_released():
    from bnpc_amd import handoff
    before = set(os.listdir('/dev/shm'))
    got = []
    fails = handoff.run_jobs(_big_result, [(i, 40000) for i in range(3)],
        got.append)
    assert not fails and sorted(r['no'] for r in got) == [0, 1, 2]
    for r in got:
        want = _big_result(r['no'], 40000)
        assert np.array_equal(r['labels'], want['labels'])
        assert np.array_equal(r['trace'], want['trace'])
        r['trace'][0] = 1.0             # arrays are private and writable
    assert set(os.listdir('/dev/shm')) == before
    # the in-band part is small: the arrays travelled out of band
    data, name, sizes = handoff.pack(_big_result(0, 40000))
    assert len(data) < 2000 and sum(sizes) > 40000 * 50 * 8
    assert handoff.unpack((data, name, sizes))['tag'] == 'x' * 10


def test_a_worker_that_dies_is_reported():
    from bnpc_amd import handoff
    got = []
    fails = handoff.run_jobs(_dies_quietly, [(i,) for i in range(3)],
        got.append)
    assert sorted(got) == [0, 2]
    assert len(fails) == 1 and 'exit code 3' in fails[0]


def test_workers_are_spawned_once_the_parent_touched_the_gpu(monkeypatch):
    from bnpc_amd import handoff
    monkeypatch.setattr(_lib, '_touched', {'pid': os.getpid()})
    got = []
    # a spawned child re-imports this module: the job must be importable
    fails = handoff.run_jobs(_big_result, [(4, 10)], got.append)
    assert not fails and got[0]['no'] == 4


def test_single_rank_bench_harness_needs_no_torch():
    """bench.py --gpus 1 must run where torch is not installed (the product
    has no PyTorch dependency): the import is blocked in a child process."""
    import subprocess
    code = ("import sys; sys.modules['torch'] = None; "
        f"sys.path.insert(0, {ROOT!r}); import bench; "
        "r = bench.Ranks().init(); "
        "t = bench.timed_steps(r, lambda i: None, 1, 3); r.close(); "
        "import bnpc_amd.model, bnpc_amd.mcmc, libs.CRP; "
        "assert 'torch.distributed' not in sys.modules; print('ok', t >= 0)")
    env = {k: v for k, v in os.environ.items()
        if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, '-c', code], capture_output=True,
        text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().endswith('ok True')


def test_host_threads_are_kept_on_the_devices_numa_node(monkeypatch, tmp_path):
    """bind_near_device: PCI address -> numa_node -> cpulist -> affinity."""
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 2:
        pytest.skip('needs two CPUs')
    pci = tmp_path / 'pci' / '0000:0d:00.0'
    pci.mkdir(parents=True)
    (pci / 'numa_node').write_text('1\n')
    half = have[:len(have) // 2]
    for node, cpus in ((0, have[len(have) // 2:]), (1, half)):
        d = tmp_path / 'node' / f'node{node}'
        d.mkdir(parents=True)
        (d / 'cpulist').write_text(','.join(str(c) for c in cpus) + '\n')

    class Lib:
        @staticmethod
        def bnpc_device_pci_bus_id(device, buf, n):
            buf.value = b'0000:0D:00.0'
            return 0
    monkeypatch.setattr(_lib, 'load', lambda: Lib)
    try:
        monkeypatch.setenv('BNPC_NUMA_BIND', '0')
        assert _lib.bind_near_device(0, str(tmp_path / 'pci'),
            str(tmp_path / 'node')) is None
        assert sorted(os.sched_getaffinity(0)) == have
        monkeypatch.setenv('BNPC_NUMA_BIND', '1')
        assert _lib.bind_near_device(0, str(tmp_path / 'pci'),
            str(tmp_path / 'node')) == 1
        assert sorted(os.sched_getaffinity(0)) == half
        # no sysfs entry for the device: nothing changes
        assert _lib.bind_near_device(0, str(tmp_path / 'nope'),
            str(tmp_path / 'node')) is None
        # a second context on a GPU of the OTHER node re-binds
        # against the mask the process started with (not against the first
        # binding, whose intersection with the other node is empty) ...
        (pci / 'numa_node').write_text('0\n')
        assert _lib.bind_near_device(0, str(tmp_path / 'pci'),
            str(tmp_path / 'node')) == 0
        assert sorted(os.sched_getaffinity(0)) == have[len(have) // 2:]
        # ... threads that exist already (the host thread team) move along ...
        import threading
        seen, go = [], threading.Event()

        def parked():
            go.wait(10)
            seen.append(sorted(os.sched_getaffinity(0)))
        t = threading.Thread(target=parked)
        t.start()
        (pci / 'numa_node').write_text('1\n')
        assert _lib.bind_near_device(0, str(tmp_path / 'pci'),
            str(tmp_path / 'node')) == 1
        go.set()
        t.join()
        assert seen == [half]
        # ... and closing the last bound context restores the original mask
        _lib.release_binding()
        _lib.release_binding()
        assert sorted(os.sched_getaffinity(0)) == half
        _lib.release_binding()
        assert sorted(os.sched_getaffinity(0)) == have
    finally:
        os.sched_setaffinity(0, have)
        _lib._affinity.update(pid=None, original=None, bound=0)


def test_host_threads_are_divided_among_chains_sharing_a_gpu(monkeypatch):
    """8 chains on 2 GPUs: 4 per GPU (and per NUMA node): each worker's
    default team is a quarter of the node's logical CPUs divided by 4, and
    idle ranks spin briefly; a chain alone on its GPU keeps the full team; an
    explicit BNPC_HOST_THREADS / BNPC_HOST_SPIN_US wins."""
    from bnpc_amd import mcmc
    monkeypatch.delenv('BNPC_HOST_THREADS', raising=False)
    monkeypatch.delenv('BNPC_HOST_SPIN_US', raising=False)
    monkeypatch.delenv('BNPC_HOST_SHARE', raising=False)
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(128)))
    monkeypatch.setattr(os, 'cpu_count', lambda: 256)
    monkeypatch.setattr(_lib, '_threads_memo', {})
    monkeypatch.setattr(_lib, 'numa_node_count', lambda: 2)
    # 8 GPUs on 2 nodes, 8 chains: 4 chains per node although 1 per GPU
    assert _lib.host_share(8, 8) == 4 and _lib.host_share(2, 8) == 1
    assert _lib.host_share(8, 1) == 8 and _lib.host_share(3, 8, nodes=1) == 3
    # greedy or frugal goes by the logical CPUs a chain is left
    # with on its node, not by "shares a node": the 8-GPU target (4 chains
    # per 128-CPU node: 32 each) keeps the settings of a chain alone
    got = _lib.host_settings(8, 8, apply=False)
    assert got == {'share': 4, 'cpus_per_chain': 32, 'greedy': True,
        'spin_us': 300}
    assert _lib.host_settings(64, 8, apply=False)['greedy'] is False
    assert _lib.host_settings(64, 8, apply=False)['cpus_per_chain'] == 4
    assert 'BNPC_HOST_SHARE' not in os.environ
    for chain in range(8):
        mcmc._bind_worker_to_gpu(chain, n_devices=2, n_chains=8)
        assert os.environ['BNPC_HOST_SHARE'] == '4'
        assert 'BNPC_HOST_SPIN_US' not in os.environ
        assert os.environ['BNPC_DEVICE'] == str(chain % 2)
        assert _lib.host_threads() == 8
        assert _lib.threads_for(10 ** 6) == 8
    # a small host: 8 chains on the 8 CPUs of one node give ranks back at once
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(8)))
    monkeypatch.setattr(os, 'cpu_count', lambda: 8)
    monkeypatch.setattr(_lib, 'numa_node_count', lambda: 1)
    monkeypatch.setattr(_lib, '_threads_memo', {})
    mcmc._bind_worker_to_gpu(3, n_devices=2, n_chains=8)
    assert os.environ['BNPC_HOST_SHARE'] == '8'
    assert os.environ['BNPC_HOST_SPIN_US'] == '5'
    assert _lib.host_threads() == 1
    monkeypatch.delenv('BNPC_HOST_SPIN_US')
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(128)))
    monkeypatch.setattr(os, 'cpu_count', lambda: 256)
    monkeypatch.setattr(_lib, 'numa_node_count', lambda: 2)
    monkeypatch.setattr(_lib, '_threads_memo', {})
    mcmc._bind_worker_to_gpu(0, n_devices=8, n_chains=2)
    assert os.environ['BNPC_HOST_SHARE'] == '1'
    assert 'BNPC_HOST_SPIN_US' not in os.environ
    assert _lib.host_threads() == 8
    assert _lib.threads_for(10 ** 6) == 32
    monkeypatch.setenv('BNPC_HOST_THREADS', '5')
    monkeypatch.setenv('BNPC_HOST_SHARE', '4')
    assert _lib.host_threads() == 5
    monkeypatch.delenv('BNPC_HOST_SHARE')
    monkeypatch.delenv('BNPC_DEVICE', raising=False)


def test_results_nobody_reads_leave_no_shared_memory(tmp_path):
    """when the parent fails on an early result (on_done raises),
    the blocks the other workers packed are released and the workers are
    joined - nothing of this run stays in /dev/shm."""
    import glob
    from bnpc_amd import handoff

    def before():
        return set(glob.glob(f'/dev/shm/{handoff.SHM_PREFIX}*'))
    start = before()

    def boom(result):
        raise ValueError('parent failed on a result')
    with pytest.raises(ValueError):
        handoff.run_jobs(_unread_result, [(i,) for i in range(3)], boom)
    assert before() == start
    # a worker whose parent end is gone releases its own block
    recv_end, send_end = __import__('multiprocessing').Pipe(duplex=False)
    recv_end.close()
    with pytest.raises(OSError):
        handoff._worker(_unread_result, (7,), send_end)
    assert before() == start


def _unread_result(i):
    return {'i': i, 'samples': np.full((64, 1024), i, dtype=np.int64)}
