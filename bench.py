#!/usr/bin/env python3
"""bench.py - MCMC steps/s of the MI355X hot path on BASELINE.json's metric
configuration (config 3: synthetic 5,000 cells x 1,000 mutations, 20 % missing,
learned error rates), one independent chain per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python bench.py --gpus N ...        (spawns its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = Chain.do_step + Chain.update_results (libs/MCMC.py:375-388 of the
reference), the unit of the reference's own "secs. per MCMC step" line
(libs/dpmmIO.py:310-315).  Warm-up steps include step 1, the sweep from the
random initial state with K0 ~ 0.63 N clusters; its time is reported
separately (first_step_s).  The timed K steps run in the converged regime in
which a chain spends all but its first step.

Rank 0 prints ONE JSON line.  `value` is the whole-job rate: steps of all
chains / max-over-ranks wall time.  Chains are independent (weak scaling, no
collective in the data path); torch.distributed (gloo) is used only for the
barrier and the max-reduction of the timings.

roofline: the cells x clusters x mutations kernel (k_ll) at the workload's
first-sweep shape, timed live with HIP events on the library's stream; the
binding roof is the FP64 vector pipe (bound "valu": N K M adds against
39.3e12 adds/s), the HBM fraction rides along.
cpu_baseline: the CPU oracle (NumPy restatement of the reference, 1 core)
stepping from the SAME post-warm-up state, on rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_ADDS_PEAK = 39.3e12       # FP64 vector: 78.6 TFLOP/s = 39.3e12 FMA|add/s

CONFIGS = {
    # name: (N, M, true clusters, missing, learned errors)
    'c2': (1000, 200, 10, 0.10, False),
    'c3': (5000, 1000, 10, 0.20, True),
    'c4': (10000, 2000, 20, 0.20, True),
    'c5': (50000, 5000, 50, 0.20, True),   # -smp 0.5 -sms 5 (MOVE_OVERRIDES)
    # many clones: a running chain with 64 < K < the tiling threshold (the
    # configs above collapse to K = 10-54).  c3k = config 3's shape with 200
    # true clusters; k150 = the same regime at a size the CPU oracle walks in
    # seconds (parity soaks, tests)
    'c3k': (5000, 1000, 200, 0.20, True),
    'k150': (2000, 500, 150, 0.20, True),
}
# move settings that differ from the CLI defaults (BASELINE.json configs)
MOVE_OVERRIDES = {'c5': dict(sm_prob=.5, sm_steps=5)}


def synth(seed, N, M, C, miss, FP_true=0.001, FN_true=0.1):
    """SURVEY.md section 8(d) generator."""
    rng = np.random.RandomState(seed)
    geno = (rng.random_sample((C, M)) < 0.3)
    z = rng.randint(0, C, N)
    X = geno[z]
    u = rng.random_sample((N, M))
    obs = np.where(X == 1, u >= FN_true, u < FP_true).astype(np.float64)
    obs[rng.random_sample((N, M)) < miss] = np.nan
    return obs


def make_model(mod_fixed, mod_learn, data, learned):
    if learned:
        # CLI defaults of the reference (run_BnpC.py:67-90)
        return mod_learn.CRP_errors_learning(data, DP_alpha=[-1, -1],
            param_beta=[.25, .25], FP_mean=0.01, FP_sd=0.01, FN_mean=0.2,
            FN_sd=0.1)
    return mod_fixed.CRP(data, DP_alpha=[-1, -1], param_beta=[.25, .25],
        FN_error=0.1, FP_error=0.001)


MCMC_PARAMS = dict(sm_prob=.33, dpa_prob=.25, sm_ratios=[.75, .25], sm_steps=3)


def new_chain(model, learned, total_steps, config=None):
    from bnpc_amd.mcmc import Chain_steps
    params = dict(MCMC_PARAMS, error_prob=.25 if learned else 0.,
        param_proposal_sd=np.array([0.1, 0.25, 0.5]))
    params.update(MOVE_OVERRIDES.get(config, {}))
    return Chain_steps(model, 1, total_steps, int(total_steps * .33), params,
        0, False)


def step(chain, i, burn_in):
    """do_step + update_results: ONE native call on the device model
    (Chain.step -> CRP.native_step), the two calls for the oracle."""
    chain.step(i, i < burn_in)


class MoveClock:
    """Wall time of the timed window by move: light wrappers around the
    model's move methods (the driver calls them through the instance), so
    that windows of different lengths can be reconciled - what a step costs
    depends on which moves it drew."""

    MOVES = ('update_assignments_Gibbs', 'update_assignments_split_merge',
        'update_DP_alpha', 'update_parameters', 'update_error_rates',
        'get_lprior_full', 'get_ll_full_deferred')

    def __init__(self, model):
        self.acc = {}
        self.on = False
        self.model = model
        # a model that makes whole steps natively keeps this clock itself
        # (bnpc_chain.clock_ns); wrapping its methods would send every step
        # back through the interpreter
        self.native0 = self._native_clocks()
        if self.native0 is not None:
            return
        for name in self.MOVES:
            fn = getattr(model, name, None)
            if fn is None:
                continue

            def timed(*a, _fn=fn, _n=name, **k):
                if not self.on:
                    return _fn(*a, **k)
                t0 = time.perf_counter()
                out = _fn(*a, **k)
                dt = time.perf_counter() - t0
                key = _n
                if _n == 'update_assignments_split_merge':
                    key = 'split' if out[1] == 0 else 'merge'
                    key += '_accepted' if out[0][0] else '_rejected'
                e = self.acc.setdefault(key, [0.0, 0])
                e[0] += dt
                e[1] += 1
                return out
            setattr(model, name, timed)

    def _native_clocks(self):
        nat = getattr(self.model, '_nat', None)
        if nat is None or os.environ.get('BNPC_NATIVE_STEP', '1') == '0':
            return None
        return list(nat.st.clock_ns), list(nat.st.clock_calls)

    def report(self, steps, elapsed):
        if self.native0 is not None:
            from bnpc_amd import _lib
            ns, calls = self._native_clocks()
            out, total = {}, 0.0
            for i, key in enumerate(_lib.STEP_CLOCKS):
                n = calls[i] - self.native0[1][i]
                t = (ns[i] - self.native0[0][i]) * 1e-9
                if n:
                    out[key] = {'calls': n,
                        'ms_per_call': round(1e3 * t / n, 4),
                        'ms_per_step': round(1e3 * t / steps, 4)}
                    if key != 'gibbs_waits_for_device':     # part of gibbs
                        total += t
            out['other'] = {'ms_per_step':
                round(1e3 * (elapsed - total) / steps, 4)}
            out['clock'] = 'native (bnpc_chain.clock_ns)'
            return out
        short = {'update_assignments_Gibbs': 'gibbs',
            'update_DP_alpha': 'dp_alpha', 'update_parameters': 'parameters',
            'update_error_rates': 'error_rates',
            'get_lprior_full': 'record_prior',
            'get_ll_full_deferred': 'record_ll_issue'}
        out, total = {}, 0.0
        for key, (t, n) in sorted(self.acc.items()):
            out[short.get(key, key)] = {'calls': n,
                'ms_per_call': round(1e3 * t / n, 4),
                'ms_per_step': round(1e3 * t / steps, 4)}
            total += t
        out['other'] = {'ms_per_step':
            round(1e3 * (elapsed - total) / steps, 4)}
        return out


class Ranks:
    """One process per GPU (torch.distributed.run contract).  No collective
    in the data path: with more than one rank, gloo carries the barrier and
    the max of the timings; a single rank needs no torch at all (the product
    has no PyTorch dependency)."""

    def __init__(self):
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.dist = None
        self.torch = None
        self.ctx = None         # the chain's device context, once it exists

    def init(self):
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self.torch = torch
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29577')
            dist.init_process_group('gloo', rank=self.rank,
                world_size=self.world)
            self.dist = dist
        # one rank per GPU; if there are more ranks than GPUs (a 2-rank smoke
        # run on a 1-GPU box) ranks share devices round-robin
        from bnpc_amd import _lib
        try:
            n_dev = max(1, _lib.device_count())
        except RuntimeError:        # no GPU: the CPU tests of this harness
            n_dev = 1
        self.device = self.local_rank % n_dev
        # ranks bound to the same NUMA node divide its CPUs (as the chains of
        # a multi-chain run do, bnpc_amd.mcmc._bind_worker_to_gpu)
        local = int(os.environ.get('LOCAL_WORLD_SIZE', self.world))
        self.host_settings = _lib.host_settings(local, n_dev, keep_env=True)
        return self

    def barrier_sync(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.ctx is not None:
            self.ctx.sync()     # hipStreamSynchronize on the chain's stream

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def gather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank (gloo)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def identity(self, own_steps_s=None):
        """What this rank ran on: its device ordinal, the PCI bus id behind
        it (two ranks on ONE GPU report the same one) and its own rate."""
        from bnpc_amd import _lib
        try:
            bus = _lib.device_pci_bus_id(self.device)
        except (RuntimeError, OSError):     # no GPU: the CPU harness tests
            bus = None
        return {'rank': self.rank, 'local_rank': self.local_rank,
            'device': self.device, 'pci_bus_id': bus,
            'steps_s': own_steps_s}

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def timed_steps(ranks, step_fn, first, last):
    """Time steps first..last inclusive: barrier + device sync on both sides,
    MAX over ranks.  ranks.own_s keeps this rank's own time up to the end of
    its last step (before the closing barrier)."""
    ranks.barrier_sync()
    t0 = time.perf_counter()
    for i in range(first, last + 1):
        step_fn(i)
    if ranks.ctx is not None:
        ranks.ctx.sync()
    ranks.own_s = time.perf_counter() - t0
    ranks.barrier_sync()
    return ranks.max_over_ranks(time.perf_counter() - t0)


def devices_of(ranks, steps):
    """Every rank's device, gathered on all ranks: (list of identities,
    number of DISTINCT GPUs among them - by PCI bus id; None without GPUs)."""
    own = getattr(ranks, 'own_s', None)
    ids = ranks.gather(ranks.identity(
        None if not own else round(steps / own, 3)))
    buses = {d['pci_bus_id'] for d in ids}
    return ids, (None if None in buses else len(buses))


def pmc_file(config='c3'):
    """The newest profiles/r*/pmc_final.json (config 3; pmc_<config>.json for
    the others) taken with THIS build's sources (its _meta.source_digest), as
    (relative path, contents); else the newest one there is (refused below,
    with the reason), else (None, None)."""
    import glob
    from bnpc_amd import build
    want = build.source_digest()
    newest = None
    name = 'pmc_final.json' if config == 'c3' else f'pmc_{config}.json'
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', name)),
            reverse=True):
        try:
            with open(path) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            continue
        rel = os.path.relpath(path, ROOT)
        if (pmc.get('_meta') or {}).get('source_digest') == want:
            return rel, pmc
        if newest is None:
            newest = (rel, pmc)
    return newest or (None, None)


def load_pmc_traffic(kernel_substr, config='c3'):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC
    passes (FETCH_SIZE and WRITE_SIZE in separate runs, KiB units), or None.
    The counters cannot be read from inside this process, so the file must be
    of THIS build: it carries the digest of the library's sources it was
    taken with (tools/pmc_collect.py; tools/r04_evidence.sh regenerates it
    and the bench line in one lease), and a file of another build is refused
    - `traffic` is then null and `traffic_source` says why.  gfx950
    correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-byte
    requests of a wide coalesced read at 64 bytes; the likelihood kernels'
    misses are taken by their 16-byte-per-lane streams (the L2 prefetch of
    k_ll8_asm, the table staging of k_ll8_lds), so the fetch counter is
    doubled (calibrated in round 3 against the same launch with the prefetch
    off: 68.3 MiB of scalar-stream fetches = 2.02 x 33.8;
    profiles/r03/README.md).  WRITE_SIZE is taken as it reads."""
    rel, pmc = pmc_file(config)
    if pmc is None:
        return None, f'no PMC passes of config {config} under profiles/'
    from bnpc_amd import build
    have = (pmc.get('_meta') or {}).get('source_digest')
    want = build.source_digest()
    if have != want:
        return None, (f'{rel} refused: taken with sources {have}, this '
            f'build is {want} (re-run the evidence script)')
    for name, ctr in pmc.items():
        if name != '_meta' and kernel_substr in name:
            try:
                kib = 2 * ctr['FETCH_SIZE']['mean'] + ctr['WRITE_SIZE']['mean']
            except KeyError:
                continue
            return int(kib * 1024), rel
    return None, f'{rel} has no kernel matching {kernel_substr!r}'


def ll_roofline(ctx, rng, N, M, K, reps, traffic=None, traffic_src=None):
    """One cells x clusters x mutations evaluation for K clusters, timed live
    with HIP events on the library's stream: the sums kernel alone
    (`launch_ms`, what the rocprof stats average) and the whole device-side
    evaluation (`eval_ms`: element tables + sums + combine)."""
    theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    ctx.ll_theta(0, theta, 0.01, 0.2, fetch=False)
    ctx.sync()
    kernel, _, chunks = ctx.last_launch()
    ctx.bench_ll(2)
    ms = ctx.bench_ll(reps)
    ctx.ll_theta(0, theta, 0.01, 0.2, fetch=False)
    ctx.sync()
    ctx.bench_ll_full(2)
    ms_full = ctx.bench_ll_full(reps)
    alg_bytes = N * M / 4 + 4 * K * M + 8 * N * K
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    evals = N * K / (ms * 1e-3)
    adds = evals * M            # algorithmic FP64 adds/s: one per element
    return {
        'kernel': kernel, 'bound': 'valu',
        # N K M FP64 adds per launch against the
        # FP64 vector rate (39.3e12 adds/s = half the 78.6 TFLOP/s FMA peak)
        'achieved': round(adds / 1e12, 4), 'peak': FP64_ADDS_PEAK / 1e12,
        'unit': 'TFLOP/s', 'frac': round(adds / FP64_ADDS_PEAK, 4),
        'traffic': traffic, 'traffic_source': traffic_src,
        'note': 'FP64 adds, one per cell x cluster x mutation (arithmetic '
            'intensity ~4K adds/byte: VALU-bound for K >= 2, SURVEY.md 8(d));'
            ' the exec-mask formulation ISSUES two masked v_add_f64 per '
            'element, so the issue-limited ceiling is 0.5 (`issue`); the HBM '
            'roof is in `hbm`',
        'shape': {'N': N, 'M': M, 'K': K, 'mutation_chunks': chunks},
        'launch_ms': round(ms, 5),
        'eval_ms': round(ms_full, 5),
        'algorithmic_flops': int(N) * int(K) * int(M),
        'algorithmic_bytes': int(alg_bytes),
        'issue': {
            'issued_adds_per_element': 2,
            'frac_of_issue_peak': round(2 * adds / FP64_ADDS_PEAK, 4),
        },
        'hbm': {
            'achieved': round(gbs, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(gbs / HBM_PEAK_GBS, 5),
        },
    }, evals


def spawn_ranks(n, argv):
    """`bench.py --gpus N` without a launcher: N rank processes of this very
    command (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run
    sets them), started BEFORE anything in this process has touched a GPU -
    the parent never does.  Rank 0 prints the line on the stdout it inherits;
    the parent waits, ends the others if one fails, and returns the first
    non-zero exit code (libs/MCMC.py:100-120: the reference's pool of
    chains)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r),
            WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
            MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + list(argv),
            env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:          # the others wait in a barrier
                    q.terminate()
        time.sleep(0.05)
    return rc


def dry_run(ranks, args):
    """--dry: the rank harness alone (rendezvous, barriers, max over ranks,
    the gathered identities) around steps that only sleep - no chain, no
    device.  What the CPU tests run as a COMMAND."""
    if args.dry_fail_rank == ranks.rank:
        os._exit(7)             # (the test of the spawner's failure path)
    elapsed = timed_steps(ranks, lambda i: time.sleep(0.002), 1, args.steps)
    devices, distinct = devices_of(ranks, args.steps)
    if ranks.rank == 0:
        print(json.dumps({'dry': True, 'ranks': ranks.world,
            'gpus_asked': args.gpus,
            'n_gpus': distinct if distinct is not None else ranks.world,
            'steps': args.steps,
            'value': round(ranks.world * args.steps / elapsed, 3),
            'devices': devices, 'host_settings': ranks.host_settings}))
    ranks.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='c3', choices=sorted(CONFIGS))
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--cpu-steps', type=int, default=12,
        help='oracle steps timed for cpu_baseline (0 = skip)')
    ap.add_argument('--cpu-seconds', type=float, default=60.,
        help='... but no further step once this much CPU time is spent (at '
            'least one step is made: a warm-up of 0 puts the first sweep, '
            'minutes on the CPU, into the leg)')
    ap.add_argument('--kernel-reps', type=int, default=5)
    ap.add_argument('--sustained-steps', type=int, default=200,
        help='steps run AFTER the timed window for the line\'s `sustained` '
            'rate (0 = skip); the window itself is untouched')
    ap.add_argument('--device-steps', type=int, default=50,
        help='steps run after those with per-launch device timers on, for '
            '`window.device_ms_per_step` (0 = skip)')
    ap.add_argument('--dry', action='store_true',
        help='the rank harness only: no chain, no device (CPU tests)')
    ap.add_argument('--dry-fail-rank', type=int, default=-1,
        help=argparse.SUPPRESS)
    args = ap.parse_args()

    # --gpus N means N ranks.  Under a launcher (WORLD_SIZE set) the two must
    # agree; without one this process starts the ranks itself, before any GPU
    # call, and only relays their exit code.
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is not None and int(env_world) != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: the '
            'launcher\'s rank count and --gpus must agree', file=sys.stderr)
        sys.exit(2)
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    ranks = Ranks().init()
    rank, world = ranks.rank, ranks.world
    os.environ['BNPC_DEVICE'] = str(ranks.device)
    if args.dry:
        return dry_run(ranks, args)

    import libs.CRP as dev_fixed
    import libs.CRP_learning_errors as dev_learn

    N, M, C, miss, learned = CONFIGS[args.config]
    data = synth(0, N, M, C, miss)

    # chain seeds as the reference derives them (libs/MCMC.py:102-104)
    np.random.seed(args.seed)
    seeds = np.random.randint(0, 2 ** 32 - 1, world)
    np.random.seed(seeds[rank])

    model = make_model(dev_fixed, dev_learn, data, learned)
    model.init()
    K0 = len(model.cells_per_cluster)
    total = args.warmup + args.steps
    chain = new_chain(model, learned, total, args.config)
    burn = int(total * .33)

    # ---- warm-up (untimed): includes the first sweep from K0 clusters ----
    first_step_s = None
    for i in range(1, args.warmup + 1):
        t0 = time.perf_counter()
        step(chain, i, burn)
        if i == 1:
            first_step_s = time.perf_counter() - t0
    K_warm = len(model.cells_per_cluster)
    ranks.ctx = model._dev()

    # snapshot of the post-warm-up state for the CPU baseline
    snap = None
    if rank == 0 and world == 1 and args.cpu_steps > 0:
        snap = dict(assignment=model.assignment.copy(),
            parameters=model.parameters.copy(),
            cells_per_cluster=dict(model.cells_per_cluster),
            DP_a=model.DP_a, FP=model.FP, FN=model.FN,
            rng=np.random.get_state())

    # ---- timed region: exactly K steps -----------------------------------
    import gc
    gc.collect()
    clock = MoveClock(model)
    clock.on = True
    # (the host.* counters of the line are the WINDOW's: deltas against this)
    stats0 = dict(model.host_stats())
    screen0 = model._dev().mh_screen_stats()
    cpu0 = time.process_time()      # all threads of this process
    elapsed = timed_steps(ranks, lambda i: step(chain, i, burn),
        args.warmup + 1, total)
    proc_cpu_s = time.process_time() - cpu0
    clock.on = False
    K_end = len(model.cells_per_cluster)
    ml_end = float(chain.results['ML'][total])
    stats1 = dict(model.host_stats())
    screen1 = model._dev().mh_screen_stats()
    window = clock.report(args.steps, elapsed)      # (this rank's phases)
    devices, distinct_gpus = devices_of(ranks, args.steps)   # (collective)

    # ---- after the window: what the chain sustains, and its device time ---
    # The driver's window may be short (20 steps: 8 ms at config 3) and its
    # rate moves with the moves it happened to draw; the reference's own
    # metric is a whole-run average (libs/dpmmIO.py:310-315).  The chain walks
    # on: `sustained` = the same timing over --sustained-steps more steps.
    sustained = device_time = None
    at = total
    if args.sustained_steps > 0:
        chain.add_slots(args.sustained_steps + args.device_steps + 1)
        own_window = ranks.own_s
        clock_s = MoveClock(model)      # (this leg's own phase clocks)
        clock_s.on = True
        stats_s0 = dict(model.host_stats())
        s_el = timed_steps(ranks, lambda i: step(chain, i, False), at + 1,
            at + args.sustained_steps)
        clock_s.on = False
        ranks.own_s = own_window
        at += args.sustained_steps
        sustained = {'steps': args.sustained_steps,
            'steps_s': round(world * args.sustained_steps / s_el, 3),
            'ms_per_step': round(1e3 * s_el / args.sustained_steps, 4),
            'K_end': len(model.cells_per_cluster),
            # what these steps were made of (every one of them is recorded
            # with its parameter rows: none is burn-in)
            'phases': clock_s.report(args.sustained_steps, s_el),
            # (steps / moves made as one native call, cells swept / taken in
            # runs, walkers started / adopted: deltas over this leg)
            'host': {key: val - stats_s0.get(key, 0)
                for key, val in model.host_stats().items()
                if key in ('native_steps', 'native_moves', 'swept',
                    'stride_used', 'ahead_begun', 'ahead_taken')}}
    if rank == 0 and args.device_steps > 0:
        # per-launch timers (the start / stop timestamps of every kernel
        # dispatch, as rocprofv3's kernel trace reads them): their sum over
        # these steps is the device's busy time
        if args.sustained_steps <= 0:
            chain.add_slots(args.device_steps + 1)
        dctx = model._dev()
        dctx.launch_timers(True)
        t0 = time.perf_counter()
        for i in range(at + 1, at + args.device_steps + 1):
            step(chain, i, False)
        dctx.sync()
        wall = time.perf_counter() - t0
        dev_ms, launches = dctx.launch_timers(False)
        at += args.device_steps
        device_time = {'steps': args.device_steps,
            'device_ms_per_step': round(dev_ms / args.device_steps, 4),
            'launches_per_step': round(launches / args.device_steps, 2),
            'ms_per_step_with_timers': round(1e3 * wall / args.device_steps,
                4)}

    # ---- roofline of the dominant kernel, measured live -------------------
    roofline = roofline_converged = None
    extra = {}
    if rank == 0:
        ctx = model._dev()
        rng = np.random.RandomState(1)
        # the first-sweep shape: the launch that dominates the device time of
        # a chain (not inside the timed region, which is the converged regime)
        roofline, evals = ll_roofline(ctx, rng, N, M, K0, args.kernel_reps)
        # (the committed PMC passes of this config and this build, if any)
        roofline['traffic'], roofline['traffic_source'] = \
            load_pmc_traffic(roofline['kernel'], args.config)
        extra['ll_evals_per_s_K0'] = evals
        # the launch that IS inside the timed region: K_end clusters
        roofline_converged, _ = ll_roofline(ctx, rng, N, M, K_end,
            max(20, args.kernel_reps))
        roofline_converged['note'] = (
            'the evaluation inside the timed region (K = K_end): a few '
            'microseconds of work, latency-bound; eval_ms = element tables + '
            'sums + combine; traffic: mean per launch of the split sums '
            'kernel + the combine pass over the bench run of the PMC passes')
        names = ['k_ll8_asm<2, true>']
        if 'combine' in roofline_converged['kernel']:
            names.append('k_ll_combine')
        parts = [load_pmc_traffic(name, args.config) for name in names]
        if all(p[0] is not None for p in parts):
            roofline_converged['traffic'] = sum(p[0] for p in parts)
            roofline_converged['traffic_source'] = parts[0][1]
        else:
            roofline_converged['traffic_source'] = \
                [p[1] for p in parts if p[0] is None][0]
        for Kc in sorted({10, 64, K_end}):
            r, ev = ll_roofline(ctx, rng, N, M, Kc, max(20, args.kernel_reps))
            extra[f'll_evals_per_s_K{Kc}'] = ev
            extra[f'll_launch_us_K{Kc}'] = round(r['launch_ms'] * 1e3, 2)
            extra[f'll_eval_us_K{Kc}'] = round(r['eval_ms'] * 1e3, 2)

    # ---- CPU baseline: the oracle from the same state, 1 core -------------
    cpu = parity = None
    if snap is not None:
        from oracle import crp_numpy as O
        om = make_model(O, O, data, learned)
        om.assignment = snap['assignment'].copy()
        om.parameters = snap['parameters'].copy()
        om.cells_per_cluster = dict(snap['cells_per_cluster'])
        om.DP_a, om.FP, om.FN = snap['DP_a'], snap['FP'], snap['FN']
        om.init_DP_prior()
        np.random.set_state(snap['rng'])
        ochain = new_chain(om, learned, args.cpu_steps, args.config)
        t0 = time.perf_counter()
        cpu_done = 0
        for i in range(1, args.cpu_steps + 1):
            step(ochain, i, 0)
            cpu_done = i
            if time.perf_counter() - t0 > args.cpu_seconds:
                break
        cpu_s = time.perf_counter() - t0
        # the parity gate of this line: the oracle
        # walked its steps from the GPU chain's post-warm-up snapshot on the
        # same stream, so they ARE the GPU chain's first timed steps -
        # identical assignments, log-likelihoods to 1e-9 - or the line is
        # not printed
        n_chk = min(cpu_done, args.steps)
        gpu_res, cpu_res = chain.results, ochain.results
        same = all(np.array_equal(
            gpu_res['assignments'][args.warmup + i],
            cpu_res['assignments'][i]) for i in range(1, n_chk + 1))
        g_ml = np.asarray(gpu_res['ML'][args.warmup + 1:
            args.warmup + n_chk + 1], dtype=np.float64)
        c_ml = np.asarray(cpu_res['ML'][1:n_chk + 1], dtype=np.float64)
        ml_rel = float(np.max(np.abs(g_ml / c_ml - 1))) if n_chk else 0.0
        parity = {'steps': n_chk, 'assignments_identical': bool(same),
            'ml_max_rel': ml_rel, 'tolerance': 1e-9,
            'against': 'oracle/crp_numpy.py from the same snapshot and '
                'stream position'}
        if not same or not ml_rel <= 1e-9:
            print(json.dumps({'parity_check': parity, 'error':
                'the GPU chain and the CPU oracle part ways'}),
                file=sys.stderr)
            ranks.close()
            sys.exit(3)
        # kernel-level: _calc_ll of a few cells against K0 clusters
        theta = np.clip(np.random.RandomState(1).uniform(size=(K0, M)),
            1e-5, 1 - 1e-5).astype(np.float32)
        t0 = time.perf_counter()
        ncell = 4
        for r in range(ncell):
            om._calc_ll(data[[r]], theta)
        kt = time.perf_counter() - t0
        import scipy
        cpu = {
            'value': round(cpu_done / cpu_s, 4), 'unit': 'steps/s',
            'cores': 1, 'kind': 'port',
            'sample': f'{cpu_done} MCMC steps (do_step+update_results) '
                f'of the NumPy oracle from the GPU chain\'s post-warm-up '
                f'state (K={K_warm}), {cpu_s:.1f} s',
            'll_evals_per_s_K0': ncell * K0 / kt,
            'host_cpus': os.cpu_count(),
            'numpy': np.__version__, 'scipy': scipy.__version__,
        }

    if rank == 0:
        from bnpc_amd import _lib, model as pmodel
        seen, kept = screen1[0] - screen0[0], screen1[1] - screen0[1]
        stats = {key: stats1[key] - stats0.get(key, 0) for key in stats1}
        host_info = {
            'threads': _lib.host_threads(),
            'threads_wide_batches': _lib.threads_for(K_end * M),
            'native_mh_batch': pmodel._native_kernels() is not None,
            'native_beta': pmodel._native_beta(),
            # the private interfaces / start-up comparisons the fast paths
            # rest on (False = the documented fallback ran; all of them forced
            # off: BNPC_STREAM_LIVE=0 BNPC_NATIVE_MH=0 BNPC_NATIVE_BETA=0)
            'fast_paths': pmodel.fast_paths(),
            'numa_node': getattr(model._dev(), 'numa_node', None),
            'cpus': len(os.sched_getaffinity(0)),
            # CPU time of ALL threads of the chain's process per timed step
            # (the team's spinning included) / per wall ms: what a chain
            # costs the host it shares with the other chains of a node
            'cpu_ms_per_step': round(1e3 * proc_cpu_s / args.steps, 3),
            'cpu_busy_threads': round(proc_cpu_s / elapsed, 2),
            # (counters of the TIMED WINDOW: deltas over its steps)
            # cells of its sweeps / decided from the device's hint without a
            # scan / of those, between the row's two best columns
            'sweep_cells': stats['swept'],
            'sweep_hinted': stats['hint_used'],
            'sweep_pairs': stats['pair_used'],
            'sweep_triples': stats['triple_used'],
            # ... of sweep_hinted, decided in the loop's lane for such cells
            'sweep_lane': stats['lane_used'],
            # ... of sweep_lane, taken whole runs at a time (cells that stay in
            # the cluster that dominates them: the lane's stride)
            'sweep_stride': stats.get('stride_used', 0),
            # steps made as ONE native call (bnpc_chain_step) / split-merge
            # moves made as one native call (bnpc_sm_move)
            'native_steps': stats['native_steps'],
            'native_moves': stats['native_moves'],
            # parameter batches whose draws a walker on the aside thread took
            # ahead on a copy of the stream: started / adopted by the batch
            # (the live stream stood where the walker's did) / rows adopted
            'mh_ahead': [stats.get('ahead_begun', 0),
                stats.get('ahead_taken', 0), stats.get('ahead_rows', 0)],
            # parameter-batch entries screened on the device / share of them
            # the host still had to evaluate (accepted or in doubt)
            'mh_screened': seen,
            'mh_left_to_host': round(kept / seen, 4) if seen else None,
        }
        value = world * args.steps / elapsed
        if device_time is not None:
            # device busy time per step (measured on the steps right after
            # the sustained leg, timers on) and the share of a step it is
            per = (sustained or {}).get('ms_per_step') \
                or 1e3 * elapsed / args.steps
            window.update(device_time)
            window['device_busy_frac'] = round(
                device_time['device_ms_per_step'] / per, 4)
        line = {
            'metric': ('MCMC steps/s, 5k cells x 1k muts' if args.config == 'c3'
                else f'MCMC steps/s, {N} cells x {M} muts')
                + ' (+ cell x cluster log-lik evals/s in ll_evals_per_s_K0)',
            'value': round(value, 3), 'unit': 'steps/s',
            # the number of DISTINCT GPUs the ranks ran on (PCI bus ids): ranks
            # that share one GPU count once
            'n_gpus': distinct_gpus if distinct_gpus is not None else world,
            'ranks': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': f'{args.config}: synthetic {N} cells x {M} muts, '
                    f'{int(miss * 100)}% missing, '
                    f'{"learned" if learned else "fixed"} errors, '
                    '1 chain per GPU, '
                    + ('moves ' + str(MOVE_OVERRIDES[args.config])
                        if args.config in MOVE_OVERRIDES
                        else 'CLI-default moves'),
                'chains': world, 'data_seed': 0, 'mcmc_seed': args.seed,
                'K0': K0, 'K_after_warmup': K_warm, 'K_end': K_end,
            },
            'devices': devices,
            'per_rank_steps_s': [d['steps_s'] for d in devices],
            'ranks_share_a_gpu': distinct_gpus is not None
                and distinct_gpus < world,
            'host': host_info,
            'first_step_s': None if first_step_s is None
                else round(first_step_s, 4),
            # what the timed window was made of (rank 0): a window's value
            # depends on its move mix - split/merge steps cost ~3x a Gibbs step
            'window': window,
            'sustained': sustained,
            'ML_end': ml_end,
            'roofline': roofline,
            'roofline_converged': roofline_converged,
            'cpu_baseline': cpu,
            'parity_check': parity,
        }
        line.update(extra)
        if cpu:
            line['speedup_vs_cpu_baseline'] = round(value / cpu['value'], 2)
        print(json.dumps(line))
    ranks.close()


if __name__ == '__main__':
    main()
