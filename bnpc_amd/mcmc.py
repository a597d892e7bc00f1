"""Sampler driver: chain -> GPU placement, the per-step move schedule, traces.

The caller side of the hot path.  What the driver has to do is fixed by the
reference (/root/reference/libs/MCMC.py: MCMC :26-193, Chain :200-342,
Chain_steps :349-388, Chain_time :395-440), because the model classes - the
GPU-backed ``bnpc_amd.model`` and the CPU oracle alike - must see the
reference's call sequence and the global legacy ``np.random`` stream must be
consumed in the reference's order:

    step := [u<sm_prob ? split_merge : Gibbs] -> [u<dpa_prob ? DP alpha]
            -> update_parameters -> [learning & u<error_prob ? error rates]
            -> record (ML, MAP, alpha, FN, FP, assignment, params)

How it is organised is this build's own:

  * `TraceStore`   the per-chain sample storage (the `results` dict every
                   downstream consumer reads), pre-allocated and grown in
                   blocks;
  * `Tally`        acceptance counters of the MH moves, by name;
  * `advance()`    ONE step of the move schedule, a free function of
                   (model, knobs, tally);
  * `StepBudget` / `Deadline`   when a chain stops and which steps are
                   burn-in;
  * `Chain`        glues the four; `Chain_steps` / `Chain_time` are the two
                   constructors the reference's names promise;
  * `MCMC`         seeds, the fork pool (worker i -> GPU i mod #GPUs: one chain
                   per GPU, no collective - SURVEY.md section 8(e)), lugsail
                   extension rounds.

Deliberate differences from the reference, all additive: worker exceptions are
re-raised in the parent (the reference drops them, MCMC.py:113-120);
error-rate learning is detected by the model's surface as well as by its
module path (MCMC.py:206-209), so the CPU oracle can be driven too.
"""
from copy import deepcopy
from datetime import datetime
import multiprocessing as mp
import os

import numpy as np

np.seterr(divide='raise', over='ignore', under='ignore', invalid='raise')

KFD_NODES = '/sys/class/kfd/kfd/topology/nodes'


# ---------------------------------------------------------------- placement
def _node_has_simds(node):
    try:
        with open(os.path.join(KFD_NODES, node, 'properties')) as f:
            for line in f:
                if line.startswith('simd_count'):
                    return int(line.split()[1]) > 0
    except OSError:
        pass
    return False


def _listed(name):
    """Entries of a *_VISIBLE_DEVICES variable, or None if it is not set.
    An empty value hides every device, as the runtime reads it."""
    value = os.environ.get(name)
    if value is None:
        return None
    return [e for e in value.replace(' ', '').split(',') if e != '']


def _visible_gpus():
    """Number of GPUs chains may be spread over = the device ordinals
    0..n-1 the HIP runtime will offer a worker.  No HIP call happens here
    (the parent must stay fork-safe), so the count is read the way the
    runtime reads it: BNPC_NUM_DEVICES if given; else the length of
    HIP_VISIBLE_DEVICES (or CUDA_VISIBLE_DEVICES, its alias), which indexes
    into what ROCR_VISIBLE_DEVICES leaves; else the length of
    ROCR_VISIBLE_DEVICES; else the GPU nodes the kernel driver lists."""
    forced = os.environ.get('BNPC_NUM_DEVICES')
    if forced:
        return max(1, int(forced))
    try:
        physical = sum(_node_has_simds(d) for d in os.listdir(KFD_NODES))
    except OSError:
        physical = 0
    rocr = _listed('ROCR_VISIBLE_DEVICES')
    if rocr is not None:
        physical = len(rocr)
    for name in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        hip = _listed(name)
        if hip is not None:
            # ordinals beyond what ROCR leaves visible are ignored by HIP
            return max(1, min(len(hip), physical) if physical else len(hip))
    return max(1, physical)


def device_for_chain(chain_index, n_devices=None):
    """Chain i runs on device ordinal i mod #visible (one chain per GPU while
    there are GPUs left, round-robin beyond)."""
    return chain_index % (n_devices or _visible_gpus())


def _bind_worker_to_gpu(chain_index, n_devices=None, n_chains=1):
    """Called in the worker before its model creates a device context.  The
    assignment is explicit: a BNPC_DEVICE inherited from the parent's
    environment does not pin every chain to one GPU.  BNPC_HOST_SHARE tells
    the worker how many chains end up on the CPUs of its NUMA node (each is
    bound to the node of its GPU): the default host thread team is a share of
    that node (_lib.host_threads), and whether the chain spins and wakes ranks
    like a chain alone goes by the logical CPUs that leaves it
    (_lib.host_settings)."""
    from bnpc_amd import _lib
    n_devices = n_devices or _visible_gpus()
    os.environ['BNPC_DEVICE'] = str(device_for_chain(chain_index, n_devices))
    # (a chain left with fewer than _lib.GREEDY_MIN_CPUS logical CPUs gives
    # idle ranks back quickly and wakes few: 8 chains x 4 ranks on one node's
    # 16 CPUs each, round 3: 2990 steps/s spinning 50 us, 3170 spinning 5 us)
    _lib.host_settings(n_chains, n_devices)


# ------------------------------------------------------------------- traces
class TraceStore:
    """Samples of one chain, in the layout of the reference's
    `Chain.results` (MCMC.py:231-305) because posterior inference, PSRF and
    the writers index it by these keys:

        ML, MAP, DP_alpha, FN, FP   float64 (slot 0 = state after init)
        assignments                 int (slots x cells)
        params                      float32 (post-burn-in slots x clusters x
                                    mutations), clusters in ascending-id
                                    order, zero-padded to the largest cluster
                                    count seen; created at the first
                                    post-burn-in sample
        burn_in, PSRF, ...          scalars added by the owners
    """

    SCALARS = ('ML', 'MAP', 'DP_alpha', 'FN', 'FP')
    GROW_BLOCK = 200

    @staticmethod
    def _zeros(shape, dtype=float):
        """np.zeros whose pages EXIST: the large traces (labels: 8 N bytes
        per step; parameter rows: 4 K M) come from fresh anonymous memory, and
        the first write to every 4 KiB page is a fault and a page to clear -
        a megabyte and a half per recorded step at config 5: ~0.1 ms of the
        step's recording.  Paid here, once, by the host team (one thread:
        6-16 ms for config 5's 99 MB of parameter rows, inside the step that
        takes the first post-burn-in sample)."""
        if len(shape) < 2 or int(np.prod(shape)) * np.dtype(dtype).itemsize \
                < 1 << 20:
            return np.zeros(shape, dtype=dtype)
        from bnpc_amd._lib import rows_copy_zero
        return rows_copy_zero(np.empty(shape, dtype=dtype))

    @staticmethod
    def _widened(old, shape):
        """`old` with rows appended (axis 0) and / or every row zero-padded
        (axis 1) to `shape` - np.append / np.pad of the reference
        (MCMC.py:275-279, 289-294) - into fresh memory, on the host team."""
        from bnpc_amd._lib import rows_copy_zero
        old = np.ascontiguousarray(old)
        new = np.empty(shape, dtype=old.dtype)
        n = old.shape[0]
        rows_copy_zero(new[:n], old)
        if shape[0] > n:
            rows_copy_zero(new[n:])
        return new

    def __init__(self, slots, n_cells, n_muts):
        self.n_cells = n_cells
        self.n_muts = n_muts
        self.data = {key: np.zeros(slots) for key in self.SCALARS}
        self.data['assignments'] = self._zeros((slots, n_cells), dtype=int)

    @property
    def slots(self):
        return self.data['ML'].size

    def grow(self, extra=None, with_params=False):
        """Append `extra` empty slots (default: a block, at most doubling)."""
        extra = extra or min(self.GROW_BLOCK, self.slots)
        d = self.data
        if with_params and 'params' in d:
            old = d['params']
            d['params'] = self._widened(old,
                (old.shape[0] + extra,) + old.shape[1:])
        for key in self.SCALARS:
            d[key] = np.append(d[key], np.zeros(extra))
        d['assignments'] = np.append(d['assignments'],
            np.zeros((extra, self.n_cells), int), axis=0)

    def put_state(self, slot, model):
        """a8: the total log-likelihood of every step is what the 1e-6 parity
        criterion is checked on (MCMC.py:252-258)."""
        d = self.data
        deferred = getattr(model, 'get_ll_full_deferred', None)
        if deferred is None:
            log_lik = model.get_ll_full()
            log_prior = model.get_lprior_full()
        else:       # the prior is host work: it runs under the launch
            pick_up = deferred()
            log_prior = model.get_lprior_full()
            log_lik = pick_up()
        d['ML'][slot] = log_lik
        d['MAP'][slot] = log_lik + log_prior
        d['DP_alpha'][slot] = model.DP_a
        d['FN'][slot] = model.FN
        d['FP'][slot] = model.FP
        d['assignments'][slot] = model.assignment

    def record_target(self, slot, with_params):
        """Where a natively made step (CRP.native_step) writes the state it
        records: (addresses of the ML / MAP / DP_alpha / FN / FP slots,
        address of the assignment row, address of the parameter block of the
        slot or 0 - the block does not exist before the first post-burn-in
        sample -, its capacity in clusters)."""
        d = self.data
        params = d.get('params')
        # (the base addresses and the checks of the arrays' layouts, made
        # once per SET of arrays: they are replaced when a trace grows or the
        # parameter block is re-padded, never changed in place)
        key = tuple(id(d[k]) for k in self.SCALARS) \
            + (id(d['assignments']), id(params))
        cache = self.__dict__.get('_targets')
        if cache is None or cache[0] != key:
            from bnpc_amd._lib import ptr
            labels = d['assignments']
            labels_ok = labels.dtype == np.int64 \
                and labels.flags['C_CONTIGUOUS']
            params_at = 0
            if params is not None and params.dtype == np.float32 \
                    and params.flags['C_CONTIGUOUS']:
                params_at = ptr(params)
            cache = self._targets = (key, [ptr(d[k]) for k in self.SCALARS],
                ptr(labels) if labels_ok else 0, params_at)
        _, bases, labels_at, params_at = cache
        if not labels_at:
            return None
        block, cap = 0, 0
        if with_params and params_at:
            row = slot - (self.slots - params.shape[0])
            if 0 <= row < params.shape[0]:
                cap = params.shape[1]
                block = params_at + 4 * row * cap * self.n_muts
        return [b + 8 * slot for b in bases], \
            labels_at + 8 * slot * self.n_cells, block, cap

    def put_params(self, slot, model):
        """MCMC.py:260-282: parameter rows of the populated clusters, zero
        padded to the largest cluster count seen.  The array is allocated a
        few clusters wider than that (re-padding the whole trace for every
        new maximum costs a millisecond a time) and cut back to the
        reference's shape when the chain hands its results over (`finish`)."""
        d = self.data
        live = np.sort(np.fromiter(model.cells_per_cluster.keys(), dtype=int))
        if 'params' not in d:
            self._k_seen = 0
            d['params'] = self._zeros((self.slots - slot,
                live.size + self.PARAMS_SPARE, self.n_muts), dtype=np.float32)
        wider = live.size - d['params'].shape[1]
        if wider > 0:
            old = d['params']
            d['params'] = self._widened(old, (old.shape[0],
                old.shape[1] + wider + self.PARAMS_SPARE, self.n_muts))
        self._k_seen = max(getattr(self, '_k_seen', 0), live.size)
        row = slot - (self.slots - d['params'].shape[0])
        d['params'][row][:live.size] = model.parameters[live]

    PARAMS_SPARE = 8

    def finish(self):
        """Results in the reference's layout: the parameter trace exactly as
        wide as the largest cluster count seen."""
        k = getattr(self, '_k_seen', None)
        if k is not None and 'params' in self.data \
                and self.data['params'].shape[1] != k:
            self.data['params'] = np.ascontiguousarray(
                self.data['params'][:, :k])

    def __getstate__(self):
        """Chains return from their workers through a pipe: cluster labels
        travel in the narrowest integer type that holds them."""
        state = self.__dict__.copy()
        state.pop('_targets', None)     # (addresses of this process's arrays)
        data = dict(state['data'])
        labels = data.get('assignments')
        if isinstance(labels, np.ndarray) and labels.dtype.kind == 'i' \
                and labels.size:
            for narrow in (np.int16, np.int32):
                if labels.min() >= np.iinfo(narrow).min \
                        and labels.max() <= np.iinfo(narrow).max:
                    data['assignments'] = labels.astype(narrow)
                    state['_label_dtype'] = labels.dtype.str
                    break
        state['data'] = data
        return state

    def __setstate__(self, state):
        wide = state.pop('_label_dtype', None)
        self.__dict__.update(state)
        if wide is not None:
            self.data['assignments'] = \
                self.data['assignments'].astype(np.dtype(wide))

    def drop_unused_tail(self):
        """Slots never written still hold MAP == 0 (MCMC.py:431-436)."""
        unused = int((self.data['MAP'] == 0).sum())
        if unused:
            self.data = {key: val[:-unused] for key, val in self.data.items()}


class Tally:
    """[accepted, declined] per MH move since the last report."""

    MOVES = ('parameters', 'splits', 'merges', 'FP', 'FN')

    def __init__(self):
        self.reset()

    def reset(self):
        self.counts = {name: np.zeros(2) for name in self.MOVES}

    def add(self, name, accepted_declined):
        # (element by element: `array += tuple` converts the tuple first -
        # 1 us a time, three times a step)
        c = self.counts[name]
        c[0] += accepted_declined[0]
        c[1] += accepted_declined[1]

    def report(self, fix_assign, learning_errors):
        from bnpc_amd.io import show_MH_acceptance
        show_MH_acceptance(self.counts['parameters'], 'parameters', 1)
        names = [] if fix_assign else ['splits', 'merges']
        names += ['FP', 'FN'] if learning_errors else []
        for name in names:
            show_MH_acceptance(self.counts[name], name, 2)
        self.reset()


# ------------------------------------------------------------ move schedule
def advance(model, knobs, tally, fix_assign=False, learning_errors=False):
    """One step of the sampler (MCMC.py:320-342).  Each `np.random.random()`
    is a draw of the reference's stream at the reference's position; the
    error-rate draw exists only for models that learn their error rates, the
    first two only when assignments are free."""
    if not fix_assign:
        if np.random.random() < knobs['sm_prob']:
            outcome, move = model.update_assignments_split_merge(
                knobs['sm_ratios'], knobs['sm_steps'])
            tally.add('splits' if move == 0 else 'merges', outcome)
        else:
            model.update_assignments_Gibbs()
        if np.random.random() < knobs['dpa_prob']:
            model.update_DP_alpha()

    declined, accepted = model.update_parameters()
    tally.add('parameters', (accepted, declined))

    if learning_errors and np.random.random() < knobs['error_prob']:
        FP_outcome, FN_outcome = model.update_error_rates()
        tally.add('FP', FP_outcome)
        tally.add('FN', FN_outcome)


def _learns_errors(model):
    return type(model).__module__ == 'libs.CRP_learning_errors' \
        or callable(getattr(model, 'update_error_rates', None))


def _before(value, limit):
    """`value < limit`; False when the two cannot be compared (a burn-in that
    was invalidated)."""
    try:
        return bool(value < limit)
    except TypeError:
        return False


# ------------------------------------------------------------------ budgets
class StepBudget:
    """`count` steps, the first `burn_in` of them warm-up."""

    def __init__(self, count, burn_in):
        self.count = count
        self.burn_in = burn_in

    def initial_slots(self):
        return self.count + 1

    def first_sample_is_warmup(self):
        return self.burn_in != 0


class Deadline:
    """Run until `end_time`; samples before `burn_in` (a time) are warm-up."""

    def __init__(self, end_time, burn_in):
        self.end_time = end_time
        self.burn_in = burn_in

    def initial_slots(self):
        return 500

    def first_sample_is_warmup(self):
        return True


# -------------------------------------------------------------------- chain
class Chain:
    """One MCMC chain = model + move knobs + budget + traces."""

    def __init__(self, model, knobs, no, budget, verbosity=1,
                fix_assign=False):
        self.model = model
        self.mcmc = knobs
        self.no = no
        self.budget = budget
        self.burn_in = budget.burn_in
        self.verbosity = verbosity
        self.fix_assign = fix_assign
        self.learning_errors = _learns_errors(model)
        self.tally = Tally()
        self.trace = TraceStore(budget.initial_slots(), model.cells_total,
            model.muts_total)
        self.update_results(0, budget.first_sample_is_warmup())

    def __str__(self):
        return f'Chain: {self.no:0>2d}'

    @property
    def results(self):
        return self.trace.data

    def get_result(self):
        return self.trace.data

    def get_steps(self):
        return self.trace.slots

    def do_step(self):
        native = getattr(self.model, 'native_step', None)
        if native is not None:
            done = native(self.mcmc, self.fix_assign, self.learning_errors)
            if done is not None:
                self._tally_native(done)
                return
        advance(self.model, self.mcmc, self.tally, self.fix_assign,
            self.learning_errors)

    def _slot(self, step, burn_in):
        trace = self.trace
        if step == trace.slots:     # open-ended runs outgrow their traces
            try:
                trace.grow(with_params=not burn_in)
            except MemoryError:     # wrap around; nothing is burn-in any more
                step %= trace.slots
                self.burn_in = np.nan
        return step

    def update_results(self, step, burn_in=True):
        """Record the state after `step` (MCMC.py:242-282)."""
        trace = self.trace
        step = self._slot(step, burn_in)
        trace.put_state(step, self.model)
        if not burn_in:
            trace.put_params(step, self.model)

    def _tally_native(self, done):
        tally = self.tally
        if done['sm'] is not None:
            tally.add('splits' if done['move'] == 'split' else 'merges',
                done['sm'])
        declined, accepted = done['parameters']
        tally.add('parameters', (accepted, declined))
        if done['errors'] is not None:
            tally.add('FP', done['errors'][0])
            tally.add('FN', done['errors'][1])

    def step(self, step, burn_in=True):
        """do_step + update_results(step, burn_in): ONE native call when the
        model offers it (CRP.native_step: the moves, the updates and the
        recorded likelihood / prior without the interpreter in between), else
        the two calls."""
        native = getattr(self.model, 'native_step', None)
        if native is None:
            self.do_step()
            self.update_results(step, burn_in)
            return
        trace = self.trace
        slot = self._slot(step, burn_in)
        target = trace.record_target(slot, not burn_in)
        done = native(self.mcmc, self.fix_assign, self.learning_errors,
            target) if target is not None else None
        if done is None:
            self.do_step()
            self.update_results(slot, burn_in)
            return
        self._tally_native(done)
        if not done['recorded']:
            trace.put_state(slot, self.model)
        if not burn_in:
            if done.get('params_recorded'):
                trace._k_seen = max(getattr(trace, '_k_seen', 0),
                    len(self.model.cells_per_cluster))
            else:
                trace.put_params(slot, self.model)

    def _report(self, headline):
        print(f'\t{self}\tstep:\t{headline}\n\t\tmean MH accept. ratio:')
        self.tally.report(self.fix_assign, self.learning_errors)


class Chain_steps(Chain):
    """A chain with a fixed number of steps (MCMC.py:349-388)."""

    def __init__(self, model, no, steps, burn_in, mcmc, verbosity=1,
                fix_assign=False):
        super().__init__(model, mcmc, no, StepBudget(steps, burn_in),
            verbosity, fix_assign)

    def set_steps(self, n):
        self.budget.count = n

    def add_slots(self, n):
        self.trace.grow(n, with_params=True)

    def run(self, init_steps=0):
        todo = self.budget.count
        # reference quirk kept: fewer than 9 steps make the report interval
        # zero and the modulo raises ZeroDivisionError (MCMC.py:378)
        interval = (todo + 1) // 10
        for step in range(1, todo + 1):
            if step % interval == 0 and self.verbosity > 1:
                self._report(f'{step + init_steps: >3} / {todo + init_steps}')
            self.step(step + init_steps, _before(step, self.burn_in))
        self.trace.finish()
        self.trace.data['burn_in'] = self.burn_in


class Chain_time(Chain):
    """A chain that runs until a wall-clock deadline (MCMC.py:395-440)."""

    def __init__(self, model, no, end_time, burn_in, mcmc, verbosity=1,
                fix_assign=False):
        super().__init__(model, mcmc, no, Deadline(end_time, burn_in),
            verbosity, fix_assign)

    def run(self):
        end = self.budget.end_time
        step = 0
        now = datetime.now()
        while now <= end:
            if step % 1000 == 0 and self.verbosity > 1:
                left = (end - now).seconds / 60
                self._report(f'{step: >3}\t(remaining: {left:.1f} mins.)')
            step += 1
            self.step(step, _before(now, self.burn_in))
            now = datetime.now()
        self.trace.drop_unused_tail()
        self.trace.finish()
        kept = self.trace.data['params'].shape[0]
        self.trace.data['burn_in'] = self.trace.slots - kept


# ------------------------------------------------------------------- driver
class MCMC:
    """Seeds and runs the chains (MCMC.py:26-193)."""

    PROPOSAL_SD = (0.1, 0.25, 0.5)
    LUGSAIL_ROUND = 200

    def __init__(self, model, sm_prob=0.33, dpa_prob=0.5, error_prob=0.1,
                sm_ratios=[0.75, 0.25], sm_steps=5):
        self.model = model
        self.chains = []
        self.seeds = []
        self.params = dict(sm_prob=sm_prob, dpa_prob=dpa_prob,
            error_prob=error_prob, sm_ratios=sm_ratios, sm_steps=sm_steps,
            param_proposal_sd=np.array(self.PROPOSAL_SD))

    def __str__(self):
        p = self.params
        return ''.join([
            'Move probabilitites:\n',
            f'\tSplit/merge:\t{p["sm_prob"]}\n',
            f'\t\tsplit/merge ratio:\t{p["sm_ratios"]}\n',
            f'\t\tintermediate Gibbs:\t{p["sm_steps"]}\n',
            f'\tCRP a_0 update:\t{p["dpa_prob"]}\n',
            f'\tErrors update:\t{p["error_prob"]}\n'])

    def get_seeds(self):
        return self.seeds

    def get_results(self):
        out = [chain.get_result() for chain in self.chains]
        if not out or 'burn_in' not in out[0]:
            raise RuntimeError('Error in sampling from MCMC')
        return out

    @staticmethod
    def _plan(run_var):
        """Termination modes (MCMC.py:79-94): (int steps, burn-in steps),
        (float PSRF cutoff, _) or (end time, burn-in time)."""
        head = run_var[0]
        if isinstance(head, (int, np.integer)):
            return Chain_steps, run_var, None
        if isinstance(head, float):
            first_round = max(10, int(1 / (head ** 2 - 1)))
            return Chain_steps, (first_round, 0), head
        return Chain_time, run_var, None

    def run(self, run_var, seed, n=1, verbosity=1, assign_file='', debug=False):
        chain_type, run_var, cutoff = self._plan(run_var)
        assign = None
        if assign_file:
            from bnpc_amd.io import load_txt
            assign = load_txt(assign_file)

        workers = min(n, mp.cpu_count())
        # master seed -> one seed per chain (MCMC.py:100-104)
        if seed > 0:
            np.random.seed(seed)
        self.seeds = np.random.randint(0, 2 ** 32 - 1, workers)

        if debug:       # single chain in this process
            np.random.seed(self.seeds[0])
            print(f'\nSeed set to: {self.seeds[0]}\n')
            self.chains.append(
                self.run_chain(chain_type, run_var, assign, 0, 2))
            return

        quiet = 0 if cutoff else verbosity
        jobs = [(chain_type, run_var, assign, i, quiet)
            for i in range(workers)]
        self._fan_out(self.run_chain, jobs, self.chains.append)
        self.chains.sort(key=lambda chain: chain.no)
        if cutoff:
            self.run_lugsail_chains(cutoff, workers, verbosity)

    @staticmethod
    def _fan_out(fn, jobs, on_done):
        """One worker PROCESS per job (the reference's pool has as many
        workers as chains, MCMC.py:100-120).  Workers are forked, so the model
        and the chains reach them without a pickle; results come back through
        POSIX shared memory (bnpc_amd.handoff).  A worker that raises, or dies
        without reporting, fails the run here after all have finished."""
        from bnpc_amd import handoff
        failures = handoff.run_jobs(fn, jobs, on_done)
        if failures:
            raise RuntimeError(f'chain worker failed: {failures[0]}')

    def run_chain(self, chain_type, run_var, assign, i, verbosity):
        """Worker body (MCMC.py:126-135): seed, private model copy, init, run."""
        _bind_worker_to_gpu(i, n_chains=len(self.seeds))
        np.random.seed(self.seeds[i])
        model = deepcopy(self.model)
        model.init(assign=assign)
        chain = chain_type(model, i + 1, *run_var, self.params, verbosity,
            isinstance(assign, list))
        chain.run()
        return chain

    def run_lugsail_chains(self, cutoff, cores, verbosity, n=None):
        """Extend all chains in rounds of `n` steps until the lugsail PSRF
        estimate passes `cutoff` (MCMC.py:138-177)."""
        from bnpc_amd.postproc import get_lugsail_batch_means_est
        n = n or self.LUGSAIL_ROUND
        done = self.chains[0].results['ML'].size
        while True:
            PSRF = get_lugsail_batch_means_est(
                [(c.results['ML'], done // 2) for c in self.chains])
            if verbosity > 1:
                print(f'\tPSRF at {done}:\t{PSRF:.5f}')
            for chain in self.chains:
                chain.results.setdefault('PSRF', []).append((done, PSRF))
            if PSRF <= cutoff:
                break
            try:
                self._fan_out(self.extend_chain,
                    [(i, n) for i in range(cores)], self.replace_chain)
            except KeyboardInterrupt:
                print('Manual termination')
                break
            done += n

        burn_in = (done // 2) + 1
        for chain in self.chains:
            res = chain.results
            res['burn_in'] = burn_in
            res['params'] = res['params'][burn_in:]
            res['PSRF_cutoff'] = cutoff

    def extend_chain(self, chain_no, add_steps):
        """MCMC.py:180-189; the worker re-seeds with the chain's own seed, as
        the reference does."""
        _bind_worker_to_gpu(chain_no, n_chains=len(self.seeds))
        np.random.seed(self.seeds[chain_no])
        chain = self.chains[chain_no]
        already = chain.get_steps()
        chain.add_slots(add_steps)
        chain.set_steps(add_steps)
        chain.run(init_steps=already - 1)
        return chain_no, chain

    def replace_chain(self, numbered_chain):
        chain_no, chain = numbered_chain
        self.chains[chain_no] = chain
