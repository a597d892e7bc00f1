"""Sampler driver: chain pool, per-step move schedule, trace storage.

The caller side of the hot path.  Behaviour follows the reference driver
/root/reference/libs/MCMC.py (MCMC :26-193, Chain :200-342, Chain_steps
:349-388, Chain_time :395-440) so that the model classes (GPU-backed
``bnpc_amd.model`` or the CPU oracle) see exactly the reference's call
sequence and the global legacy ``np.random`` stream is consumed in the
reference's order:

    step := [u<sm_prob ? split_merge : Gibbs] -> [u<dpa_prob ? DP alpha]
            -> update_parameters -> [learning & u<error_prob ? error rates]
            -> update_results (ML, MAP, alpha, FN, FP, assignment, params)

Differences from the reference, all additive:
  * worker exceptions are re-raised in the parent (the reference drops them,
    MCMC.py:113-120),
  * a chain index -> GPU mapping: worker i exports BNPC_DEVICE=i mod #GPUs
    before the model creates its device context (one chain per GPU, no
    collective - SURVEY.md section 8(e)),
  * `learning errors` is detected by the model's surface, not only by its
    module path (MCMC.py:206-209), so the CPU oracle can be driven too.
"""
from copy import deepcopy
from datetime import datetime
import multiprocessing as mp
import os

import numpy as np

np.seterr(divide='raise', over='ignore', under='ignore', invalid='raise')


def _visible_gpus():
    """Number of GPUs to spread chains over (no HIP call: fork-safe)."""
    env = os.environ.get('BNPC_NUM_DEVICES')
    if env:
        return max(1, int(env))
    try:
        n = len([d for d in os.listdir('/sys/class/kfd/kfd/topology/nodes')
            if _is_gpu_node(d)])
        return max(1, n)
    except OSError:
        return 1


def _is_gpu_node(node):
    try:
        with open(f'/sys/class/kfd/kfd/topology/nodes/{node}/properties') as f:
            for line in f:
                if line.startswith('simd_count'):
                    return int(line.split()[1]) > 0
    except OSError:
        pass
    return False


class MCMC:
    """MCMC.py:26-193"""

    def __init__(self, model, sm_prob=0.33, dpa_prob=0.5, error_prob=0.1,
                sm_ratios=[0.75, 0.25], sm_steps=5):
        self.model = model
        self.chains = []
        self.seeds = []
        self.params = {
            'sm_prob': sm_prob,
            'dpa_prob': dpa_prob,
            'error_prob': error_prob,
            'param_proposal_sd': np.array([0.1, 0.25, 0.5]),
            'sm_ratios': sm_ratios,
            'sm_steps': sm_steps,
        }

    def __str__(self):
        return ('Move probabilitites:\n'
            '\tSplit/merge:\t{sm_prob}\n\t\tsplit/merge ratio:\t{sm_ratios}\n'
            '\t\tintermediate Gibbs:\t{sm_steps}\n'
            '\tCRP a_0 update:\t{dpa_prob}\n'
            '\tErrors update:\t{error_prob}\n').format(**self.params)

    def get_results(self):
        results = [chain.get_result() for chain in self.chains]
        if not results or 'burn_in' not in results[0]:
            raise RuntimeError('Error in sampling from MCMC')
        return results

    def get_seeds(self):
        return self.seeds

    def run(self, run_var, seed, n=1, verbosity=1, assign_file='', debug=False):
        """MCMC.py:79-123"""
        cutoff = None
        if isinstance(run_var[0], (int, np.integer)):
            chain_type = Chain_steps
        elif isinstance(run_var[0], float):
            chain_type = Chain_steps
            cutoff = run_var[0]
            run_var = (max(10, int(1 / (cutoff ** 2 - 1))), 0)
            verbosity_ls = verbosity
            verbosity = 0
        else:
            chain_type = Chain_time

        if assign_file:
            from bnpc_amd.io import load_txt
            assign = load_txt(assign_file)
        else:
            assign = None

        cores = min(n, mp.cpu_count())
        if seed > 0:
            np.random.seed(seed)
        self.seeds = np.random.randint(0, 2 ** 32 - 1, cores)

        if debug:
            np.random.seed(self.seeds[0])
            print(f'\nSeed set to: {self.seeds[0]}\n')
            self.chains.append(
                self.run_chain(chain_type, run_var, assign, 0, 2))
            return

        self._pool_map(self.run_chain,
            [(chain_type, run_var, assign, i, verbosity)
                for i in range(cores)], self.chains.append)
        self.chains.sort(key=lambda c: c.no)

        if cutoff:
            self.run_lugsail_chains(cutoff, cores, verbosity_ls)

    @staticmethod
    def _pool_map(fn, arg_list, callback):
        errors = []
        pool = mp.get_context('fork').Pool(len(arg_list))
        for args in arg_list:
            pool.apply_async(fn, args, callback=callback,
                error_callback=errors.append)
        pool.close()
        pool.join()
        if errors:
            raise RuntimeError(f'chain worker failed: {errors[0]!r}') \
                from errors[0]

    def run_chain(self, chain_type, run_var, assign, i, verbosity):
        """MCMC.py:126-135 (+ chain -> GPU mapping)"""
        os.environ.setdefault('BNPC_DEVICE', str(i % _visible_gpus()))
        np.random.seed(self.seeds[i])
        model = deepcopy(self.model)
        model.init(assign=assign)
        chain = chain_type(model, i + 1, *run_var, self.params, verbosity,
            isinstance(assign, list))
        chain.run()
        return chain

    def run_lugsail_chains(self, cutoff, cores, verbosity, n=200):
        """MCMC.py:138-177"""
        from bnpc_amd.postproc import get_lugsail_batch_means_est
        steps_run = self.chains[0].results['ML'].size
        while True:
            PSRF = get_lugsail_batch_means_est(
                [(c.results['ML'], steps_run // 2) for c in self.chains])
            if verbosity > 1:
                print(f'\tPSRF at {steps_run}:\t{PSRF:.5f}')
            for chain in self.chains:
                chain.results.setdefault('PSRF', []).append((steps_run, PSRF))
            if PSRF <= cutoff:
                break
            try:
                self._pool_map(self.extend_chain,
                    [(i, n) for i in range(cores)], self.replace_chain)
            except KeyboardInterrupt:
                print('Manual termination')
                break
            steps_run += n

        burn_in = (steps_run // 2) + 1
        for chain in self.chains:
            chain.results['burn_in'] = burn_in
            chain.results['params'] = chain.results['params'][burn_in:]
            chain.results['PSRF_cutoff'] = cutoff

    def extend_chain(self, chain_no, add_steps):
        """MCMC.py:180-189 (re-seeds with the chain's seed, as the reference)"""
        os.environ.setdefault('BNPC_DEVICE', str(chain_no % _visible_gpus()))
        np.random.seed(self.seeds[chain_no])
        chain = self.chains[chain_no]
        old_steps = chain.get_steps()
        chain._extend_results(add_steps, False)
        chain.set_steps(add_steps)
        chain.run(init_steps=old_steps - 1)
        return chain_no, chain

    def replace_chain(self, new_chain):
        self.chains[new_chain[0]] = new_chain[1]


class Chain:
    """One MCMC chain: the move schedule and its trace store.

    Follows libs/MCMC.py:200-342.  `results` keeps the reference's layout
    because everything downstream reads it: float traces ML, MAP, DP_alpha,
    FN, FP (entry 0 = initial state), `assignments` (samples x cells),
    after burn-in `params` (samples x clusters x mutations float32, clusters in
    sorted-id order, zero-padded to the largest cluster count seen), and
    `burn_in`.
    """

    SCALAR_TRACES = ('ML', 'MAP', 'DP_alpha', 'FN', 'FP')
    # rows of MH_counter (cols: accepted, declined)
    ROW_PARAMS, ROW_SPLIT, ROW_MERGE, ROW_FP, ROW_FN = range(5)

    def __init__(self, model, mcmc, no, verbosity=1, fix_assign=False):
        self.model = model
        self.mcmc = mcmc
        self.no = no
        # the reference keys this on the module path (MCMC.py:206-209)
        self.learning_errors = \
            type(model).__module__ == 'libs.CRP_learning_errors' \
            or callable(getattr(model, 'update_error_rates', None))
        self.results = {}
        self.MH_counter = np.zeros((5, 2))
        self.verbosity = verbosity
        self.fix_assign = fix_assign

    def __str__(self):
        return f'Chain: {self.no:0>2d}'

    def get_result(self):
        return self.results

    def run(self, *args):
        pass

    # ------------------------------------------------------------ traces
    def init_results(self, steps):
        for key in self.SCALAR_TRACES:
            self.results[key] = np.zeros(steps)
        self.results['assignments'] = np.zeros(
            (steps, self.model.cells_total), dtype=int)

    def _capacity(self):
        return self.results['ML'].size

    def update_results(self, step, burn_in=True):
        """Record the state after `step` (MCMC.py:242-282)."""
        free_slots = self._capacity() - step
        if free_slots == 0:         # time-limited runs grow their traces
            try:
                self._extend_results(burn_in=burn_in)
            except MemoryError:
                step %= self._capacity()
                self.burn_in = np.nan
        self._record_state(step)
        if not burn_in:
            self._record_parameters(step, free_slots)

    def _record_state(self, step):
        model, res = self.model, self.results
        log_lik = model.get_ll_full()
        res['ML'][step] = log_lik
        res['MAP'][step] = log_lik + model.get_lprior_full()
        res['DP_alpha'][step] = model.DP_a
        res['FN'][step] = model.FN
        res['FP'][step] = model.FP
        res['assignments'][step] = model.assignment

    def _record_parameters(self, step, free_slots):
        model, res = self.model, self.results
        live = np.sort(np.fromiter(model.cells_per_cluster.keys(), dtype=int))
        if 'params' not in res:
            res['params'] = np.zeros(
                (free_slots, live.size, model.muts_total), dtype=np.float32)
        trace = res['params']
        if live.size > trace.shape[1]:
            trace = res['params'] = np.pad(trace,
                [(0, 0), (0, live.size - trace.shape[1]), (0, 0)],
                mode='constant')
        first_kept = self._capacity() - trace.shape[0] + 1
        trace[step - first_kept + 1][:live.size] = model.parameters[live]

    def _extend_results(self, add_size=None, burn_in=True):
        """MCMC.py:285-305"""
        res = self.results
        extra = add_size or min(200, self._capacity())
        if not burn_in:
            res['params'] = np.append(res['params'], np.zeros(
                (extra, res['params'].shape[1], self.model.muts_total)),
                axis=0)
        for key in self.SCALAR_TRACES:
            res[key] = np.append(res[key], np.zeros(extra))
        res['assignments'] = np.append(res['assignments'],
            np.zeros((extra, self.model.cells_total), int), axis=0)

    # ------------------------------------------------------------ reporting
    def stdout_progress(self):
        from bnpc_amd.io import show_MH_acceptance
        rows = [(self.ROW_PARAMS, 'parameters', 1)]
        if not self.fix_assign:
            rows += [(self.ROW_SPLIT, 'splits', 2), (self.ROW_MERGE, 'merges', 2)]
        if self.learning_errors:
            rows += [(self.ROW_FP, 'FP', 2), (self.ROW_FN, 'FN', 2)]
        for row, name, tabs in rows:
            show_MH_acceptance(self.MH_counter[row], name, tabs)
        self.MH_counter = np.zeros((5, 2))

    # ------------------------------------------------------------ one step
    def do_step(self):
        """The move schedule (MCMC.py:320-342).  Every `np.random.random()`
        below is a draw of the reference's stream, in its order; the error
        draw only happens for models that learn their error rates."""
        model, prob = self.model, self.mcmc
        if not self.fix_assign:
            if np.random.random() < prob['sm_prob']:
                outcome, move = model.update_assignments_split_merge(
                    prob['sm_ratios'], prob['sm_steps'])
                row = self.ROW_SPLIT if move == 0 else self.ROW_MERGE
                self.MH_counter[row] += outcome
            else:
                model.update_assignments_Gibbs()
            if np.random.random() < prob['dpa_prob']:
                model.update_DP_alpha()

        declined, accepted = model.update_parameters()
        self.MH_counter[self.ROW_PARAMS] += (accepted, declined)

        if self.learning_errors \
                and np.random.random() < prob['error_prob']:
            FP_outcome, FN_outcome = model.update_error_rates()
            self.MH_counter[self.ROW_FP] += FP_outcome
            self.MH_counter[self.ROW_FN] += FN_outcome


class Chain_steps(Chain):
    """A chain that runs a fixed number of steps (MCMC.py:349-388)."""

    def __init__(self, model, no, steps, burn_in, mcmc, verbosity=1,
                fix_assign=False):
        super().__init__(model, mcmc, no, verbosity, fix_assign)
        self.steps = steps + 1
        self.burn_in = burn_in
        self.init_results(steps + 1)
        self.update_results(0, burn_in != 0)

    def set_steps(self, n):
        self.steps = n + 1

    def get_steps(self):
        return self._capacity()

    def stdout_progress(self, step_no, total):
        print(f'\t{self}\tstep:\t{step_no: >3} / {total - 1}\n'
            '\t\tmean MH accept. ratio:')
        super().stdout_progress()

    def _in_burn_in(self, step):
        try:
            return step < self.burn_in
        except TypeError:
            return False

    def run(self, init_steps=0):
        # reference quirk kept: with fewer than 9 steps the report interval
        # is zero and the modulo below raises ZeroDivisionError
        report_every = self.steps // 10
        last = self.steps + init_steps
        for step in range(1, self.steps):
            if step % report_every == 0 and self.verbosity > 1:
                self.stdout_progress(step + init_steps, last)
            self.do_step()
            self.update_results(step + init_steps, self._in_burn_in(step))
        self.results['burn_in'] = self.burn_in


class Chain_time(Chain):
    """A chain that runs until a wall-clock deadline (MCMC.py:395-440)."""

    def __init__(self, model, no, end_time, burn_in, mcmc, verbosity=1,
                fix_assign=False):
        super().__init__(model, mcmc, no, verbosity, fix_assign)
        self.end_time = end_time
        self.burn_in = burn_in
        self.init_results(500)
        self.update_results(0)

    def stdout_progress(self, step_no, total):
        print(f'\t{self}\tstep:\t{step_no: >3}\t(remaining: {total:.1f} mins.)'
            '\n\t\tmean MH accept. ratio:')
        super().stdout_progress()

    def run(self):
        step = 0
        while True:
            now = datetime.now()
            if now > self.end_time:
                break
            if step % 1000 == 0 and self.verbosity > 1:
                self.stdout_progress(step, (self.end_time - now).seconds / 60)
            step += 1
            self.do_step()
            try:
                warming_up = now < self.burn_in
            except TypeError:
                warming_up = False
            self.update_results(step, warming_up)

        # drop the unused tail of the pre-allocated traces
        unused = int((self.results['MAP'] == 0).sum())
        if unused:
            self.results = {key: values[:-unused]
                for key, values in self.results.items()}
        self.results['burn_in'] = self._capacity() \
            - self.results['params'].shape[0]
