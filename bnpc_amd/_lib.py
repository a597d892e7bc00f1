"""ctypes binding of include/bnpc_hip.h (libbnpc_hip.so).

The product path has NO CPU fallback: if the shared library is missing or a
HIP call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# BNPC_LIB: another build of the same library (the sanitizer builds of
# bnpc_amd/build.py); never a different implementation
LIB_PATH = os.environ.get('BNPC_LIB') or os.path.join(_PKG, 'libbnpc_hip.so')

# os.environ.get costs 0.5 us a call (key and value are encoded / decoded every
# time) and a step asks for a dozen switches: read the process environment's
# own byte-keyed table where CPython exposes it (still live: a switch changed
# through os.environ is seen at once), else os.environ
_environ_raw = getattr(os.environ, '_data', None)
if not isinstance(_environ_raw, dict) \
        or (_environ_raw and not isinstance(next(iter(_environ_raw)), bytes)):
    _environ_raw = None


def env(name, default=None):
    """os.environ.get(name, default), cheaply."""
    if _environ_raw is None:
        return os.environ.get(name, default)
    val = _environ_raw.get(name.encode() if isinstance(name, str) else name)
    return default if val is None else os.fsdecode(val)


MAX_VIEWS = 7
TILE_SLOTS = 4      # include/bnpc_hip.h: BNPC_TILE_SLOTS
MAX_TRIALS = 4
HINT_COLS_MAX = 32767   # columns of a hinted sweep (bnpc_top2 holds them as int16)

_i64 = C.c_int64
# array arguments travel as plain addresses (building a typed ctypes pointer
# per argument costs more than some of the calls)
_pd = _pf = _pi64 = _pi32 = C.c_void_p
_host_pd = C.POINTER(C.c_double)            # double* results (pinned buffers)
_ppd = C.POINTER(_host_pd)
_ctx = C.c_void_p


class MT19937(C.Structure):
    _fields_ = [('key', C.c_uint32 * 624), ('pos', C.c_int32)]


# bnpc_top2 (include/bnpc_hip.h) as a NumPy record
TOP2 = np.dtype([('best', np.float64), ('second', np.float64),
    ('third', np.float32), ('fourth', np.float32), ('e2', np.float32),
    ('e3', np.float32), ('ll_best', np.float64),
    ('ll_second', np.float64), ('ll_third', np.float64), ('col', np.int16),
    ('col2', np.int16), ('col3', np.int16), ('row_here', np.int16)])
assert TOP2.itemsize == 64


class GibbsState(C.Structure):
    _fields_ = [('n_cells', _i64), ('ld', _i64), ('n_cols', _i64),
        ('n_active', _i64), ('pos', _i64), ('new_cell', _i64),
        ('pos_end', _i64), ('row_base', _i64), ('threads', _i64),
        ('hint', C.c_void_p), ('hint_prior', C.c_void_p), ('hint_cols', _i64),
        ('hint_used', _i64), ('matrix_wait', C.c_void_p),
        ('matrix_wait_arg', C.c_void_p), ('pair_used', _i64),
        ('birth_ctx', C.c_void_p), ('birth_view', C.c_int32),
        ('birth_put', C.c_int32), ('birth_rows', _i64),
        ('theta_host', C.c_void_p), ('beta_p', C.c_double),
        ('beta_q', C.c_double), ('tmin', C.c_double), ('tmax', C.c_double),
        ('FP', C.c_double), ('FN', C.c_double), ('gauss', C.c_void_p),
        ('born', C.c_void_p), ('born_cap', _i64), ('n_born', _i64),
        ('triple_used', _i64), ('hint_in_order', _i64), ('lane_used', _i64),
        ('stride_used', _i64)]


class MoveState(C.Structure):
    """bnpc_move_state (include/bnpc_hip.h)"""
    _fields_ = [('move', C.c_int32), ('scan_no', C.c_int32),
        ('view', C.c_int32), ('uniform_prior', C.c_int32),
        ('threads', C.c_int32), ('threads_wide', C.c_int32), ('K', _i64),
        ('ids', C.c_void_p), ('sizes', C.c_void_p), ('N', _i64), ('M', _i64),
        ('assignment', C.c_void_p), ('parameters', C.c_void_p),
        ('param_stride', _i64), ('DP_a', C.c_double), ('sd', C.c_void_p),
        ('n_sd', _i64), ('FP', C.c_double), ('FN', C.c_double),
        ('p', C.c_double), ('q', C.c_double), ('tmin', C.c_double),
        ('tmax', C.c_double), ('fill', C.c_double), ('gauss', C.c_void_p),
        ('accepted', C.c_int32), ('pad_', C.c_int32), ('cl_i', _i64),
        ('cl_j', _i64), ('moved', _i64), ('n_cells', _i64),
        ('log_A', C.c_double)]


class MHArgs(C.Structure):
    """bnpc_mh_args (include/bnpc_hip.h)"""
    _fields_ = [('G', _i64), ('M', _i64), ('old_theta', C.c_void_p),
        ('n1', C.c_void_p), ('n0', C.c_void_p), ('sd', C.c_void_p),
        ('n_sd', _i64), ('tmin', C.c_double), ('tmax', C.c_double),
        ('FP', C.c_double), ('FN', C.c_double), ('p', C.c_double),
        ('q', C.c_double), ('uniform_prior', C.c_int),
        ('trans_prob', C.c_int), ('known_theta', C.c_void_p),
        ('known_prior', C.c_void_p), ('sd_idx', C.c_void_p),
        ('U', C.c_void_p), ('u', C.c_void_p), ('new_theta', C.c_void_p),
        ('prior_out', C.c_void_p), ('A', C.c_void_p),
        ('log_prob', C.c_void_p), ('declined', C.c_void_p),
        ('threads', C.c_int), ('screen', C.c_void_p),
        ('screen_theta', C.c_void_p), ('flagged_estimate', _i64),
        ('flag_counts', C.c_void_p), ('prior_seq_sum', C.c_void_p)]


class LogAArgs(C.Structure):
    """bnpc_accept_args (include/bnpc_hip.h)"""
    _fields_ = [('G', _i64), ('M', _i64), ('new_theta', C.c_void_p),
        ('old_theta', C.c_void_p), ('std', C.c_void_p), ('n1', C.c_void_p),
        ('n0', C.c_void_p), ('fmin', C.c_double), ('fmax', C.c_double),
        ('tmin', C.c_double), ('tmax', C.c_double), ('FP', C.c_double),
        ('FN', C.c_double), ('p', C.c_double), ('q', C.c_double),
        ('uniform_prior', C.c_int), ('clip', C.c_int), ('A', C.c_void_p),
        ('sum', C.c_void_p), ('threads', C.c_int)]


class ChainState(C.Structure):
    """bnpc_chain (include/bnpc_hip.h): the model and the knobs of one chain
    as bnpc_chain_step reads and updates them"""
    _fields_ = [('N', _i64), ('M', _i64), ('assignment', C.c_void_p),
        ('parameters', C.c_void_p), ('param_stride', _i64),
        ('ids', C.c_void_p), ('sizes', C.c_void_p), ('K', _i64),
        ('crp_prior', C.c_void_p), ('DP_a', C.c_double),
        ('dpa_shape', C.c_double), ('dpa_rate', C.c_double),
        ('FP', C.c_double), ('FN', C.c_double), ('p', C.c_double),
        ('q', C.c_double), ('tmin', C.c_double), ('tmax', C.c_double),
        ('mix0', C.c_double), ('mix1', C.c_double),
        ('uniform_prior', C.c_int32), ('learning', C.c_int32),
        ('sd', C.c_void_p), ('n_sd', _i64), ('FP_prior', C.c_double * 4),
        ('FN_prior', C.c_double * 4), ('FP_sd', C.c_double * 3),
        ('FN_sd', C.c_double * 3), ('sm_prob', C.c_double),
        ('dpa_prob', C.c_double), ('error_prob', C.c_double),
        ('sm_ratios', C.c_double * 2), ('sm_steps', C.c_int32),
        ('fix_assign', C.c_int32), ('threads', C.c_int32),
        ('threads_wide', C.c_int32), ('wide_from', _i64),
        ('sweep_bytes', _i64), ('view_move', C.c_int32),
        ('sweep_hint', C.c_int32), ('gauss', C.c_void_p),
        ('phase', C.c_int32), ('need', C.c_int32),
        ('rec_scalars', C.c_void_p * 5), ('rec_assignment', C.c_void_p),
        ('rec_params', C.c_void_p), ('rec_params_cap', _i64),
        ('move', C.c_int32), ('sm_accepted', C.c_int32), ('sm_cells', _i64),
        ('alpha_updated', C.c_int32), ('errors_updated', C.c_int32),
        ('FP_accepted', C.c_int32), ('FN_accepted', C.c_int32),
        ('par_declined', _i64), ('par_accepted', _i64),
        ('rec_params_done', C.c_int32), ('pad_', C.c_int32),
        ('ML', C.c_double), ('lprior', C.c_double), ('swept', _i64),
        ('hint_used', _i64), ('pair_used', _i64), ('triple_used', _i64),
        ('native_moves', _i64), ('steps', _i64), ('lane_used', _i64),
        ('stride_used', _i64), ('clock_ns', _i64 * 10),
        ('clock_calls', _i64 * 10), ('work', C.c_void_p)]


PHASE_ASSIGN, PHASE_ALPHA, PHASE_PARAMS, PHASE_ERRORS, PHASE_RECORD = range(5)
NEED_NONE, NEED_MOVE, NEED_GIBBS, NEED_PARAMS, NEED_ERRORS, NEED_RECORD = \
    range(6)
STEP_CLOCKS = ('gibbs', 'split_accepted', 'split_rejected', 'merge_accepted',
    'merge_rejected', 'dp_alpha', 'parameters', 'error_rates', 'record',
    'gibbs_waits_for_device')


# the version bnpc_abi_version() of a matching library reports (bumped with
# every change of a structure or signature of include/bnpc_hip.h)
ABI_VERSION = 12

# name -> (restype, argtypes); must list every symbol of include/bnpc_hip.h
SIGNATURES = {
    'bnpc_last_error': (C.c_char_p, []),
    'bnpc_abi_version': (C.c_int, []),
    'bnpc_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'bnpc_device_info': (C.c_int, [C.c_int, C.c_char_p, C.c_int,
        C.POINTER(C.c_int)]),
    'bnpc_device_pci_bus_id': (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    'bnpc_team_stress': (C.c_int, [_i64, C.c_int, C.c_int, C.c_uint64,
        C.POINTER(_i64), C.POINTER(_i64)]),
    'bnpc_team_size': (C.c_int, [C.c_int]),
    'bnpc_rows_copy_zero': (C.c_int, [C.c_void_p, _i64, C.c_void_p, _i64, _i64,
        _i64, _i64, C.c_int]),
    'bnpc_create': (C.c_int, [C.c_int, _i64, _i64, _pd, C.POINTER(_ctx)]),
    'bnpc_create_codes': (C.c_int, [C.c_int, _i64, _i64,
        C.c_void_p, C.POINTER(_ctx)]),
    'bnpc_create_planes': (C.c_int, [C.c_int, _i64, _i64, C.c_void_p,
        C.POINTER(_ctx)]),
    'bnpc_pack_codes': (C.c_int, [C.c_void_p, _i64, _i64, _i64, _i64,
        C.c_void_p]),
    'bnpc_unpack_codes': (C.c_int, [C.c_void_p, _i64, _i64, C.c_void_p, _i64,
        C.c_void_p]),
    'bnpc_destroy': (C.c_int, [_ctx]),
    'bnpc_reload_options': (C.c_int, [_ctx]),
    'bnpc_shape': (C.c_int, [_ctx, _pi64, _pi64]),
    'bnpc_cell_counts': (C.c_int, [_ctx, _pi32, _pi32]),
    'bnpc_view_set': (C.c_int, [_ctx, C.c_int, _pi64, _i64]),
    'bnpc_view_set_slot': (C.c_int, [_ctx, C.c_int, _pi64, _i64, C.c_int]),
    'bnpc_view_size': (C.c_int, [_ctx, C.c_int, _pi64]),
    'bnpc_ll_theta': (C.c_int, [_ctx, C.c_int, _pf, _i64, C.c_double,
        C.c_double, _pd, _i64]),
    'bnpc_ll_theta_pinned': (C.c_int, [_ctx, C.c_int, _pf, _i64, C.c_double,
        C.c_double, _i64, _ppd]),
    'bnpc_ll_theta_pinned_top2': (C.c_int, [_ctx, C.c_int, _pf, _i64,
        C.c_double, C.c_double, _i64, _pd, _ppd, C.POINTER(C.c_void_p)]),
    'bnpc_ll_theta_pinned_top2_issue': (C.c_int, [_ctx, C.c_int, _pf, _i64,
        C.c_double, C.c_double, _i64, _pd, _ppd, C.POINTER(C.c_void_p)]),
    'bnpc_matrix_wait': (C.c_int, [_ctx]),
    'bnpc_hints_wait': (C.c_int, [_ctx]),
    'bnpc_ll_theta_pinned_sums_issue': (C.c_int, [_ctx, C.c_int, _pf, _i64,
        C.c_double, C.c_double, _i64, _pd, _ppd]),
    'bnpc_hints_in_order_issue': (C.c_int, [_ctx, _pi64,
        C.POINTER(C.c_void_p)]),
    'bnpc_theta_put': (C.c_int, [_ctx, _i64, _pf, _i64]),
    'bnpc_ll_rows_pinned': (C.c_int, [_ctx, C.c_int, _pi64, _i64, C.c_double,
        C.c_double, _i64, _ppd]),
    'bnpc_ll_rows_issue': (C.c_int, [_ctx, C.c_int, _pi64, _i64, C.c_double,
        C.c_double, _i64, C.c_int]),
    'bnpc_ll_rows_wait': (C.c_int, [_ctx, C.c_int, _ppd]),
    'bnpc_ll_rows_issue_hint': (C.c_int, [_ctx, C.c_int, _pi64, _i64,
        C.c_double, C.c_double, _i64, C.c_int, _pd]),
    'bnpc_ll_rows_wait_hint': (C.c_int, [_ctx, C.c_int, _ppd,
        C.POINTER(C.c_void_p)]),
    'bnpc_ll_tables': (C.c_int, [_ctx, C.c_int, _pd, _pd, _i64, _pd, _i64]),
    'bnpc_colcounts': (C.c_int, [_ctx, _pi64, _pi64, _i64, _pi32, _pi32]),
    'bnpc_view_counts': (C.c_int, [_ctx, C.c_int, _pi64, _i64, _pi32, _pi32]),
    'bnpc_colcounts_by_label': (C.c_int, [_ctx, _pi64, _pi64, _i64, _pi32,
        _pi32]),
    'bnpc_ll_total': (C.c_int, [_ctx, _pf, _i64, _pd, _pd, C.c_int, _pd]),
    'bnpc_ll_total_issue': (C.c_int, [_ctx, _pf, _i64, _pd, _pd, C.c_int]),
    'bnpc_ll_total_wait': (C.c_int, [_ctx, _pd]),
    'bnpc_bench_ll': (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_float)]),
    'bnpc_bench_ll_full': (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_float)]),
    'bnpc_launch_timers': (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_double),
        C.POINTER(C.c_int64)]),
    'bnpc_mh_ahead_stats': (C.c_int, [_ctx, C.POINTER(C.c_int64),
        C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'bnpc_last_launch': (C.c_int, [_ctx, C.c_char_p, C.c_int,
        C.POINTER(_i64), C.POINTER(C.c_int)]),
    'bnpc_timer_start': (C.c_int, [_ctx]),
    'bnpc_timer_stop': (C.c_int, [_ctx, C.POINTER(C.c_float)]),
    'bnpc_sync': (C.c_int, [_ctx]),
    'bnpc_mt_random_sample': (C.c_double, [C.POINTER(MT19937)]),
    'bnpc_mt_permutation': (C.c_int, [C.POINTER(MT19937), _i64, _pi64]),
    'bnpc_mt_mh_draws': (C.c_int, [C.POINTER(MT19937), _i64, _i64, _i64,
        _pi32, _pd, _pd]),
    'bnpc_mt_beta': (C.c_int, [C.POINTER(MT19937), C.c_void_p, _i64, _pd,
        _pd, _pd]),
    'bnpc_mt_beta_theta': (C.c_int, [C.POINTER(MT19937), C.c_void_p, _i64,
        C.c_double, C.c_double, _pi32, _pi32, C.c_double, C.c_double,
        C.c_double, _pf]),
    'bnpc_log_diff_pi': (C.c_int, [_pd, _pd, _i64, _pd]),
    'bnpc_mh_batch': (C.c_int, [C.c_void_p, C.POINTER(MT19937),
        C.POINTER(MHArgs),
        C.POINTER(C.c_int)]),
    'bnpc_mh_screen': (C.c_int, [_ctx, C.c_int, C.POINTER(MHArgs),
        C.c_void_p, C.c_void_p]),
    'bnpc_mh_batch_dev': (C.c_int, [_ctx, C.c_void_p, C.POINTER(MT19937),
        C.POINTER(MHArgs), C.c_int, C.POINTER(C.c_int)]),
    'bnpc_label_counts_and_batch': (C.c_int, [_ctx, C.c_void_p,
        C.POINTER(MT19937), _pi64, _pi64, C.POINTER(MHArgs),
        C.POINTER(C.c_int)]),
    'bnpc_mh_screen_stats': (C.c_int, [_ctx, C.POINTER(_i64),
        C.POINTER(_i64)]),
    'bnpc_rg_scan_step': (C.c_int, [_ctx, C.c_void_p, C.POINTER(MT19937),
        C.c_int, _i64, _pi64, C.c_double, C.POINTER(MHArgs), _pi32, _pi32,
        C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    'bnpc_log_accept': (C.c_int, [C.c_void_p, C.POINTER(LogAArgs),
        C.POINTER(C.c_int)]),
    'bnpc_sm_move': (C.c_int, [_ctx, C.c_void_p, C.POINTER(MT19937),
        C.POINTER(MoveState), C.POINTER(C.c_int)]),
    'bnpc_move_propose': (C.c_int, [C.c_void_p, C.POINTER(MT19937),
        C.POINTER(MoveState), C.c_void_p, C.POINTER(_i64), C.POINTER(_i64),
        C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int)]),
    'bnpc_np_sum': (C.c_int, [C.c_void_p, _i64, C.POINTER(C.c_double)]),
    'bnpc_pair_pick': (C.c_int, [C.c_int, C.c_double, _i64, _i64, _i64,
        C.c_double, C.POINTER(_i64)]),
    'bnpc_two_way_pick': (C.c_int, [C.c_int, C.c_double, C.c_double,
        C.c_double, C.POINTER(_i64)]),
    'bnpc_triple_pick': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, _i64,
        C.c_double, C.POINTER(_i64)]),
    'bnpc_tn_ppf_scalar': (C.c_int, [C.c_void_p, C.c_double, C.c_double,
        C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double),
        C.POINTER(C.c_int)]),
    'bnpc_tn_logpdf_scalar': (C.c_int, [C.c_void_p, C.c_double, C.c_double,
        C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double),
        C.POINTER(C.c_int)]),
    'bnpc_beta_logpdf_f32': (C.c_int, [C.c_void_p, _pf, _i64, C.c_double,
        C.c_double, C.c_void_p, C.c_void_p, _pd, _pd, C.c_int]),
    'bnpc_dominated_cdf': (C.c_int, [_i64, _i64, _pd]),
    'bnpc_gibbs_sweep': (C.c_int, [C.POINTER(GibbsState), C.POINTER(MT19937),
        _pi64, _pd, _pd, _pd, _pi64, _pi64, _pi64, _pi64, _pi64, _pd]),
    'bnpc_parse_matrix': (C.c_int, [C.c_char_p, C.c_char, C.c_int, C.c_int,
        C.c_void_p, _pi64, _pi64]),
    'bnpc_codist': (C.c_int, [C.c_int, _pi32, _i64, _i64, _pi32]),
    'bnpc_post_create': (C.c_int, [C.c_int, _pi32, _i64, _i64,
        C.POINTER(C.c_void_p), C.POINTER(_i64)]),
    'bnpc_post_fetch': (C.c_int, [C.c_void_p, _pi32, _pd]),
    'bnpc_post_mpear': (C.c_int, [C.c_void_p, C.c_void_p, _i64, _pi64]),
    'bnpc_post_ward': (C.c_int, [C.c_void_p, _pd]),
    'bnpc_post_ward_stats': (C.c_int, [C.c_void_p, C.POINTER(_i64),
        C.POINTER(_i64)]),
    'bnpc_post_destroy': (C.c_int, [C.c_void_p]),
    'bnpc_rg_scan': (C.c_int, [C.POINTER(MT19937), C.c_int, _i64, _pd,
        C.c_double, _pi64, _pi64, _pd]),
    'bnpc_mt_gamma': (C.c_int, [C.POINTER(MT19937), C.c_void_p, C.c_double,
        C.c_double, C.POINTER(C.c_double)]),
    'bnpc_chain_open': (C.c_int, [C.POINTER(ChainState)]),
    'bnpc_chain_close': (C.c_int, [C.POINTER(ChainState)]),
    'bnpc_chain_step': (C.c_int, [_ctx, C.c_void_p, C.POINTER(MT19937),
        C.POINTER(ChainState)]),
    'bnpc_chain_update_alpha': (C.c_int, [C.c_void_p, C.POINTER(MT19937),
        C.POINTER(ChainState)]),
    'bnpc_gamma_logpdf_scalar': (C.c_int, [C.c_void_p, C.c_double, C.c_double,
        C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
}

_lib = None


def load():
    """Load libbnpc_hip.so (once).  Raises RuntimeError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: the HIP extension has not been built. '
            'Run `python -m bnpc_amd.build` (needs hipcc, no GPU). There is '
            'no CPU fallback for the product path.')
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as err:
        raise RuntimeError(f'cannot load {LIB_PATH}: {err}') from err
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    built = lib.bnpc_abi_version()
    if built != ABI_VERSION:
        raise RuntimeError(
            f'{LIB_PATH} was built for ABI version {built}, this binding '
            f'expects {ABI_VERSION} (structures of include/bnpc_hip.h have '
            'changed): run `python -m bnpc_amd.build`')
    _lib = lib
    return lib


class DeviceMemoryError(RuntimeError):
    """An entry point reports that its working set does not fit the device
    (return code 5): the one failure a caller may answer with another route."""


def check(rc, what=''):
    if rc != 0:
        msg = load().bnpc_last_error().decode('utf-8', 'replace')
        kind = DeviceMemoryError if rc == 5 else RuntimeError
        raise kind(f'libbnpc_hip {what} failed (code {rc}): {msg}')


_from_buffer = C.c_char.from_buffer
_addressof = C.addressof


def ptr(arr, ctype=None):
    """Address of a C-contiguous array (the ctype is documentation).
    `arr.ctypes.data` builds a helper object per call (0.7-1.3 us, three
    dozen times a step); the buffer protocol gives the same address in 0.2 us
    for the writable, contiguous, non-empty arrays that are the rule here."""
    try:
        return _addressof(_from_buffer(arr))
    except (TypeError, ValueError):
        return arr.ctypes.data


def as_i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# Whether THIS process has made a HIP call: a process that has must not fork
# workers that use the GPU (they would inherit a runtime they cannot use), so
# the chain driver spawns them instead (bnpc_amd.handoff).
_touched = {'pid': None}


def mark_gpu_touched():
    _touched['pid'] = os.getpid()


def gpu_touched():
    return _touched['pid'] == os.getpid()


def device_count():
    mark_gpu_touched()
    n = C.c_int(0)
    check(load().bnpc_device_count(C.byref(n)), 'device_count')
    return n.value


def device_pci_bus_id(device=0):
    """PCI bus id of a device ordinal (domain:bus:device.function): what tells
    two GPUs apart whatever the visible-device lists say."""
    mark_gpu_touched()
    buf = C.create_string_buffer(32)
    check(load().bnpc_device_pci_bus_id(int(device), buf, 32),
        'device_pci_bus_id')
    return buf.value.decode().lower()


def device_info(device=0):
    mark_gpu_touched()
    name = C.create_string_buffer(64)
    cus = C.c_int(0)
    check(load().bnpc_device_info(device, name, 64, C.byref(cus)),
        'device_info')
    return name.value.decode(), cus.value


def _cpus_of_node(node, sysfs='/sys/devices/system/node'):
    cpus = set()
    with open(os.path.join(sysfs, f'node{node}', 'cpulist')) as f:
        for part in f.read().strip().split(','):
            if part:
                lo, _, hi = part.partition('-')
                cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_node_count(sysfs='/sys/devices/system/node'):
    """NUMA nodes that have CPUs (1 if the layout cannot be read)."""
    n = 0
    try:
        for entry in os.listdir(sysfs):
            if entry.startswith('node') and entry[4:].isdigit():
                try:
                    n += bool(_cpus_of_node(int(entry[4:]), sysfs))
                except (OSError, ValueError):
                    pass
    except OSError:
        pass
    return max(1, n)


def host_share(n_chains, n_devices, nodes=None):
    """How many of `n_chains` concurrent chains, dealt round-robin over
    `n_devices` GPUs, end up on the CPUs of one NUMA node (every chain is bound
    to the node of its GPU; the GPUs are assumed spread evenly over the
    nodes): what BNPC_HOST_SHARE carries."""
    nodes = nodes or numa_node_count()
    groups = max(1, min(int(nodes), int(n_devices), int(n_chains)))
    return max(1, -(-int(n_chains) // groups))


# logical CPUs of its NUMA node a chain must have to itself to keep the
# settings of a chain alone (team spinning 300 us between the batches of a
# step, a rank per 2 blocks for the small batches: 3.4 busy threads); with
# fewer it gives idle ranks back after 5 us and wakes a rank per 12 blocks
# (1.4 busy threads).  Round 5 keyed this on "shares a node" - on the target
# (8 GPUs, 2 x 64 cores: 4 chains per node, 32 logical CPUs each) every chain
# then ran the frugal settings with 28 CPUs idle.  Round 6, ranks of bench.py
# sharing the one GPU and one node of a box (profiles/r06/
# bench_ranks_sharing_one_gpu.jsonl; rule against frugal forced): 2 ranks (64
# CPUs each) 4167 / 4171 steps/s, 4 ranks (32 each: the target's share) 8267 /
# 8172, 8 ranks (16 each) 8789 / 10 967 - the line lies between 16 and 32.
GREEDY_MIN_CPUS = 24


def node_cpus(nodes=None):
    """Logical CPUs of one NUMA node as this process may use them: the
    online CPUs divided by the nodes, capped by the affinity mask (a process
    already bound to a node, a cgroup)."""
    nodes = nodes or numa_node_count()
    total = os.cpu_count() or 1
    return max(1, min(_host_cores(), max(1, total // max(1, int(nodes)))))


def host_settings(n_chains, n_devices, nodes=None, apply=True,
            keep_env=False):
    """The host side of one of `n_chains` concurrent chains on `n_devices`
    GPUs (bnpc_amd.mcmc's workers, bench.py's ranks): how many chains share
    the CPUs of its NUMA node, how many logical CPUs that leaves it, and
    whether it keeps the settings of a chain alone (`greedy`).  apply=True
    exports them: BNPC_HOST_SHARE always (the default team size goes by it),
    BNPC_HOST_SPIN_US=5 for a frugal chain unless the user set one.
    keep_env: a BNPC_HOST_SHARE the user exported is taken as it is."""
    share = host_share(n_chains, n_devices, nodes)
    if keep_env and env('BNPC_HOST_SHARE'):
        try:
            share = max(1, int(env('BNPC_HOST_SHARE')))
        except ValueError:
            pass
    per_chain = node_cpus(nodes) // share
    greedy = share == 1 or per_chain >= GREEDY_MIN_CPUS
    out = {'share': share, 'cpus_per_chain': per_chain, 'greedy': greedy}
    if apply:
        os.environ['BNPC_HOST_SHARE'] = str(share)
        if not greedy:
            os.environ.setdefault('BNPC_HOST_SPIN_US', '5')
    out['spin_us'] = int(os.environ.get('BNPC_HOST_SPIN_US') or 300) \
        if apply else (300 if greedy else 5)
    return out


_affinity = {'pid': None, 'original': None, 'bound': 0}


def _set_affinity_all_threads(cpus):
    """The affinity mask of EVERY thread of this process (the host thread
    team may exist already: a new mask on the calling thread alone would
    leave its workers where they were)."""
    try:
        tids = [int(t) for t in os.listdir('/proc/self/task')]
    except (OSError, ValueError):
        tids = [0]
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
        except (OSError, ProcessLookupError):
            pass            # a thread that has just ended
    os.sched_setaffinity(0, cpus)


def bind_near_device(device, pci_sysfs='/sys/bus/pci/devices',
            node_sysfs='/sys/devices/system/node'):
    """Keep this process (all its threads, and the host threads it starts
    later) on the CPUs of the NUMA node the device hangs off: a chain's sweep
    walks a matrix the GPU has just written into pinned host memory, which
    the runtime places on that node; from the other socket every row is a
    remote access (measured on the 2-socket MI355X host: 200 us against
    220-300 us per sweep).  The mask the process had BEFORE its first binding
    is remembered: a later context on a GPU of another node re-binds against
    that, and release_binding() (Context.close) restores it when the last
    bound context goes.  Best effort - returns the node, or None if nothing
    was changed (a single-node host, no sysfs entry, BNPC_NUMA_BIND=0, or an
    affinity mask that does not reach the node)."""
    if os.environ.get('BNPC_NUMA_BIND', '1') == '0':
        return None
    try:
        if _affinity['pid'] != os.getpid():     # first use / a forked child
            _affinity.update(pid=os.getpid(),
                original=os.sched_getaffinity(0), bound=0)
        buf = C.create_string_buffer(32)
        check(load().bnpc_device_pci_bus_id(int(device), buf, 32),
            'device_pci_bus_id')
        bdf = buf.value.decode().lower()
        with open(os.path.join(pci_sysfs, bdf, 'numa_node')) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        original = _affinity['original']
        near = _cpus_of_node(node, node_sysfs) & original
        if not near or near == original:
            return None
        _set_affinity_all_threads(near)
        _affinity['bound'] += 1
        return node
    except (OSError, ValueError, RuntimeError, AttributeError):
        return None


def release_binding():
    """Undo bind_near_device when the last context that bound the process is
    closed: the process gets the affinity mask it started with back."""
    if _affinity['pid'] != os.getpid() or _affinity['bound'] <= 0:
        return
    _affinity['bound'] -= 1
    if _affinity['bound'] == 0:
        try:
            _set_affinity_all_threads(_affinity['original'])
        except (OSError, AttributeError):
            pass


# ---------------------------------------------------------------------------
# numpy legacy stream <-> C struct
# ---------------------------------------------------------------------------
def rng_export():
    """Snapshot of the global legacy np.random state as an MT19937 struct."""
    kind, key, pos, has_gauss, cached = np.random.get_state()
    st = MT19937()
    C.memmove(st.key, ptr(key), 624 * 4)
    st.pos = pos
    return st, (has_gauss, cached)


def rng_import(st, extra):
    """Write an MT19937 struct back into the global legacy np.random state."""
    key = np.frombuffer(st.key, dtype=np.uint32).copy()
    np.random.set_state(('MT19937', key, int(st.pos), extra[0], extra[1]))


def codist(assignments, device=None):
    """Condensed int32 counts of samples in which two cells differ in label
    (bnpc_codist); assignments: samples x cells integer array."""
    a = np.ascontiguousarray(assignments, dtype=np.int32)
    S, N = a.shape
    out = np.empty(N * (N - 1) // 2, dtype=np.int32)
    if device is None:
        device = int(os.environ.get('BNPC_DEVICE', '0'))
    mark_gpu_touched()
    check(load().bnpc_codist(device, ptr(a, C.c_int32), S, N,
        ptr(out, C.c_int32)), 'codist')
    return out


def ward_finish(raw, n):
    """nn_chain's epilogue: Z = raw[argsort(height, stable)], then `label`:
    the two children of merge i become the current roots of the slots it
    names (smaller root first), the merged cluster gets id n + i and the
    union's size."""
    Z = raw[np.argsort(raw[:, 2], kind='mergesort')]
    parent = np.arange(2 * n - 1, dtype=np.int64)
    sizes = np.ones(2 * n - 1, dtype=np.int64)
    lo = Z[:, 0].astype(np.int64)
    hi = Z[:, 1].astype(np.int64)

    def find(x):
        p = x
        while parent[x] != x:
            x = parent[x]
        while parent[p] != x:
            p, parent[p] = parent[p], x
        return x
    for i in range(n - 1):
        a, b = find(lo[i]), find(hi[i])
        if a > b:
            a, b = b, a
        Z[i, 0], Z[i, 1] = a, b
        parent[a] = parent[b] = n + i
        sizes[n + i] = sizes[a] + sizes[b]
        Z[i, 3] = sizes[n + i]
    return Z


class Posterior:
    """Device-resident pair counts of a set of posterior samples (bnpc_post):
    differ[(i, j)] = samples in which cells i and j carry different labels,
    condensed in pdist order; assignments: samples x cells integers."""

    def __init__(self, assignments, device=None):
        a = np.ascontiguousarray(assignments, dtype=np.int32)
        self.S, self.N = a.shape
        if device is None:
            device = int(os.environ.get('BNPC_DEVICE', '0'))
        mark_gpu_touched()
        handle, total = C.c_void_p(), _i64(0)
        check(load().bnpc_post_create(device, ptr(a), self.S, self.N,
            C.byref(handle), C.byref(total)), 'post_create')
        self._h = handle
        self.differ_sum = total.value       # sum of all pair counts

    @property
    def pairs(self):
        return self.N * (self.N - 1) // 2

    def differ(self):
        out = np.empty(self.pairs, dtype=np.int32)
        check(load().bnpc_post_fetch(self._h, ptr(out), None), 'post_fetch')
        return out

    def dist(self):
        """differ / S as float64 (utils.get_dist), divided on the device."""
        out = np.empty(self.pairs, dtype=np.float64)
        check(load().bnpc_post_fetch(self._h, None, ptr(out)), 'post_fetch')
        return out

    def ward(self):
        """scipy.cluster.hierarchy.linkage(self.dist(), method='ward'): the
        nearest-neighbour chain on the device (bnpc_post_ward), then SciPy's
        own last two steps - the stable sort of the merges by height and the
        relabelling of the clusters (scipy/cluster/_hierarchy.pyx: nn_chain,
        label, LinkageUnionFind) - on the host."""
        n = self.N
        raw = np.empty((n - 1, 4), dtype=np.float64)
        check(load().bnpc_post_ward(self._h, ptr(raw)), 'post_ward')
        return ward_finish(raw, n)

    def ward_stats(self):
        """(full row scans, chain steps) of the last ward()"""
        scans, steps = _i64(0), _i64(0)
        check(load().bnpc_post_ward_stats(self._h, C.byref(scans),
            C.byref(steps)), 'post_ward_stats')
        return scans.value, steps.value

    def mpear_sums(self, labels):
        """labels: (C, N) integer array of C candidate clusterings ->
        int64 (C,): sum of differ over the pairs that share a label."""
        labels = np.asarray(labels)
        assert labels.ndim == 2 and labels.shape[1] == self.N
        if labels.size and (labels.min() < 0 or labels.max() >= 65534):
            raise ValueError('cluster labels must lie in [0, 65534)')
        lab = np.ascontiguousarray(labels, dtype=np.uint16)
        out = np.empty(lab.shape[0], dtype=np.int64)
        for c0 in range(0, lab.shape[0], 1024):
            part = np.ascontiguousarray(lab[c0:c0 + 1024])
            check(load().bnpc_post_mpear(self._h, ptr(part),
                part.shape[0], ptr(out[c0:])), 'post_mpear')
        return out

    def close(self):
        if getattr(self, '_h', None):
            load().bnpc_post_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass


_live = {}


def rng_live():
    """Pointer to the MT19937 state of NumPy's global legacy RandomState, for
    in-place draws from C (no get_state / set_state round trip), or None if
    this NumPy does not expose it as expected (then callers exchange a copy).
    Checked once per process against np.random.get_state()."""
    pid = os.getpid()
    if env('BNPC_STREAM_LIVE', '1') == '0':     # copies through get_state
        return None
    rs = np.random.mtrand._rand
    # (np.random.set_bit_generator / a replaced _rand give the stream another
    # state block: the pointer is looked up again whenever the owners change)
    if _live.get('pid') != pid or _live.get('rs') is not rs \
            or _live.get('bg') is not getattr(rs, '_bit_generator', None):
        _live.clear()
        _live['pid'] = pid
        _live['ptr'] = None
        _live['rs'] = rs
        try:
            bg = _live['bg'] = rs._bit_generator
            if type(bg).__name__ == 'MT19937':
                p = C.cast(bg.ctypes.state_address, C.POINTER(MT19937))
                kind, key, pos = np.random.get_state()[:3]
                if kind == 'MT19937' and p.contents.pos == pos and \
                        np.array_equal(np.frombuffer(p.contents.key,
                            dtype=np.uint32), key):
                    _live['ptr'] = p
        except Exception:
            _live['ptr'] = None
    return _live['ptr']


class LegacyGauss(C.Structure):
    """bnpc_legacy_gauss (include/bnpc_hip.h) = the (has_gauss, gauss) part of
    NumPy's aug_bitgen_t"""
    _fields_ = [('has_gauss', C.c_int32), ('pad_', C.c_int32),
        ('gauss', C.c_double)]


_gauss_live = {}


def gauss_live():
    """Pointer to the cached Gaussian (has_gauss, gauss) of NumPy's global
    legacy RandomState, for native draws that use the polar method (the
    legacy gamma / beta samplers), or None.  NumPy has no accessor for it:
    the pair sits in the RandomState object itself (its aug_bitgen_t), so it
    is LOCATED once per process - set_state() plants two different marker
    values, the object's memory is searched for the (int, double) pair that
    follows them, the candidate must be unique, must read back (0, 0.0) after
    a reset and a value written through the pointer must come back out of
    get_state().  Anything unexpected: None, and the callers exchange the
    pair through get_state / set_state instead."""
    pid = os.getpid()
    rs_now = np.random.mtrand._rand
    if _gauss_live.get('pid') == pid and _gauss_live.get('owner') is rs_now \
            and _gauss_live.get('bg') is getattr(rs_now, '_bit_generator',
                None):
        return _gauss_live['ptr']
    _gauss_live.clear()
    _gauss_live.update(pid=pid, ptr=None, owner=rs_now,
        bg=getattr(rs_now, '_bit_generator', None))
    if os.environ.get('BNPC_STREAM_LIVE', '1') in ('0', 'rng'):
        return None
    saved = np.random.get_state()
    try:
        rs = np.random.mtrand._rand
        base, size = id(rs), type(rs).__basicsize__
        if not 64 <= size <= 1 << 16:
            return None
        kind, key, pos = saved[:3]

        def hits(marker):
            np.random.set_state((kind, key, pos, 1, marker))
            raw = C.string_at(base, size)
            found = set()
            for off in range(0, size - 15, 8):
                if int.from_bytes(raw[off:off + 4], 'little') == 1 and \
                        np.frombuffer(raw, np.float64, 1, off + 8)[0] == marker:
                    found.add(off)
            return found
        cand = hits(0.7421875123) & hits(-3.1403125456)
        if len(cand) != 1:
            return None
        ptr = C.cast(base + cand.pop(), C.POINTER(LegacyGauss))
        np.random.set_state((kind, key, pos, 0, 0.0))
        if ptr.contents.has_gauss != 0 or ptr.contents.gauss != 0.0:
            return None
        ptr.contents.has_gauss, ptr.contents.gauss = 1, 1.2509765625
        back = np.random.get_state()
        if back[3] != 1 or back[4] != 1.2509765625:
            return None
        _gauss_live['ptr'] = ptr
    except Exception:       # noqa: BLE001 - any surprise means "not live"
        _gauss_live['ptr'] = None
    finally:
        np.random.set_state(saved)
    return _gauss_live['ptr']


class NumpyGaussStream:
    """`with NumpyGaussStream() as (rng, gauss):` - the global legacy stream
    AND its cached Gaussian for native draws: both in place when they can be
    located, else copies that are written back on exit."""

    _fast = {}      # (pid, RandomState, bit generator) -> the two pointers

    def __enter__(self):
        # (both pointers hold while the process, the global RandomState and
        # its bit generator are the ones they were located in: three identity
        # tests instead of the two look-ups and a cast per step)
        rs = np.random.mtrand._rand
        fast = NumpyGaussStream._fast
        if fast.get('rs') is rs and fast.get('pid') == os.getpid() \
                and fast.get('bg') is getattr(rs, '_bit_generator', None) \
                and fast.get('env') == env('BNPC_STREAM_LIVE', '1'):
            self._gauss = fast['gauss']
            return fast['pair']
        self._rng = rng_live()
        self._gauss = gauss_live() if self._rng is not None else None
        if self._gauss is not None:
            pair = (self._rng, C.cast(self._gauss, C.c_void_p))
            fast.clear()
            fast.update(rs=rs, pid=os.getpid(),
                bg=getattr(rs, '_bit_generator', None),
                env=env('BNPC_STREAM_LIVE', '1'), gauss=self._gauss,
                pair=pair)
            return pair
        self._copy, extra = rng_export()
        self._g = LegacyGauss(int(extra[0]), 0, float(extra[1]))
        return C.pointer(self._copy), C.cast(C.pointer(self._g), C.c_void_p)

    def __exit__(self, *exc):
        if self._gauss is None:
            rng_import(self._copy, (int(self._g.has_gauss),
                float(self._g.gauss)))
        return False


def beta(a, b):
    """np.random.beta(a, b) for equal-shaped float64 arrays, natively on the
    global stream (bnpc_mt_beta)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    assert a.shape == b.shape
    out = np.empty(a.shape)
    with NumpyGaussStream() as (rng, g):
        check(load().bnpc_mt_beta(rng, g, a.size, ptr(a), ptr(b), ptr(out)),
            'mt_beta')
    return out


def beta_theta(p, q, n1, n0, fkt, tmin, tmax):
    """float32 profile row clip(Beta(p + n1 * fkt, q + n0 * fkt)) from int32
    column counts, natively on the global stream (bnpc_mt_beta_theta)."""
    n1 = np.ascontiguousarray(n1, dtype=np.int32)
    n0 = np.ascontiguousarray(n0, dtype=np.int32)
    assert n1.shape == n0.shape and n1.ndim == 1
    out = np.empty(n1.size, dtype=np.float32)
    with NumpyGaussStream() as (rng, g):
        check(load().bnpc_mt_beta_theta(rng, g, n1.size, float(p), float(q),
            ptr(n1), ptr(n0), float(fkt), float(tmin), float(tmax),
            ptr(out)), 'mt_beta_theta')
    return out


class NumpyStream:
    """`with NumpyStream() as rng:` yields a ctypes reference to the global
    legacy stream for native draws: NumPy's own state in place when possible,
    else a copy that is written back on exit."""

    def __enter__(self):
        self._ptr = rng_live()
        if self._ptr is not None:
            return self._ptr
        self._copy, self._extra = rng_export()
        return C.pointer(self._copy)

    def __exit__(self, *exc):
        if self._ptr is None:
            rng_import(self._copy, self._extra)
        return False


def permutation(n):
    """np.random.permutation(n), natively on the global stream (the legacy
    shuffle: one masked-rejection interval draw per element)."""
    out = np.empty(n, dtype=np.int64)
    with NumpyStream() as rng:
        check(load().bnpc_mt_permutation(rng, n, ptr(out)), 'mt_permutation')
    return out


def mh_draws(G, M, n_sd):
    """(sd_idx int32 (G, M), U (G, M), u (G, M)): the per-cluster draws of
    MH_cluster_params from the global stream, natively."""
    sd_idx = np.empty((G, M), dtype=np.int32)
    U = np.empty((G, M))
    u = np.empty((G, M))
    with NumpyStream() as rng:
        check(load().bnpc_mt_mh_draws(rng, G, M, n_sd, ptr(sd_idx, C.c_int32),
            ptr(U, C.c_double), ptr(u, C.c_double)), 'mh_draws')
    return sd_idx, U, u


_threads_memo = {}


def _host_share():
    """Chains running next to this one on the same GPU / NUMA node
    (BNPC_HOST_SHARE, set per worker by bnpc_amd.mcmc)."""
    try:
        return max(1, int(env('BNPC_HOST_SHARE') or 1))
    except ValueError:
        return 1


def _host_cores():
    """Logical CPUs this process may run on."""
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def _default_team(cap):
    """A chain alone: one rank per CPU up to `cap`.  Chains sharing a node:
    a quarter of the logical CPUs divided among them (measured on the
    2 x 64-core MI355X host, 128 logical CPUs per node, config 3, aggregate
    steps/s: 8 chains x 16 ranks 1980, x 8 2550, x 4 3170, x 2 2830;
    4 chains x 16 1990, x 8 2280, x 4 2020; 2 chains x 16 1320, x 8 1240)."""
    share = _host_share()
    if share == 1:
        return max(1, min(cap, _host_cores()))
    return max(1, min(cap, _host_cores() // (4 * share)))


def host_threads():
    """Size of the native host thread team (BNPC_HOST_THREADS; default: up to
    8 ranks for a chain alone - see _default_team; 1 = the calling thread
    only).  The ranks a job really uses follow its size (bnpc_hostmath.cpp)."""
    env_threads = env('BNPC_HOST_THREADS')
    key = (env_threads, env('BNPC_HOST_SHARE'))
    n = _threads_memo.get(key)
    if n is None:
        try:
            n = int(env_threads or 0)
        except ValueError:
            n = 0
        if n < 1:
            n = _default_team(8)
        _threads_memo[key] = n
    return n


_mh_scratch = {}


def _mh_buffers(G, M):
    """Per-shape scratch of mh_batch that never leaves it (the draws, the log
    acceptance ratios) and a pre-filled argument block: allocating and
    describing them per call costs as much as a small batch itself."""
    key = (os.getpid(), G, M)
    buf = _mh_scratch.get(key)
    if buf is None:
        if len(_mh_scratch) > 16:
            _mh_scratch.clear()
        sd_idx = np.empty((G, M), dtype=np.int32)
        U = np.empty((G, M))
        u = np.empty((G, M))
        A = np.empty((G, M))
        args = MHArgs()
        args.G, args.M = G, M
        args.sd_idx, args.U, args.u = (ptr(x) for x in (sd_idx, U, u))
        args.A = ptr(A)
        buf = _mh_scratch[key] = (args, sd_idx, U, u, A)
    return buf


MH_WIDE_FROM = 65536        # batch entries from which up to 32 ranks are used


def threads_for(elements):
    """Team ranks for a parameter batch of `elements` matrix entries: from
    65536 entries on (MH_WIDE_FROM; round 3: the device screens the
    batches of a converged step, a config-3 chain runs as fast on 4 ranks as
    on 32, and every extra rank that is woken costs) up to 32 ranks,
    unless BNPC_HOST_THREADS pins the number or the node is shared with other
    chains.  Measured on the 2 x 64-core host, us per bnpc_mh_batch call, 16
    -> 32 ranks: 3 x 1000 (a restricted scan) 67-76 -> 52-58, 10 x 1000
    120-133 -> 75-107, 50 x 5000 2.5 ms -> 1.4 ms; config-3 bench, six
    interleaved pairs: median 826 -> 869 steps/s.  (The sweeps' team scan does
    not gain from more than 16.)"""
    n = host_threads()
    if elements >= MH_WIDE_FROM and env('BNPC_HOST_THREADS') is None:
        n = max(n, min(_default_team(32), _host_cores() // 2))
    return n


def rows_copy_zero(dst, src=None):
    """bnpc_rows_copy_zero: the rows (axis 0) of `dst` - fresh memory, e.g.
    np.empty - written for the first time on the host team: row r = row r of
    `src` (the same dtype and number of rows, at most as many bytes per row)
    followed by zeros; zeros alone without `src`.  Returns dst."""
    rows = dst.shape[0]
    if rows == 0 or dst.nbytes == 0:
        return dst
    assert dst.flags['C_CONTIGUOUS'] and dst.flags['WRITEABLE']
    width = dst.nbytes // rows
    copy = 0
    if src is not None:
        assert src.flags['C_CONTIGUOUS'] and src.dtype == dst.dtype \
            and src.shape[0] == rows and src.nbytes <= dst.nbytes
        copy = src.nbytes // rows
    check(load().bnpc_rows_copy_zero(ptr(dst), width,
        ptr(src) if copy else None, copy, rows, copy, width,
        threads_for(dst.nbytes)), 'rows_copy_zero')
    return dst


def mh_batch(kernels, old, n1, n0, sd, tmin, tmax, FP, FN, p, q, uniform,
            trans_prob, known=None, want_prior=False, draws=None,
            threads=None, ctx=None, counts_src=0, label=None):
    """bnpc_mh_batch: the draws and the arithmetic of MH_cluster_params for
    the G rows of `old` (float32 G x M).  `draws` = (sd_idx, U, u) evaluates
    given draws instead of taking them from the global stream.  With `ctx`
    (a device Context whose resident counts - counts_src 0: by label, 1: the
    last view counts - are those of n1 / n0) the device screens the batch
    first and the host evaluates only what it leaves (bnpc_mh_batch_dev);
    with `label` = (assignment, ids) the per-cluster counts are made in the
    same call and n1 / n0 RECEIVE them (bnpc_label_counts_and_batch).
    Returns (status, new, log_prob, declined, prior, (sd_idx, U, u)); status
    1: only the draws are valid - they are views of scratch that the next
    call overwrites."""
    old = np.ascontiguousarray(old, dtype=np.float32)
    G, M = old.shape
    n1 = np.ascontiguousarray(n1, dtype=np.int32)
    n0 = np.ascontiguousarray(n0, dtype=np.int32)
    sd = np.ascontiguousarray(sd, dtype=np.float64)
    assert n1.shape == n0.shape == (G, M)
    a, sd_idx, U, u, _ = _mh_buffers(G, M)
    if draws is not None:
        sd_idx[...] = draws[0]
        U[...] = draws[1]
        u[...] = draws[2]
    new = np.empty((G, M), dtype=np.float32)
    prior = np.empty((G, M)) if want_prior and not uniform else None
    log_prob = np.empty(G)
    declined = np.empty(G, dtype=np.int64)
    kt = kp = None
    if known is not None and not uniform:
        kt = np.ascontiguousarray(known[0], dtype=np.float32)
        kp = np.ascontiguousarray(known[1], dtype=np.float64)
        assert kt.shape == kp.shape == (G, M)
    a.old_theta, a.n1, a.n0 = ptr(old), ptr(n1), ptr(n0)
    a.sd, a.n_sd = ptr(sd), sd.size
    a.tmin, a.tmax, a.FP, a.FN, a.p, a.q = tmin, tmax, FP, FN, p, q
    a.uniform_prior, a.trans_prob = int(bool(uniform)), int(bool(trans_prob))
    a.known_theta = ptr(kt) if kt is not None else None
    a.known_prior = ptr(kp) if kp is not None else None
    a.new_theta = ptr(new)
    a.prior_out = ptr(prior) if prior is not None else None
    a.log_prob, a.declined = ptr(log_prob), ptr(declined)
    a.threads = threads_for(G * M) if threads is None else threads
    a.screen = None
    status = C.c_int(0)
    lib = load()
    handle = getattr(ctx, '_h', None)
    if label is not None:
        assert handle and draws is None and not trans_prob
        cells, ids = as_i64(label[0]), as_i64(label[1])
        assert ids.size == G and n1.flags['WRITEABLE']
        with NumpyStream() as rng:
            check(lib.bnpc_label_counts_and_batch(handle,
                C.addressof(kernels), rng, ptr(cells),
                ptr(ids), C.byref(a), C.byref(status)),
                'label_counts_and_batch')
    elif handle and not trans_prob:
        if draws is None:
            with NumpyStream() as rng:
                check(lib.bnpc_mh_batch_dev(handle, C.addressof(kernels), rng,
                    C.byref(a), counts_src, C.byref(status)), 'mh_batch_dev')
        else:
            check(lib.bnpc_mh_batch_dev(handle, C.addressof(kernels), None,
                C.byref(a), counts_src, C.byref(status)), 'mh_batch_dev')
    elif draws is None:
        with NumpyStream() as rng:
            check(lib.bnpc_mh_batch(C.addressof(kernels), rng, C.byref(a),
                C.byref(status)), 'mh_batch')
    else:
        check(lib.bnpc_mh_batch(C.addressof(kernels), None, C.byref(a),
            C.byref(status)), 'mh_batch')
    return status.value, new, log_prob, declined, prior, (sd_idx, U, u)


def log_A(kernels, new, old, std, n1, n0, fmin, fmax, tmin, tmax, FP, FN, p,
            q, uniform, clip, threads=None):
    """bnpc_log_accept: (A (G, M), row sums (G,)) or None when the library leaves
    an element to SciPy."""
    new = np.ascontiguousarray(np.atleast_2d(new), dtype=np.float32)
    old = np.ascontiguousarray(np.atleast_2d(old), dtype=np.float32)
    std = np.ascontiguousarray(np.atleast_2d(std), dtype=np.float64)
    n1 = np.ascontiguousarray(np.atleast_2d(n1), dtype=np.int32)
    n0 = np.ascontiguousarray(np.atleast_2d(n0), dtype=np.int32)
    G, M = new.shape
    assert old.shape == std.shape == n1.shape == n0.shape == (G, M)
    A = np.empty((G, M))
    total = np.empty(G)
    a = LogAArgs(G, M, ptr(new), ptr(old), ptr(std),
        ptr(n1), ptr(n0), fmin, fmax, tmin, tmax, FP, FN, p, q,
        int(bool(uniform)), int(bool(clip)), ptr(A), ptr(total),
        host_threads() if threads is None else threads)
    status = C.c_int(0)
    check(load().bnpc_log_accept(C.addressof(kernels), C.byref(a),
        C.byref(status)), 'log_A')
    return None if status.value else (A, total)


def rg_scan_step(ctx, kernels, view, n, rg_assignment, DP_a, theta3, sd, tmin,
            tmax, FP, FN, p, q, uniform, trans_prob=False, threads=None):
    """bnpc_rg_scan_step on the device context `ctx`: (status, new theta
    (G, M), n1, n0 (G, M) int32, (sd_idx, U, u), scan log-prob, batch
    log-probs (G,)); rg_assignment is updated in place.  G = 3 rows (launch
    clusters + merged) or 2.  status 1: the scan and the counts are done and
    the draws taken; the parameter rows are to be evaluated from the draws."""
    theta3 = np.ascontiguousarray(theta3, dtype=np.float32)
    G, M = theta3.shape
    assert G in (2, 3) and rg_assignment.dtype == np.int64 \
        and rg_assignment.flags['C_CONTIGUOUS'] and rg_assignment.size == n - 2
    sd = np.ascontiguousarray(sd, dtype=np.float64)
    a, sd_idx, U, u, _ = _mh_buffers(G, M)
    n1 = np.empty((G, M), dtype=np.int32)
    n0 = np.empty((G, M), dtype=np.int32)
    new = np.empty((G, M), dtype=np.float32)
    log_prob = np.empty(G)
    declined = np.empty(G, dtype=np.int64)
    a.old_theta, a.n1, a.n0 = ptr(theta3), ptr(n1), ptr(n0)
    a.sd, a.n_sd = ptr(sd), sd.size
    a.tmin, a.tmax, a.FP, a.FN, a.p, a.q = tmin, tmax, FP, FN, p, q
    a.uniform_prior = int(bool(uniform))
    a.trans_prob = int(bool(trans_prob))
    a.known_theta = a.known_prior = a.prior_out = None
    a.new_theta = ptr(new)
    a.log_prob, a.declined = ptr(log_prob), ptr(declined)
    a.threads = threads_for(G * M) if threads is None else threads
    a.screen = None
    status = C.c_int(0)
    scan_prob = C.c_double(0.0)
    with NumpyStream() as rng:
        check(load().bnpc_rg_scan_step(ctx._h, C.addressof(kernels), rng, view,
            n, ptr(rg_assignment), float(DP_a), C.byref(a),
            ptr(n1), ptr(n0), C.byref(scan_prob),
            C.byref(status)), 'rg_scan_step')
    return (status.value, new, n1, n0, (sd_idx, U, u), scan_prob.value,
        log_prob)


def f32_up(x):
    """float32 values not below the float64 ones (rounded to nearest, one step
    up where that fell short): the record's bounds third / fourth."""
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(all='ignore'):
        f = x.astype(np.float32)
        low = f.astype(np.float64) < x
        return np.where(low, np.nextafter(f, np.float32(np.inf)), f) \
            .astype(np.float32)


def hints_from_matrix(mat, col_prior):
    """The hint records the device returns for the first K = len(col_prior)
    columns of `mat` (k_row_top2: the four largest entries of ll + prior,
    first one on ties, the columns and log-likelihoods of the three largest)
    - for the CPU stand-in of the device and the tests."""
    mat = np.asarray(mat)
    col_prior = np.asarray(col_prior, dtype=np.float64)
    K = col_prior.size
    ll = mat[:, :K]
    post = ll + col_prior[None, :]
    n = post.shape[0]
    rows = np.arange(n)
    order = np.argsort(-post, axis=1, kind='stable')
    hint = np.zeros(n, dtype=TOP2)
    for rank, (val, lik, col) in enumerate((('best', 'll_best', 'col'),
            ('second', 'll_second', 'col2'), ('third', 'll_third', 'col3'))):
        if rank < K:
            c = order[:, rank]
            hint[val] = f32_up(post[rows, c]) if val == 'third' \
                else post[rows, c]
            hint[lik] = ll[rows, c]
            hint[col] = np.where(np.isfinite(hint[val]) | (rank == 0), c, -1)
        else:
            hint[val], hint[col] = -np.inf, -1
    hint['fourth'] = f32_up(post[rows, order[:, 3]]) if K > 3 else -np.inf
    # the weights of the second / third column relative to the first, priors
    # left out (float32)
    with np.errstate(all='ignore'):
        hint['e2'] = np.where(hint['col2'] >= 0,
            np.exp(hint['ll_second'] - hint['ll_best']), 0.0)
        hint['e3'] = np.where(hint['col3'] >= 0,
            np.exp(hint['ll_third'] - hint['ll_best']), 0.0)
    return hint


def wide_hints_from_matrix(mat, col_prior):
    """The hint records of a tile (k_row_top2_wide): the largest entry of
    ll + prior over the first len(col_prior) columns, its column (first one on
    ties) as 32 bits in col | col2 << 16, the largest entry among the other
    columns - for the CPU stand-in of the device and the tests."""
    mat = np.asarray(mat)
    col_prior = np.asarray(col_prior, dtype=np.float64)
    K = col_prior.size
    post = mat[:, :K] + col_prior[None, :]
    n = post.shape[0]
    rows = np.arange(n)
    col = np.argmax(post, axis=1)           # first maximum
    hint = np.zeros(n, dtype=TOP2)
    hint['best'] = post[rows, col]
    rest = post.copy()
    rest[rows, col] = -np.inf
    hint['second'] = rest.max(axis=1) if K > 1 else -np.inf
    hint['third'] = hint['fourth'] = -np.inf
    hint['col'] = (col & 0xffff).astype(np.uint16).view(np.int16)
    hint['col2'] = (col >> 16).astype(np.uint16).view(np.int16)
    hint['col3'] = -1
    hint['row_here'] = 2
    return hint


def np_sum(a):
    """np.sum of a float64 vector as the library restates it (checker)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = C.c_double(0.0)
    check(load().bnpc_np_sum(ptr(a, C.c_double), a.size, C.byref(out)),
        'np_sum')
    return out.value


def _move_state(move, scan_no, ids, sizes, assignment, parameters, DP_a, sd,
            FP, FN, p, q, uniform, tmin, tmax, fill, view):
    st = MoveState()
    st.move, st.scan_no, st.view = int(move), int(scan_no), int(view)
    st.uniform_prior = int(bool(uniform))
    N, M = parameters.shape
    st.threads, st.threads_wide = threads_for(3 * M), host_threads()
    st.K, st.ids, st.sizes = ids.size, ptr(ids), ptr(sizes)
    st.N, st.M = N, M
    st.assignment, st.parameters = ptr(assignment), \
        ptr(parameters)
    st.param_stride = parameters.strides[0] // 4
    st.DP_a, st.sd, st.n_sd = float(DP_a), ptr(sd), sd.size
    st.FP, st.FN, st.p, st.q = float(FP), float(FN), float(p), float(q)
    st.tmin, st.tmax, st.fill = float(tmin), float(tmax), float(fill)
    return st


def _move_arrays(ids, sizes, assignment, parameters, sd):
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    sizes = np.ascontiguousarray(sizes, dtype=np.int64)
    sd = np.ascontiguousarray(sd, dtype=np.float64)
    assert assignment.dtype == np.int64 and assignment.flags['C_CONTIGUOUS'] \
        and assignment.flags['WRITEABLE'] and parameters.dtype == np.float32 \
        and parameters.ndim == 2 and parameters.strides[1] == 4 \
        and parameters.flags['WRITEABLE'] and ids.size == sizes.size \
        and assignment.size == parameters.shape[0]
    return ids, sizes, sd


def sm_move(ctx, kernels, move, scan_no, ids, sizes, assignment, parameters,
            DP_a, sd, FP, FN, p, q, uniform, tmin, tmax, fill, view):
    """bnpc_sm_move: a whole split (move 0) / merge (1) move on the global
    stream; assignment (int64) and parameters (float32, row = cluster id) are
    updated in place when the move is accepted.  None: not done natively -
    the stream is where it was, nothing was modified.  Else (accepted, cl_i,
    cl_j, moved cells, cells of the move, log acceptance ratio)."""
    ids, sizes, sd = _move_arrays(ids, sizes, assignment, parameters, sd)
    st = _move_state(move, scan_no, ids, sizes, assignment, parameters, DP_a,
        sd, FP, FN, p, q, uniform, tmin, tmax, fill, view)
    status = C.c_int(1)
    with NumpyGaussStream() as (rng, gauss):
        st.gauss = gauss
        check(load().bnpc_sm_move(ctx._h, C.addressof(kernels), rng,
            C.byref(st), C.byref(status)), 'sm_move')
    if status.value:
        return None
    return (bool(st.accepted), st.cl_i, st.cl_j, st.moved, st.n_cells,
        st.log_A)


def move_propose(kernels, move, ids, sizes, assignment):
    """The proposal of a split / merge move alone on the global stream
    (checker): (cells, cells of the first cluster, positions in ids, size
    term, sizes of the other clusters) or None."""
    assignment = np.ascontiguousarray(assignment, dtype=np.int64)
    N = assignment.size
    dummy = np.zeros((N, 1), dtype=np.float32)
    ids, sizes, sd = _move_arrays(ids, sizes, assignment, dummy, [1.0])
    st = _move_state(move, 0, ids, sizes, assignment, dummy, 1.0, sd, 0, 0,
        1, 1, True, 0, 1, 0.5, 1)
    cells = np.empty(N, dtype=np.int64)
    others = np.empty(max(ids.size, 1), dtype=np.int64)
    n_cells, n_first = _i64(0), _i64(0)
    picked = np.empty(2, dtype=np.int64)
    size_data = C.c_double(0.0)
    status = C.c_int(0)
    with NumpyStream() as rng:
        check(load().bnpc_move_propose(C.addressof(kernels), rng,
            C.byref(st), ptr(cells), C.byref(n_cells), C.byref(n_first),
            ptr(picked), C.byref(size_data), ptr(others),
            C.byref(status)), 'move_propose')
    if status.value:
        return None
    return (cells[:n_cells.value].copy(), n_first.value, picked,
        size_data.value, others[:ids.size - 1].copy())


def gamma(shape, scale):
    """np.random.gamma(shape, scale), one draw, natively on the global stream
    (bnpc_mt_gamma)."""
    out = C.c_double(0.0)
    with NumpyGaussStream() as (rng, g):
        check(load().bnpc_mt_gamma(rng, g, float(shape), float(scale),
            C.byref(out)), 'mt_gamma')
    return out.value


def gamma_logpdf_scalar(kernels, x, a, loc):
    """scipy.stats.gamma.logpdf(x, a, loc) on the kernel table (the prior of
    DP_a as bnpc_chain_step evaluates it), or None (left to SciPy)."""
    out, status = C.c_double(0.0), C.c_int(0)
    check(load().bnpc_gamma_logpdf_scalar(C.addressof(kernels), float(x),
        float(a), float(loc), C.byref(out), C.byref(status)),
        'gamma_logpdf_scalar')
    return None if status.value else out.value


class NativeChain:
    """A bnpc_chain with its work area (bnpc_chain_open / _close) and the
    arrays it points at: the live clusters in dict order (capacity N)."""

    def __init__(self, N, M):
        self.st = ChainState()
        self.st.N, self.st.M = N, M
        self.ids = np.zeros(N, dtype=np.int64)
        self.sizes = np.zeros(N, dtype=np.int64)
        self.st.ids, self.st.sizes = ptr(self.ids), ptr(self.sizes)
        self.keep = []          # arrays the structure points at
        self._lib = load()
        check(self._lib.bnpc_chain_open(C.byref(self.st)), 'chain_open')

    def close(self):
        if getattr(self, 'st', None) is not None and self.st.work:
            self._lib.bnpc_chain_close(C.byref(self.st))

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass


def tn_ppf_scalar(kernels, q, a, b, loc, scale):
    """truncnorm.ppf for scalars on the kernel table, or None (left to
    SciPy)."""
    out, status = C.c_double(0.0), C.c_int(0)
    check(load().bnpc_tn_ppf_scalar(C.addressof(kernels), q, a, b, loc, scale,
        C.byref(out), C.byref(status)), 'tn_ppf_scalar')
    return None if status.value else np.float64(out.value)


def tn_logpdf_scalar(kernels, x, a, b, loc, scale):
    """truncnorm.logpdf for scalars on the kernel table, or None when the
    library leaves the interval to SciPy."""
    out, status = C.c_double(0.0), C.c_int(0)
    check(load().bnpc_tn_logpdf_scalar(C.addressof(kernels), x, a, b, loc,
        scale, C.byref(out), C.byref(status)), 'tn_logpdf_scalar')
    return None if status.value else np.float64(out.value)


def beta_logpdf_f32(kernels, x, p, q, known=None, threads=None):
    """(density array shaped like x, its sum in index order)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape)
    total = C.c_double(0.0)
    kt = kp = None
    if known is not None:
        kt = np.ascontiguousarray(known[0], dtype=np.float32)
        kp = np.ascontiguousarray(known[1], dtype=np.float64)
        assert kt.shape == kp.shape == x.shape
    check(load().bnpc_beta_logpdf_f32(C.addressof(kernels),
        ptr(x, C.c_float), x.size, p, q,
        ptr(kt) if kt is not None else None,
        ptr(kp) if kp is not None else None, ptr(out, C.c_double),
        C.byref(total), threads_for(x.size) if threads is None else threads),
        'beta_logpdf_f32')
    return out, total.value


def log_diff_pi(log_p, log_q):
    """log(exp(log_p) - exp(log_q)) with SciPy's complex-logsumexp arithmetic
    (include/bnpc_hip.h: bnpc_log_diff_pi); 1-D float64 in, 1-D out."""
    log_p = np.ascontiguousarray(log_p, dtype=np.float64)
    log_q = np.ascontiguousarray(log_q, dtype=np.float64)
    out = np.empty_like(log_p)
    check(load().bnpc_log_diff_pi(ptr(log_p, C.c_double),
        ptr(log_q, C.c_double), log_p.size, ptr(out, C.c_double)),
        'log_diff_pi')
    return out


class Context:
    """Device-resident data of one chain (bnpc_ctx)."""

    def __init__(self, data=None, codes=None, device=0):
        lib = load()
        mark_gpu_touched()
        handle = _ctx()
        planes = getattr(data, 'planes', None)
        if planes is not None:      # bnpc_amd.bitplanes.BitPlanes
            N, M = data.shape
            self._keep = planes = np.ascontiguousarray(planes, dtype='<u8')
            check(lib.bnpc_create_planes(device, N, M, ptr(planes),
                C.byref(handle)), 'create_planes')
            del self._keep
        elif codes is not None:
            codes = np.ascontiguousarray(codes, dtype=np.int8)
            N, M = codes.shape
            check(lib.bnpc_create_codes(device, N, M, ptr(codes, C.c_int8),
                C.byref(handle)), 'create_codes')
        else:
            data = np.ascontiguousarray(data, dtype=np.float64)
            N, M = data.shape
            check(lib.bnpc_create(device, N, M, ptr(data, C.c_double),
                C.byref(handle)), 'create')
        self._h = handle
        self._lib = lib
        self.N, self.M = N, M
        self.device = device
        self.numa_node = bind_near_device(device)

    def close(self):
        if getattr(self, '_h', None):
            self._lib.bnpc_destroy(self._h)
            self._h = None
            if getattr(self, 'numa_node', None) is not None:
                self.numa_node = None
                release_binding()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def cell_counts(self):
        n1 = np.empty(self.N, dtype=np.int32)
        n0 = np.empty(self.N, dtype=np.int32)
        check(self._lib.bnpc_cell_counts(self._h, ptr(n1, C.c_int32),
            ptr(n0, C.c_int32)), 'cell_counts')
        return n1, n0

    def view_set(self, view, cells):
        cells = as_i64(cells)
        check(self._lib.bnpc_view_set(self._h, view, ptr(cells, C.c_int64),
            cells.size), 'view_set')
        return cells.size

    def view_set_slot(self, view, cells, slot):
        """view_set for the tile that will be issued on `slot`: returns
        without waiting for the device."""
        cells = as_i64(cells)
        check(self._lib.bnpc_view_set_slot(self._h, view,
            ptr(cells, C.c_int64), cells.size, slot), 'view_set_slot')
        return cells.size

    def view_size(self, view):
        n = _i64(0)
        check(self._lib.bnpc_view_size(self._h, view, C.byref(n)),
            'view_size')
        return n.value

    def ll_theta(self, view, theta, FP, FN, out=None, fetch=True):
        """(slots x K) float64 log-likelihoods; theta K x M float32.

        `out` may be a C-contiguous (slots, ld >= K) array: columns K..ld are
        left untouched (room for clusters opened later in a sweep)."""
        theta = np.ascontiguousarray(theta, dtype=np.float32)
        if theta.ndim == 1:
            theta = theta[None, :]
        K = theta.shape[0]
        assert theta.shape[1] == self.M
        n = self.view_size(view)
        ld = 0
        if fetch:
            if out is None:
                out = np.empty((n, K), dtype=np.float64)
            assert out.flags['C_CONTIGUOUS'] and out.shape[0] == n \
                and out.shape[1] >= K and out.dtype == np.float64
            ld = out.shape[1]
            po = ptr(out, C.c_double)
        else:
            po = None
            self._keep = theta      # borrowed until the next sync
        check(self._lib.bnpc_ll_theta(self._h, view, ptr(theta, C.c_float), K,
            float(FP), float(FN), po, ld), 'll_theta')
        return out

    def ll_theta_pinned(self, view, theta, FP, FN, ld):
        """(slots x ld) float64 NumPy VIEW of the context's pinned host
        buffer holding the log-likelihoods in columns [0, K); valid until the
        next pinned call on this context."""
        theta = np.ascontiguousarray(theta, dtype=np.float32)
        K = theta.shape[0]
        n = self.view_size(view)
        host = _host_pd()
        check(self._lib.bnpc_ll_theta_pinned(self._h, view,
            ptr(theta, C.c_float), K, float(FP), float(FN), ld,
            C.byref(host)), 'll_theta_pinned')
        if n == 0:
            return np.empty((0, ld))
        return np.ctypeslib.as_array(host, shape=(n, ld))

    def ll_theta_pinned_top2(self, view, theta, FP, FN, ld, col_prior,
                wait=True):
        """ll_theta_pinned plus the sweep's hint: (matrix view, hint) where
        hint is a structured NumPy VIEW (TOP2 records, one per slot) of
        pinned memory, or None (more than HINT_COLS_MAX columns).  wait=False
        returns once the work is queued: call hints_wait() before reading the
        hints."""
        theta = np.ascontiguousarray(theta, dtype=np.float32)
        K = theta.shape[0]
        if K > HINT_COLS_MAX:
            return self.ll_theta_pinned(view, theta, FP, FN, ld), None
        col_prior = np.ascontiguousarray(col_prior, dtype=np.float64)
        assert col_prior.size == K
        n = self.view_size(view)
        host = _host_pd()
        hint = C.c_void_p()
        fn = self._lib.bnpc_ll_theta_pinned_top2 if wait \
            else self._lib.bnpc_ll_theta_pinned_top2_issue
        check(fn(self._h, view, ptr(theta, C.c_float), K, float(FP),
            float(FN), ld, ptr(col_prior), C.byref(host), C.byref(hint)),
            'll_theta_pinned_top2')
        if n == 0:
            return np.empty((0, ld)), None
        mat = np.ctypeslib.as_array(host, shape=(n, ld))
        if not hint.value:
            return mat, None
        raw = (C.c_char * (n * TOP2.itemsize)).from_address(hint.value)
        return mat, np.frombuffer(raw, dtype=TOP2, count=n)

    def ll_theta_pinned_sums_issue(self, view, theta, FP, FN, ld, col_prior):
        """First half of a hinted sweep in visiting order
        (bnpc_ll_theta_pinned_sums_issue): the element tables and the sums
        are queued; returns the matrix view (rows by slot; complete after
        matrix_wait).  The caller draws its visiting order, then calls
        hints_in_order_issue - nothing else on the context in between."""
        theta = np.ascontiguousarray(theta, dtype=np.float32)
        K = theta.shape[0]
        col_prior = np.ascontiguousarray(col_prior, dtype=np.float64)
        n = self.view_size(view)
        assert col_prior.size == K
        host = _host_pd()
        check(self._lib.bnpc_ll_theta_pinned_sums_issue(self._h, view,
            ptr(theta, C.c_float), K, float(FP), float(FN), ld,
            ptr(col_prior), C.byref(host)), 'll_theta_pinned_sums_issue')
        self._sums_rows = n
        if n == 0:
            return np.empty((0, ld))
        return np.ctypeslib.as_array(host, shape=(n, ld))

    def hints_in_order_issue(self, order):
        """Second half (bnpc_hints_in_order_issue): the hint kernel is queued
        with record r made from row order[r]; returns the records (a view of
        pinned memory, valid after hints_wait) or None (no hint buffer: the
        matrix is complete on the host instead)."""
        order = as_i64(order)
        n = self._sums_rows
        assert order.size == n
        hint = C.c_void_p()
        check(self._lib.bnpc_hints_in_order_issue(self._h,
            ptr(order, C.c_int64), C.byref(hint)), 'hints_in_order_issue')
        if not hint.value or n == 0:
            return None
        raw = (C.c_char * (n * TOP2.itemsize)).from_address(hint.value)
        return np.frombuffer(raw, dtype=TOP2, count=n)

    def ll_theta_pinned_top2_in_order(self, view, theta, FP, FN, ld,
                col_prior, order):
        """Both halves and the wait (tests): (matrix view - rows by slot -,
        hint records with record r made from row order[r])."""
        mat = self.ll_theta_pinned_sums_issue(view, theta, FP, FN, ld,
            col_prior)
        hint = self.hints_in_order_issue(order)
        self.hints_wait()
        return mat, hint

    def hints_wait(self):
        """The hints of the last ll_theta_pinned_top2(wait=False) are
        complete."""
        check(self._lib.bnpc_hints_wait(self._h), 'hints_wait')

    def matrix_wait(self):
        """The matrix of the last ll_theta_pinned_top2 is complete."""
        check(self._lib.bnpc_matrix_wait(self._h), 'matrix_wait')

    def matrix_wait_hook(self):
        """(function address, argument) for bnpc_gibbs_state.matrix_wait."""
        return (C.cast(self._lib.bnpc_matrix_wait, C.c_void_p).value,
            self._h.value if hasattr(self._h, 'value') else self._h)

    def theta_put(self, row0, theta):
        """Store parameter rows on the device (row index = cluster id)."""
        theta = np.ascontiguousarray(theta, dtype=np.float32)
        if theta.ndim == 1:
            theta = theta[None, :]
        assert theta.shape[1] == self.M
        check(self._lib.bnpc_theta_put(self._h, int(row0),
            ptr(theta, C.c_float), theta.shape[0]), 'theta_put')

    def ll_rows_pinned(self, view, rows, FP, FN, ld):
        """ll_theta_pinned with the clusters' parameters taken from the
        resident store (rows = cluster ids)."""
        rows = as_i64(rows)
        n = self.view_size(view)
        host = _host_pd()
        check(self._lib.bnpc_ll_rows_pinned(self._h, view,
            ptr(rows, C.c_int64), rows.size, float(FP), float(FN), ld,
            C.byref(host)), 'll_rows_pinned')
        if n == 0:
            return np.empty((0, ld))
        return np.ctypeslib.as_array(host, shape=(n, ld))

    def ll_rows_issue(self, view, rows, FP, FN, ld, slot):
        """Start ll_rows_pinned for pinned buffer `slot` (0 .. TILE_SLOTS - 1)
        and return."""
        rows = as_i64(rows)
        check(self._lib.bnpc_ll_rows_issue(self._h, view,
            ptr(rows, C.c_int64), rows.size, float(FP), float(FN), ld, slot),
            'll_rows_issue')

    def ll_rows_issue_hint(self, view, rows, FP, FN, ld, slot, col_prior):
        """ll_rows_issue with the tile's hints: col_prior[k] = log prior of
        column k's cluster now."""
        rows = as_i64(rows)
        col_prior = np.ascontiguousarray(col_prior, dtype=np.float64)
        assert col_prior.size == rows.size
        check(self._lib.bnpc_ll_rows_issue_hint(self._h, view,
            ptr(rows, C.c_int64), rows.size, float(FP), float(FN), ld, slot,
            ptr(col_prior)), 'll_rows_issue_hint')

    def ll_rows_wait_hint(self, slot, n_rows, ld):
        """(matrix, hints) of the tile issued on `slot` with hints: hints is a
        TOP2 record array (one per row; the column as 32 bits in col | col2)
        or None."""
        host = _host_pd()
        hint = C.c_void_p()
        check(self._lib.bnpc_ll_rows_wait_hint(self._h, slot, C.byref(host),
            C.byref(hint)), 'll_rows_wait_hint')
        mat = np.ctypeslib.as_array(host, shape=(n_rows, ld))
        if not hint.value:
            return mat, None
        raw = (C.c_char * (n_rows * TOP2.itemsize)).from_address(hint.value)
        return mat, np.frombuffer(raw, dtype=TOP2, count=n_rows)

    def ll_rows_wait(self, slot, n_rows, ld):
        """The (n_rows, ld) result of the tile issued on `slot`."""
        host = _host_pd()
        check(self._lib.bnpc_ll_rows_wait(self._h, slot, C.byref(host)),
            'll_rows_wait')
        return np.ctypeslib.as_array(host, shape=(n_rows, ld))

    def ll_tables(self, view, L1, L0, out=None):
        L1 = np.ascontiguousarray(L1, dtype=np.float64)
        L0 = np.ascontiguousarray(L0, dtype=np.float64)
        if L1.ndim == 1:
            L1, L0 = L1[None, :], L0[None, :]
        K = L1.shape[0]
        assert L1.shape == L0.shape == (K, self.M)
        n = self.view_size(view)
        if out is None:
            out = np.empty((n, K), dtype=np.float64)
        check(self._lib.bnpc_ll_tables(self._h, view, ptr(L1, C.c_double),
            ptr(L0, C.c_double), K, ptr(out, C.c_double), 0), 'll_tables')
        return out

    def colcounts(self, segments):
        """segments: list of int arrays of cell ids -> (n1, n0) G x M int32."""
        G = len(segments)
        sizes = [len(s) for s in segments]
        offs = np.zeros(G + 1, dtype=np.int64)
        np.cumsum(sizes, out=offs[1:])
        cells = as_i64(np.concatenate([np.asarray(s, dtype=np.int64)
            for s in segments])) if offs[-1] else np.zeros(1, dtype=np.int64)
        n1 = np.zeros((G, self.M), dtype=np.int32)
        n0 = np.zeros((G, self.M), dtype=np.int32)
        check(self._lib.bnpc_colcounts(self._h, ptr(cells, C.c_int64),
            ptr(offs, C.c_int64), G, ptr(n1, C.c_int32), ptr(n0, C.c_int32)),
            'colcounts')
        return n1, n0

    def view_counts(self, view, labels, G):
        """(n1, n0) G x M int32 column counts of the view's slots grouped by
        `labels` (one per slot; < 0: not counted)."""
        labels = as_i64(labels)
        assert labels.size == self.view_size(view)
        n1 = np.empty((G, self.M), dtype=np.int32)
        n0 = np.empty((G, self.M), dtype=np.int32)
        check(self._lib.bnpc_view_counts(self._h, view, ptr(labels, C.c_int64),
            G, ptr(n1, C.c_int32), ptr(n0, C.c_int32)), 'view_counts')
        return n1, n0

    def mh_screen(self, counts_src, old, sd, draws, tmin, tmax, FP, FN, p, q,
                uniform, with_theta=False):
        """bnpc_mh_screen: uint8 (G, M), 0 where the proposal is declined for
        certain under the resident counts, 2 / 3 where it is accepted for
        certain (tests); with_theta: (flags, float32 (G, M) holding the
        proposals of the entries flagged 3)."""
        old = np.ascontiguousarray(old, dtype=np.float32)
        G, M = old.shape
        sd = np.ascontiguousarray(sd, dtype=np.float64)
        sd_idx = np.ascontiguousarray(draws[0], dtype=np.int32)
        U = np.ascontiguousarray(draws[1], dtype=np.float64)
        u = np.ascontiguousarray(draws[2], dtype=np.float64)
        a = MHArgs()
        a.G, a.M = G, M
        a.old_theta, a.sd, a.n_sd = ptr(old), ptr(sd), sd.size
        a.tmin, a.tmax, a.FP, a.FN, a.p, a.q = tmin, tmax, FP, FN, p, q
        a.uniform_prior, a.trans_prob = int(bool(uniform)), 0
        a.sd_idx, a.U, a.u = ptr(sd_idx), ptr(U), ptr(u)
        flags = np.empty((G, M), dtype=np.uint8)
        new32 = np.full((G, M), np.nan, dtype=np.float32)
        check(self._lib.bnpc_mh_screen(self._h, counts_src, C.byref(a),
            ptr(flags), ptr(new32)), 'mh_screen')
        return (flags, new32) if with_theta else flags

    def mh_screen_stats(self):
        """(elements screened so far, of those left to the host)"""
        seen, kept = _i64(0), _i64(0)
        check(self._lib.bnpc_mh_screen_stats(self._h, C.byref(seen),
            C.byref(kept)), 'mh_screen_stats')
        return seen.value, kept.value

    def reload_options(self):
        check(self._lib.bnpc_reload_options(self._h), 'reload_options')

    def colcounts_by_label(self, assignment, ids, fetch=True):
        assignment = as_i64(assignment)
        ids = as_i64(ids)
        K = ids.size
        if fetch:
            n1 = np.empty((K, self.M), dtype=np.int32)
            n0 = np.empty((K, self.M), dtype=np.int32)
            p1, p0 = ptr(n1, C.c_int32), ptr(n0, C.c_int32)
        else:
            n1 = n0 = p1 = p0 = None
        check(self._lib.bnpc_colcounts_by_label(self._h,
            ptr(assignment, C.c_int64), ptr(ids, C.c_int64), K, p1, p0),
            'colcounts_by_label')
        return n1, n0

    def ll_total(self, theta, FP, FN):
        """Total log-likelihood(s) under the resident per-cluster counts."""
        theta = np.ascontiguousarray(theta, dtype=np.float32)
        FP = np.ascontiguousarray(np.atleast_1d(FP), dtype=np.float64)
        FN = np.ascontiguousarray(np.atleast_1d(FN), dtype=np.float64)
        assert FP.shape == FN.shape and FP.ndim == 1
        E = FP.size
        out = np.empty(E, dtype=np.float64)
        check(self._lib.bnpc_ll_total(self._h, ptr(theta, C.c_float),
            theta.shape[0], ptr(FP, C.c_double), ptr(FN, C.c_double), E,
            ptr(out, C.c_double)), 'll_total')
        return out

    def ll_total_issue(self, theta, FP, FN):
        """Launch ll_total and return; ll_total_wait() picks the result up."""
        theta = np.ascontiguousarray(theta, dtype=np.float32)
        FP = np.ascontiguousarray(np.atleast_1d(FP), dtype=np.float64)
        FN = np.ascontiguousarray(np.atleast_1d(FN), dtype=np.float64)
        assert FP.shape == FN.shape and FP.ndim == 1
        check(self._lib.bnpc_ll_total_issue(self._h, ptr(theta, C.c_float),
            theta.shape[0], ptr(FP, C.c_double), ptr(FN, C.c_double),
            FP.size), 'll_total_issue')
        self._total_E = FP.size

    def ll_total_wait(self):
        out = np.empty(self._total_E, dtype=np.float64)
        check(self._lib.bnpc_ll_total_wait(self._h, ptr(out, C.c_double)),
            'll_total_wait')
        return out

    def bench_ll(self, reps=10):
        """Average duration (ms) of the last ll kernel launch, re-issued."""
        ms = C.c_float(0)
        check(self._lib.bnpc_bench_ll(self._h, reps, C.byref(ms)), 'bench_ll')
        return ms.value

    def bench_ll_full(self, reps=10):
        """Average device time (ms) of the whole last ll evaluation (tables +
        sums + combine), re-issued; call right after ll_theta / ll_tables."""
        ms = C.c_float(0)
        check(self._lib.bnpc_bench_ll_full(self._h, reps, C.byref(ms)),
            'bench_ll_full')
        return ms.value

    def mh_ahead_stats(self):
        """(walkers started, adopted by a batch, rows of draws adopted): the
        parameter batches whose draws were taken ahead on a copy of the
        stream (bnpc_mh_ahead_stats)."""
        b, t, r = _i64(0), _i64(0), _i64(0)
        check(self._lib.bnpc_mh_ahead_stats(self._h, C.byref(b), C.byref(t),
            C.byref(r)), 'mh_ahead_stats')
        return b.value, t.value, r.value

    def launch_timers(self, on):
        """Per-launch device timers (bnpc_launch_timers).  on=True: start;
        on=False: stop, returns (sum of kernel durations in ms, launches)."""
        ms, n = C.c_double(0), C.c_int64(0)
        check(self._lib.bnpc_launch_timers(self._h, 1 if on else 0,
            C.byref(ms), C.byref(n)), 'launch_timers')
        return None if on else (ms.value, n.value)

    def last_launch(self):
        """(kernel name(s), clusters, mutation chunks) of the last ll call."""
        name = C.create_string_buffer(96)
        K, ms = _i64(0), C.c_int(0)
        check(self._lib.bnpc_last_launch(self._h, name, 96, C.byref(K),
            C.byref(ms)), 'last_launch')
        return name.value.decode(), K.value, ms.value

    def timer_start(self):
        check(self._lib.bnpc_timer_start(self._h), 'timer_start')

    def timer_stop(self):
        ms = C.c_float(0)
        check(self._lib.bnpc_timer_stop(self._h, C.byref(ms)), 'timer_stop')
        return ms.value

    def sync(self):
        check(self._lib.bnpc_sync(self._h), 'sync')
