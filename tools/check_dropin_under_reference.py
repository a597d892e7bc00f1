#!/opt/conda/bin/python3.9
"""Build-container check of the drop-in claim: the
REFERENCE's own, unmodified libs/MCMC.py drives this repo's model classes.

    LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6 \\
        /opt/conda/bin/python3.9 tools/check_dropin_under_reference.py

(conda's older libstdc++ would otherwise shadow the one libamdhip64 needs)

A scratch package tree is assembled in a temp dir: libs/MCMC.py, dpmmIO.py,
utils.py, plotting.py are symlinks into /root/reference; libs/CRP.py and
libs/CRP_learning_errors.py are this repo's drop-ins.  There is no GPU in the
build container, so the device primitives are served by the NumPy stand-in of
the test-suite (tests/fake_device.py); everything else - class surface,
pickling, RNG order, native sweeps - is the product code.  The trajectory is
compared with the golden vectors captured from the pure reference.
"""
import contextlib
import io
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

tmp = tempfile.mkdtemp(prefix='dropin_')
os.makedirs(os.path.join(tmp, 'libs'))
for name in ('__init__.py', 'MCMC.py', 'dpmmIO.py', 'utils.py', 'plotting.py'):
    os.symlink(os.path.join(REF, 'libs', name), os.path.join(tmp, 'libs', name))
for name in ('CRP.py', 'CRP_learning_errors.py'):
    shutil.copy(os.path.join(ROOT, 'libs', name), os.path.join(tmp, 'libs', name))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
sys.path.insert(0, tmp)            # `libs` resolves to the scratch tree

from bnpc_amd import _lib  # noqa: E402
from fake_device import FakeContext  # noqa: E402
_lib.Context = FakeContext
try:
    _lib.load()
except (OSError, RuntimeError) as exc:
    # conda's libstdc++ shadows the one libbnpc_hip.so was linked against
    shutil.rmtree(tmp)
    sys.exit(f'cannot load {_lib.LIB_PATH}: {exc}\n'
        'run this tool as\n    LD_PRELOAD=/usr/lib/x86_64-linux-gnu/'
        'libstdc++.so.6 /opt/conda/bin/python3.9 '
        'tools/check_dropin_under_reference.py')

import libs.MCMC as ref_mcmc  # noqa: E402  (the reference's driver)
import libs.dpmmIO as ref_io  # noqa: E402
import libs.CRP  # noqa: E402
import libs.CRP_learning_errors  # noqa: E402

assert os.path.realpath(ref_mcmc.__file__).startswith(REF)
assert libs.CRP.CRP.__module__ == 'libs.CRP'
print('driver  :', os.path.realpath(ref_mcmc.__file__))
print('model   :', libs.CRP.CRP.__mro__[1].__module__)

gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'trajectories.npz'))
data = ref_io.load_data(os.path.join(REF, 'example_data', 'data.csv'),
    transpose=True)
bad = 0
for kind in ('fixed', 'learn'):
    if kind == 'fixed':
        model = libs.CRP.CRP(data, DP_alpha=[-1, -1], param_beta=[.25, .25],
            FN_error=0.1, FP_error=0.001)
        eup = 0
    else:
        model = libs.CRP_learning_errors.CRP_errors_learning(data,
            DP_alpha=[-1, -1], param_beta=[.25, .25], FP_mean=0.01,
            FP_sd=0.01, FN_mean=0.2, FN_sd=0.1)
        eup = .25
    mcmc = ref_mcmc.MCMC(model, sm_prob=.33, dpa_prob=.25, error_prob=eup,
        sm_ratios=[.75, .25], sm_steps=3)
    with contextlib.redirect_stdout(io.StringIO()):
        mcmc.run((200, 66), 42, 1, 0, '', True)
    res = mcmc.get_results()[0]
    same = np.array_equal(res['assignments'], gold[f'ex_{kind}_assignments'])
    rel = np.max(np.abs(res['ML'] / gold[f'ex_{kind}_ML'] - 1))
    learning = mcmc.chains[0].learning_errors
    print(f'{kind:5s}: reference driver, learning_errors={learning}, '
        f'assignments identical over 200 steps: {same}, '
        f'max rel ML diff {rel:.2e}')
    bad += (not same) or rel > 1e-6 or learning != (kind == 'learn')
shutil.rmtree(tmp)
print('DROP-IN OK' if not bad else 'DROP-IN FAILED')
sys.exit(1 if bad else 0)
