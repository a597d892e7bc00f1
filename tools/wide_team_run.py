"""bench.py with the wide parameter batches' team pinned to N ranks and the
small jobs' team left alone (BNPC_HOST_THREADS would pin both):
    python3 tools/wide_team_run.py N <bench.py arguments>
A measurement aid - nothing in the package reads it."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bnpc_amd import _lib  # noqa: E402

n = int(sys.argv.pop(1))
small = _lib.threads_for


def pinned(elements):
    return n if elements >= _lib.MH_WIDE_FROM else small(elements)


_lib.threads_for = pinned
sys.argv[0] = 'bench.py'
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
