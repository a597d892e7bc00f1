#!/usr/bin/env python3
"""cProfile of the BINDING around natively made steps (the whole step is one
ctypes call: what the profile books outside `bnpc_chain_step` is the
interpreter's share - state in / out, tallies, trace bookkeeping).
usage: interpreter_profile.py [config] [steps]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import libs.CRP as dev_fixed  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c5'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
N, M, C, miss, learned = bench.CONFIGS[cfg]
data = bench.synth(0, N, M, C, miss)
np.random.seed(42)
model = bench.make_model(dev_fixed, dev_learn, data, learned)
model.init()
total = steps + 10
chain = bench.new_chain(model, learned, total, cfg)
burn = int(total * .33)
for i in range(1, 11):
    bench.step(chain, i, burn)
prof = cProfile.Profile()
prof.enable()
for i in range(11, total + 1):
    bench.step(chain, i, burn)
prof.disable()
st = pstats.Stats(prof)
st.sort_stats('tottime').print_stats(22)
