import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
import libs.CRP as dev_fixed, libs.CRP_learning_errors as dev_learn
cfg = sys.argv[1]
N, M, C, miss, learned = bench.CONFIGS[cfg]
data = bench.synth(0, N, M, C, miss)
np.random.seed(42)
model = bench.make_model(dev_fixed, dev_learn, data, learned)
model.init()
steps = int(sys.argv[2])
chain = bench.new_chain(model, learned, steps + 10, cfg)
for i in range(1, 11):
    bench.step(chain, i, 0)
model._hint_used = 0
sweeps = [0]
real = model.update_assignments_Gibbs
def wrapped():
    sweeps[0] += 1
    before = model._hint_used
    real()
    per.append(N - (model._hint_used - before))
per = []
model.update_assignments_Gibbs = wrapped
for i in range(11, steps + 11):
    bench.step(chain, i, 0)
per = np.array(per)
print(cfg, 'sweeps', sweeps[0], 'undecided cells per sweep: mean', per.mean(), 'median', np.median(per), 'max', per.max(), 'sweeps with none', (per == 0).sum())
