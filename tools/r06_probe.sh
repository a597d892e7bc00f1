#!/bin/bash
tag=${1:-r06probe}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 tools/interpreter_profile.py c5 80 > $out/interpreter_profile_c5.log 2>&1
head -n 45 $out/interpreter_profile_c5.log
BNPC_TIMING=gibbs,params python3 bench.py --steps 200 --warmup 10 --cpu-steps 0 --sustained-steps 0 --device-steps 0 > $out/bench_c3_traced.json 2> $out/trace_c3_window.log
python3 tools/trace_means.py $out/trace_c3_window.log $out/bench_c3_traced.json
BNPC_TIMING=step python3 bench.py --config c5 --steps 60 --warmup 10 --cpu-steps 0 --sustained-steps 0 --device-steps 0 > $out/bench_c5_steptrace.json 2> $out/trace_c5_step.log
python3 - <<PY
import re, numpy as np
rows = [tuple(float(x) for x in re.findall(r'([0-9.]+) (?:us in the call|in its phases)', l)) for l in open('$out/trace_c5_step.log') if l.startswith('[step]')]
a = np.array(rows[10:])
print('config 5, native steps:', len(a), 'mean us in the call', a[:, 0].mean().round(1), 'in its phases', a[:, 1].mean().round(1))
import json
j = json.loads(open('$out/bench_c5_steptrace.json').read().strip().splitlines()[-1])
print('line: ms_per_step', j['ms_per_step'], 'other', j['window']['other'])
PY
