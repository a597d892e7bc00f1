#!/bin/bash
# Round 6: moves of 3 and 4 cells made natively too - the chain tests, then
# chains with many small moves against the oracle (split / merge heavy).
out=gpurun_out/${1:-r06small}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -n 3 $out/pytest_gpu.log
run() { name=$1; shift; python3 tools/parity_soak.py "$@" > $out/soak_$name.log 2>&1 & }
run c2_1500_smp09          c2 1500 301 0.9
run c2_1500_smp09_merge    c2 1500 302 0.9 ratios=.3,.7
run c2_1500_smp07_learned  c2 1500 303 0.7 data=41 learned=1
run c3_500_smp08           c3 500 304 0.8
run c3_500_smp08_merge     c3 500 305 0.8 ratios=.4,.6 data=42
run k150_400_smp08         k150 400 306 0.8
run c2_800_uniform_smp09   c2 800 307 0.9 beta=1,1 data=43
run c2_800_5scans_smp09    c2 800 308 0.9 sm_steps=5 data=44
wait
tail -q -n 1 $out/soak_*.log
python3 bench.py --cpu-steps 0 --device-steps 0 > $out/bench_c3.json 2>/dev/null
python3 -c "
import json
j = json.loads(open('$out/bench_c3.json').read().strip().splitlines()[-1])
s = j['sustained']; print(j['value'], s['steps_s'], s['host'], {k: v['calls'] for k, v in s['phases'].items() if isinstance(v, dict) and 'calls' in v}, s['phases']['other'])"
