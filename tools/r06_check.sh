#!/bin/bash
# Round 6: the GPU suite + the bench line as the driver runs it + the same
# command with --gpus 2 (it starts its two ranks itself; one GPU here).
tag=${1:-r06chk}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
timeout 2400 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -n 5 $out/pytest_gpu.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err
python3 bench.py > $out/bench_c3.json 2> $out/bench_c3.err
python3 bench.py --gpus 2 --cpu-steps 0 > $out/bench_gpus2.json 2> $out/bench_gpus2.err
python3 bench.py --config c5 --steps 100 --cpu-steps 0 > $out/bench_c5.json 2> $out/bench_c5.err
uptime >> $out/box_load.log
for f in $out/bench_*.json; do
    python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
w = j['window']
print('$f'.split('/')[-1], j['value'], j['ranks'], j['n_gpus'], j['sustained'], {k: w.get(k) for k in ('device_ms_per_step', 'launches_per_step', 'device_busy_frac', 'ms_per_step_with_timers')}, j['host'].get('mh_left_to_host'))"
done
tail -n 3 $out/*.err
