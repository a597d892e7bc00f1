#!/bin/bash
# Round 6, first call: the suite and the bench lines of the configs on round
# 5's last code, on this round's boxes (what the A/Bs below are read against).
tag=${1:-r06base}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
nproc >> $out/box_load.log
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -n 3 $out/pytest_gpu.log
for i in 1 2 3; do
    python3 bench.py --cpu-steps 0 > $out/bench_c3_$i.json 2> /dev/null
    python3 bench.py --config c5 --steps 100 --cpu-steps 0 > $out/bench_c5_$i.json 2> /dev/null
done
python3 bench.py --config c4 --steps 100 --cpu-steps 0 > $out/bench_c4_1.json 2> /dev/null
BNPC_TIMING=gibbs,params python3 bench.py --config c5 --steps 20 --warmup 6 --cpu-steps 0 > /dev/null 2> $out/host_phase_trace_c5.log
uptime >> $out/box_load.log
for f in $out/bench_*.json; do
    python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('/')[-1], j['value'], j.get('first_step_s'))"
done
