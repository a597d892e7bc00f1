#!/bin/bash
# Round 6: bench lines of the configs (3 each) + where the interpreter's share
# of a config-5 step goes.
tag=${1:-r06lines}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
args="--cpu-steps 0 --sustained-steps 0 --device-steps 0"
for i in 1 2 3; do
  for cfg in c5 c4 c3k c3; do
    steps=100; [ $cfg = c3 ] && steps=200
    python3 bench.py --config $cfg --steps $steps $args > $out/bench_${cfg}_$i.json 2> /dev/null
  done
done
python3 tools/python_overhead.py c5 100 > $out/python_overhead_c5.log 2>&1
python3 tools/python_overhead.py c3 300 > $out/python_overhead_c3.log 2>&1
uptime >> $out/box_load.log
for f in $out/bench_*.json; do
    python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
w = j['window']
print('$f'.split('/')[-1], j['value'], j['config']['K_end'], 'params', w['parameters']['ms_per_call'], 'gibbs', w.get('gibbs', {}).get('ms_per_call'), 'record', w['record']['ms_per_call'], 'other', w['other']['ms_per_step'])"
done
tail -n 12 $out/python_overhead_c5.log
for c in c5 c4; do
    BNPC_TIMING=gibbs,params python3 bench.py --config $c --steps 100 --warmup 10 $args > $out/bench_traced_$c.json 2> $out/host_phase_trace_$c.log
    python3 tools/trace_means.py $out/host_phase_trace_$c.log $out/bench_traced_$c.json > $out/host_phase_means_$c.txt 2>&1
    cat $out/host_phase_means_$c.txt
done
uptime >> $out/box_load.log
