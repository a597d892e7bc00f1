#!/bin/bash
# config 3: the walker for its parameter batch too (BNPC_MH_AHEAD=2: any batch
# size; the default starts at 16 384 entries, config 3 has 10-16 thousand)
out=gpurun_out/${1:-r06c3any}
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
args="--cpu-steps 0 --sustained-steps 0 --device-steps 0"
for i in 1 2 3 4 5; do
    python3 bench.py $args > $out/bench_c3_default_$i.json 2> /dev/null
    BNPC_MH_AHEAD=2 python3 bench.py $args > $out/bench_c3_any_$i.json 2> /dev/null
done
python3 bench.py --config c2 --steps 200 $args > $out/bench_c2_default.json 2> /dev/null
BNPC_MH_AHEAD=2 python3 bench.py --config c2 --steps 200 $args > $out/bench_c2_any.json 2> /dev/null
uptime >> $out/box_load.log
for f in $out/bench_*.json; do
    python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
w = j['window']
print('$f'.split('/')[-1], j['value'], 'params', w['parameters']['ms_per_call'], 'gibbs', w['gibbs']['ms_per_call'], j['host'].get('mh_ahead'), j['host']['cpu_busy_threads'])"
done
