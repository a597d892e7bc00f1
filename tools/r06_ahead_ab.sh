#!/bin/bash
# Round 6: the draws taken ahead (BNPC_MH_AHEAD) on / off, interleaved on one
# box: bench lines of configs 5, 4, c3k, 3 and the per-part trace of config
# 5's parameter batches either way.
tag=${1:-r06ahead}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
for i in 1 2 3; do
  for a in 1 0; do
    BNPC_MH_AHEAD=$a python3 bench.py --config c5 --steps 100 --cpu-steps 0 --sustained-steps 0 --device-steps 0 > $out/bench_c5_ahead${a}_$i.json 2> /dev/null
    BNPC_MH_AHEAD=$a python3 bench.py --config c4 --steps 100 --cpu-steps 0 --sustained-steps 0 --device-steps 0 > $out/bench_c4_ahead${a}_$i.json 2> /dev/null
  done
done
for a in 1 0; do
    BNPC_MH_AHEAD=$a python3 bench.py --config c3k --steps 100 --cpu-steps 0 --sustained-steps 0 --device-steps 0 > $out/bench_c3k_ahead${a}.json 2> /dev/null
    BNPC_MH_AHEAD=$a python3 bench.py --cpu-steps 0 --sustained-steps 0 --device-steps 0 > $out/bench_c3_ahead${a}.json 2> /dev/null
    BNPC_MH_AHEAD=$a BNPC_TIMING=mh,gibbs,params python3 bench.py --config c5 --steps 20 --warmup 6 --cpu-steps 0 --sustained-steps 0 --device-steps 0 > /dev/null 2> $out/trace_c5_ahead${a}.log
done
uptime >> $out/box_load.log
for f in $out/bench_*.json; do
    python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
w = j['window']
print('$f'.split('/')[-1], j['value'], j['config']['K_end'], 'params', w['parameters']['ms_per_call'], 'gibbs', w.get('gibbs', {}).get('ms_per_call'), j['host'].get('mh_ahead'), j['host']['cpu_busy_threads'])"
done
