#!/bin/bash
# Round 6: this build against the one before it (tools/ab/libbnpc_prev.so),
# interleaved on one box; then the shared-GPU experiment's traces; then the
# GPU suite.
tag=${1:-r06tail}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
prev=$GRAFT_REPO_ROOT/tools/ab/libbnpc_prev.so
args="--cpu-steps 0 --sustained-steps 0 --device-steps 0"
for i in 1 2 3; do
  for cfg in c5 c4 c3k c3; do
    steps=100; [ $cfg = c3 ] && steps=200
    python3 bench.py --config $cfg --steps $steps $args > $out/bench_${cfg}_new_$i.json 2> /dev/null
    BNPC_LIB=$prev python3 bench.py --config $cfg --steps $steps $args > $out/bench_${cfg}_prev_$i.json 2> /dev/null
  done
done
uptime >> $out/box_load.log
for f in $out/bench_*.json; do
    python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
w = j['window']
print('$f'.split('/')[-1], j['value'], j['config']['K_end'], 'params', w['parameters']['ms_per_call'], 'gibbs', w.get('gibbs', {}).get('ms_per_call'), j['host'].get('mh_ahead'), j['host']['cpu_busy_threads'])"
done
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -n 3 $out/pytest_gpu.log
bash tools/r06_shared_gpu.sh ${tag}_shared > $out/shared.log 2>&1
tail -n 30 $out/shared.log
