#!/bin/bash
# Round 6: the lane's stride (runs of cells that stay in the cluster that
# dominates them) on / off: the host loop alone on synthetic records
# (tools/hinted_loop_bench.py, host only) and the bench lines, interleaved.
tag=${1:-r06stride}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
for lane in on nostride; do
  export BNPC_SWEEP_LANE=$lane; [ $lane = on ] && unset BNPC_SWEEP_LANE
  echo "== lane $lane" >> $out/hinted_loop_bench.log
  python3 tools/hinted_loop_bench.py 5000 14 400 >> $out/hinted_loop_bench.log 2>&1
  python3 tools/hinted_loop_bench.py 5000 14 400 0.25 0.08 >> $out/hinted_loop_bench.log 2>&1
  python3 tools/hinted_loop_bench.py 10000 20 300 0.2 0.02 >> $out/hinted_loop_bench.log 2>&1
  python3 tools/hinted_loop_bench.py 50000 50 100 0.05 0.003 >> $out/hinted_loop_bench.log 2>&1
  python3 tools/hinted_loop_bench.py 5000 200 300 0.02 0.0 >> $out/hinted_loop_bench.log 2>&1
done
unset BNPC_SWEEP_LANE
cat $out/hinted_loop_bench.log
args="--cpu-steps 0 --sustained-steps 0 --device-steps 0"
for i in 1 2 3; do
  for cfg in c5 c4 c3; do
    steps=100; [ $cfg = c3 ] && steps=200
    python3 bench.py --config $cfg --steps $steps $args > $out/bench_${cfg}_new_$i.json 2> /dev/null
    BNPC_SWEEP_LANE=nostride python3 bench.py --config $cfg --steps $steps $args > $out/bench_${cfg}_nostride_$i.json 2> /dev/null
  done
done
BNPC_TIMING=gibbs python3 bench.py --config c5 --steps 20 --warmup 6 $args > /dev/null 2> $out/trace_c5_gibbs.log
BNPC_TIMING=gibbs python3 bench.py --config c3 --steps 40 --warmup 6 $args > /dev/null 2> $out/trace_c3_gibbs.log
uptime >> $out/box_load.log
for f in $out/bench_*.json; do
    python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
w = j['window']
print('$f'.split('/')[-1], j['value'], 'gibbs', w.get('gibbs', {}).get('ms_per_call'), 'wait', w.get('gibbs_waits_for_device', {}).get('ms_per_call'), j['host'].get('sweep_lane'), j['host'].get('sweep_stride'))"
done
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -n 3 $out/pytest_gpu.log
