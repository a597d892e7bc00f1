#!/bin/bash
# Round 6: the rows of a restricted scan's batch taken ahead (this build)
# against the build before it, interleaved; the chain tests first.
tag=${1:-r06scan}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "taken_ahead or native_moves or native_steps or fused_restricted or split_merge or config5_moves or fallback_switches" > $out/pytest_moves.log 2>&1
tail -n 3 $out/pytest_moves.log
prev=$GRAFT_REPO_ROOT/tools/ab/libbnpc_prev.so
args="--cpu-steps 0 --sustained-steps 0 --device-steps 0"
for i in 1 2 3 4; do
  for cfg in c5 c4 c3; do
    steps=100; [ $cfg = c3 ] && steps=200
    python3 bench.py --config $cfg --steps $steps $args > $out/bench_${cfg}_new_$i.json 2> /dev/null
    BNPC_LIB=$prev python3 bench.py --config $cfg --steps $steps $args > $out/bench_${cfg}_prev_$i.json 2> /dev/null
  done
done
BNPC_TIMING=move python3 bench.py --steps 60 --warmup 10 $args 2>&1 >/dev/null | grep '^\[move\]' > $out/move_trace_c3.log
BNPC_TIMING=move python3 bench.py --config c5 --steps 30 --warmup 10 $args 2>&1 >/dev/null | grep '^\[move\]' > $out/move_trace_c5.log
uptime >> $out/box_load.log
for f in $out/bench_*.json; do
    python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
w = j['window']
mv = [w[k]['ms_per_call'] for k in ('split_accepted', 'split_rejected', 'merge_accepted', 'merge_rejected') if k in w]
print('$f'.split('/')[-1], j['value'], 'moves', mv, 'params', w['parameters']['ms_per_call'], j['host'].get('mh_ahead'))"
done
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -n 3 $out/pytest_gpu.log
