#!/bin/bash
# Round 6, the last code: bench.py --gpus 1 / 2 / 4 / 8 as the driver would
# run it (the command starts its ranks itself) on the ONE GPU of a box.
out=gpurun_out/${1:-r06ranks}
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
args="--cpu-steps 0 --sustained-steps 0 --device-steps 0 --steps 200"
: > $out/bench_ranks_sharing_one_gpu_last_code.jsonl
for n in 1 2 4 8; do
    timeout 150 python3 bench.py --gpus $n $args 2> /dev/null | python3 -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'ranks': j['ranks'], 'n_gpus': j['n_gpus'], 'value': j['value'], 'per_rank_steps_s': j['per_rank_steps_s'], 'host_settings': j.get('host_settings') or j['host'].get('settings'), 'cpu_busy_threads_rank0': j['host']['cpu_busy_threads'], 'ms_per_step': j['ms_per_step']}))" >> $out/bench_ranks_sharing_one_gpu_last_code.jsonl
done
uptime >> $out/box_load.log
cat $out/bench_ranks_sharing_one_gpu_last_code.jsonl
