#!/bin/bash
# Round 6: ranks of bench.py sharing the ONE GPU
# of a box - the only hardware proxy there is for the host settings of a
# multi-GPU node.  bench.py --gpus N starts its N ranks itself.  At 1 / 2 / 4 /
# 8 ranks: the settings the placement rule picks (greedy while a chain has >=
# 8 logical CPUs of its node) against the frugal ones forced
# (BNPC_HOST_SPIN_US=5: what round 5 gave every chain that shared a node).
# Then device against host at 8 ranks: a kernel trace of ONE rank alone and of
# one rank next to 7 others - queueing shows as kernels that start later after
# they were enqueued, not as longer kernels.
tag=${1:-r06shared}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
args="--cpu-steps 0 --sustained-steps 0 --device-steps 0 --steps 200"
: > $out/bench_ranks_sharing_one_gpu.jsonl
for n in 1 2 4 8; do
  for mode in rule frugal; do
    if [ $mode = frugal ]; then export BNPC_HOST_SPIN_US=5; else unset BNPC_HOST_SPIN_US; fi
    python3 bench.py --gpus $n $args 2> /dev/null | python3 -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'ranks': j['ranks'], 'n_gpus': j['n_gpus'], 'mode': '$mode', 'value': j['value'], 'per_rank_steps_s': j['per_rank_steps_s'], 'host_threads': j['host']['threads'], 'cpu_busy_threads_rank0': j['host']['cpu_busy_threads'], 'ms_per_step': j['ms_per_step']}))" >> $out/bench_ranks_sharing_one_gpu.jsonl
  done
done
unset BNPC_HOST_SPIN_US
cat $out/bench_ranks_sharing_one_gpu.jsonl
# device or host?  one rank traced alone ...
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$out/trace_alone -o alone -- python3 $GRAFT_REPO_ROOT/bench.py $args > $GRAFT_REPO_ROOT/$out/bench_traced_alone.json 2> /dev/null
# ... and next to 7 untraced ranks' worth of load (7 independent chains on the
# same GPU, started first; the traced one runs while they do)
for r in 1 2 3 4 5 6 7; do
  (cd $GRAFT_REPO_ROOT && BNPC_HOST_SHARE=8 BNPC_HOST_SPIN_US=5 python3 bench.py --cpu-steps 0 --sustained-steps 0 --device-steps 0 --steps 6000 > /dev/null 2>&1) &
done
sleep 4
BNPC_HOST_SHARE=8 BNPC_HOST_SPIN_US=5 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$out/trace_with7 -o with7 -- python3 $GRAFT_REPO_ROOT/bench.py $args > $GRAFT_REPO_ROOT/$out/bench_traced_with7.json 2> /dev/null
wait
cd $GRAFT_REPO_ROOT
python3 tools/queueing_from_trace.py $out/trace_alone $out/trace_with7 > $out/queueing.txt 2>&1
cat $out/queueing.txt
# keep the summaries, drop the raw traces beyond a few MB
find $out -name "*_kernel_trace.csv" -size +20M -delete
uptime >> $out/box_load.log
