#!/bin/bash
# Round 6: does a wider team for the wide parameter batches pay, now that the
# draws no longer pace them?  16 / 32 (default) / 48 / 64 ranks, config 5 and
# c3k, interleaved, 3 lines each.
out=gpurun_out/${1:-r06wide}
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
args="--steps 100 --cpu-steps 0 --sustained-steps 0 --device-steps 0"
for i in 1 2 3; do
  for n in 32 16 48 64; do
    python3 tools/wide_team_run.py $n --config c5 $args > $out/c5_wide${n}_$i.json 2>/dev/null
    python3 tools/wide_team_run.py $n --config c3k $args > $out/c3k_wide${n}_$i.json 2>/dev/null
  done
done
uptime >> $out/box_load.log
for f in $out/*.json; do
    python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
w = j['window']
print('$f'.split('/')[-1], j['value'], 'threads_wide', j['host']['threads_wide_batches'], 'params', w['parameters']['ms_per_call'], 'cpu busy', j['host']['cpu_busy_threads'])"
done
