#!/bin/bash
# Round 6: the first write of the large sample traces on the host team -
# microbenchmark by team size, then config 5 / c3k / config 4 lines.
out=gpurun_out/${1:-r06alloc}
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
python3 tools/trace_alloc_bench.py > $out/trace_alloc_bench.log 2>&1
python3 tools/trace_alloc_bench.py 74 212 1000 >> $out/trace_alloc_bench.log 2>&1
python3 tools/trace_alloc_bench.py 141 22 1000 >> $out/trace_alloc_bench.log 2>&1
cat $out/trace_alloc_bench.log
args="--steps 100 --cpu-steps 0 --sustained-steps 0 --device-steps 0"
for i in 1 2 3; do
  for cfg in c5 c3k c4; do
    python3 bench.py --config $cfg $args > $out/bench_${cfg}_$i.json 2>/dev/null
  done
done
python3 bench.py --config c5 --steps 100 --cpu-steps 0 --device-steps 0 > $out/bench_c5_sustained.json 2>/dev/null
uptime >> $out/box_load.log
for f in $out/bench_*.json; do
    python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
w = j['window']
print('$f'.split('/')[-1], j['value'], 'other', w['other']['ms_per_step'], 'record', w['record']['ms_per_call'], 'sustained', (j.get('sustained') or {}).get('steps_s'), ((j.get('sustained') or {}).get('phases') or {}).get('other'))"
done
