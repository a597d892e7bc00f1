#!/bin/bash
# The round's last measurement set after the last change of the library's
# sources: tools/r05_evidence_refresh.sh (PMC passes, the lines that quote
# them), the bench lines of the other configs, then both soak sets - one
# command in one lease of a GPU box.  Writes under gpurun_out/$1.
tag=${1:-r05ev6}
out=gpurun_out/$tag
bash tools/r05_evidence_refresh.sh $tag > $out.refresh.log 2>&1
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do BNPC_HOST_THREADS=1 python3 bench.py --cpu-steps 0 > $out/bench_threads1_$i.json 2>/dev/null; done
python3 bench.py --config c2 --steps 200 > $out/bench_config2.json 2> /dev/null
python3 bench.py --config c3k --steps 100 --cpu-steps 3 > $out/bench_c3k.json 2> /dev/null
python3 bench.py --config k150 --steps 200 --cpu-steps 6 > $out/bench_k150.json 2> /dev/null
python3 bench.py --config c4 --steps 100 --cpu-steps 4 > $out/bench_config4.json 2> /dev/null
for c in c3 c3k c4 c5; do
    BNPC_TIMING=gibbs,params python3 bench.py --config $c --steps 20 --warmup 6 --cpu-steps 0 > /dev/null 2> $out/host_phase_trace_$c.log
done
uptime >> $out/box_load.log
if [ "${SOAKS:-1}" != 0 ]; then     # (SOAKS=0: the measurements only)
    bash tools/r05_soaks.sh > $out.soaks.log 2>&1
    bash tools/r05_soaks_more.sh > $out.soaks_more.log 2>&1
    uptime >> $out/box_load.log
    tail -n 1 gpurun_out/r05soak/*.log gpurun_out/r05soak2/*.log | grep "first diverging"
fi
for f in $out/bench_final_?.json $out/bench_config?.json $out/bench_c3k.json $out/bench_k150.json $out/bench_threads1_?.json $out/bench_step_by_methods.json; do
    python3 -c "
import json, sys
j = json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('/')[-1], j['value'], j.get('first_step_s'), (j.get('parity_check') or {}).get('assignments_identical'), (j.get('cpu_baseline') or {}).get('value'))"
done
