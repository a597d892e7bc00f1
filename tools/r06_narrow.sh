#!/bin/bash
# Round 6: the narrow last cluster group - its test, the launch at config 5's
# size with / without it, bench lines of config 5 either way, the suite.
tag=${1:-r06narrow}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "narrow_last or every_cluster_tiling or ll_tables_is_bit_exact" > $out/pytest_narrow.log 2>&1
tail -n 5 $out/pytest_narrow.log
tools/ubench/h2d_probe > $out/h2d_probe.log 2>&1; cat $out/h2d_probe.log
args="--cpu-steps 0 --sustained-steps 0 --device-steps 0"
for i in 1 2 3; do
    python3 bench.py --config c5 --steps 100 $args > $out/bench_c5_narrow_$i.json 2> /dev/null
    BNPC_KW=8 python3 bench.py --config c5 --steps 100 $args > $out/bench_c5_groups8_$i.json 2> /dev/null
done
BNPC_TIMING=gibbs python3 bench.py --config c5 --steps 20 --warmup 6 $args > /dev/null 2> $out/trace_c5_gibbs.log
BNPC_TIMING=gibbs python3 bench.py --config c3 --steps 40 --warmup 6 $args > /dev/null 2> $out/trace_c3_gibbs.log
tail -n 4 $out/trace_c5_gibbs.log $out/trace_c3_gibbs.log
for f in $out/bench_*.json; do
    python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
w = j['window']
print('$f'.split('/')[-1], j['value'], j['config']['K_end'], 'gibbs', w.get('gibbs', {}).get('ms_per_call'), 'wait', w.get('gibbs_waits_for_device', {}).get('ms_per_call'), j['roofline_converged']['kernel'], j['roofline_converged']['launch_ms'])"
done
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -n 3 $out/pytest_gpu.log
uptime >> $out/box_load.log
