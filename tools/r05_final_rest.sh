#!/bin/bash
# The files of tools/r05_evidence.sh that tools/r05_final_set.sh does not
# repeat - fall-back lines, host-side traces, kernel statistics of configs 5
# and c3k, microbenchmarks, the first sweep - on the round's last code.
out=gpurun_out/${1:-r05ev10}
mkdir -p $out
uptime > $out/box_load.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BNPC_STREAM_LIVE=0 BNPC_NATIVE_MH=0 BNPC_NATIVE_BETA=0 python3 bench.py --steps 100 --cpu-steps 0 > $out/bench_fallbacks.json 2>/dev/null
BNPC_MH_SCREEN=2 python3 bench.py --cpu-steps 0 > $out/bench_screen_without_theta.json 2>/dev/null
BNPC_SWEEP_LANE=0 python3 bench.py --cpu-steps 0 > $out/bench_without_the_lane.json 2>/dev/null
python3 tools/python_overhead.py c3 300 > $out/python_overhead.log 2>&1
python3 tools/mh_dev_trace.py c3k 12 > /dev/null 2> $out/mh_screen_trace_c3k.log
python3 tools/mh_dev_trace.py c5 12 > /dev/null 2> $out/mh_screen_trace_c5.log
BNPC_TIMING=move python3 bench.py --steps 60 --warmup 10 --cpu-steps 0 2>&1 >/dev/null | grep '^\[move\]' > $out/move_trace_c3.log
BNPC_TIMING=move python3 bench.py --config c5 --steps 30 --warmup 10 --cpu-steps 0 2>&1 >/dev/null | grep '^\[move\]' > $out/move_trace_c5.log
rocprofv3 --kernel-trace --stats -d $out/prof_c5 -o c5 -f csv -- \
    python3 bench.py --config c5 --steps 30 --warmup 10 --cpu-steps 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $out/prof_c3k -o c3k -f csv -- \
    python3 bench.py --config c3k --steps 50 --warmup 10 --cpu-steps 0 > /dev/null 2>&1
python3 tools/ll_microbench.py > $out/ll_microbench.md 2>&1
python3 tools/msplit_sweep.py > $out/msplit_sweep.md 2>&1
python3 tools/first_sweep_profile.py c5 > $out/first_sweep_c5.log 2>&1
uptime >> $out/box_load.log
find $out -name "*_trace.csv" -size +4M -delete
for f in bench_fallbacks bench_screen_without_theta bench_without_the_lane; do python3 -c "
import json
j = json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', j['value'])"; done
tail -2 $out/python_overhead.log; tail -3 $out/first_sweep_c5.log; cat $out/box_load.log
