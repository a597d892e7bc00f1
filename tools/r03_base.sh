out=gpurun_out/r03base; mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/gputests.log 2>&1; tail -3 $out/gputests.log
python3 bench.py --steps 200 --cpu-steps 0 > $out/bench_default.json 2>$out/bench_default.err
BNPC_HOST_THREADS=4 python3 bench.py --steps 200 --cpu-steps 0 > $out/bench_t4.json 2>/dev/null
BNPC_HOST_THREADS=1 python3 bench.py --steps 200 --cpu-steps 0 > $out/bench_t1.json 2>/dev/null
python3 tools/profile_steps.py c3 200 > $out/host_breakdown_default.log 2>&1
BNPC_HOST_THREADS=4 python3 tools/profile_steps.py c3 200 > $out/host_breakdown_t4.log 2>&1
BNPC_HOST_THREADS=1 python3 tools/profile_steps.py c3 200 > $out/host_breakdown_t1.log 2>&1
python3 -c "
import json
for t in ('default','t4','t1'):
    d=json.load(open('$out/bench_%s.json'%t)); print(t, d['value'], d['ms_per_step'], d['host'])
"
nproc; uptime
