#!/bin/bash
out=gpurun_out/${1:-r06py}
mkdir -p $out
cd $GRAFT_REPO_ROOT
uptime > $out/box_load.log
python3 tools/python_overhead.py c3 300 > $out/python_overhead_c3.log 2>&1
tail -n 2 $out/python_overhead_c3.log
for i in 1 2 3; do python3 bench.py --cpu-steps 0 --device-steps 0 > $out/bench_c3_$i.json 2>/dev/null; done
python3 bench.py --config c5 --steps 100 --cpu-steps 0 --device-steps 0 > $out/bench_c5_1.json 2>/dev/null
for f in $out/bench_*.json; do python3 -c "
import json
j = json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('/')[-1], j['value'], j['sustained']['steps_s'], 'other', j['window']['other'], j['sustained']['phases']['other'])"; done
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -n 3 $out/pytest_gpu.log
