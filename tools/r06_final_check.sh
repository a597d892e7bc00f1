#!/bin/bash
# Round 6, last call: what the driver runs at round end, on the final tree -
# the GPU suite, smoke(), the bench line in the driver's form, and
# `bench.py --gpus 2` as a command.
out=gpurun_out/${1:-r06final}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -n 3 $out/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.log 2>&1
tail -n 4 $out/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err
python3 bench.py --gpus 2 --steps 50 --warmup 10 --cpu-steps 0 > $out/bench_gpus2.json 2> $out/bench_gpus2.err
python3 -c "
import json
for f in ('$out/bench_driver_form.json', '$out/bench_gpus2.json'):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], j['value'], 'ranks', j['ranks'], 'n_gpus', j['n_gpus'], 'sustained', j['sustained'], 'traffic', j['roofline']['traffic'], 'parity', j['parity_check'] and j['parity_check']['assignments_identical'], 'device', j['window'].get('device_ms_per_step'))
"
