#!/bin/bash
out=gpurun_out/${1:-r06sus}
mkdir -p $out
cd $GRAFT_REPO_ROOT
for c in c4 c5 c3; do
  python3 bench.py --config $c --steps 100 --cpu-steps 0 --device-steps 0 > $out/bench_$c.json 2>/dev/null
  python3 -c "
import json
j = json.loads(open('$out/bench_$c.json').read().strip().splitlines()[-1])
w, s = j['window'], j['sustained']
f = lambda d: {k: (d[k]['ms_per_call'], d[k]['calls']) for k in d if isinstance(d[k], dict) and 'ms_per_call' in d[k]}
print('$c window', j['ms_per_step'], f(w), w['other'])
print('$c sustained', s['ms_per_step'], s['K_end'], f(s['phases']), s['phases']['other'])
print(j['host'])
"
done
