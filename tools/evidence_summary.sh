#!/bin/bash
# One line per bench line of an evidence set (tools/r06_evidence.sh):
#   tools/evidence_summary.sh gpurun_out/r06ev6
cd $1 || exit 1
cat box_load.log | tr '\n' ' '; echo
for f in bench_driver_form bench_final_1 bench_final_2 bench_final_3 bench_config2 bench_c3k bench_c3k_1 bench_c3k_2 bench_c3k_3 bench_k150 bench_config4 bench_c4_1 bench_c4_2 bench_c4_3 bench_config5 bench_c5_1 bench_c5_2 bench_c5_3 bench_c5_ahead0_1 bench_c5_ahead0_2 bench_c5_ahead0_3 bench_c5_nostride_1 bench_c5_nostride_2 bench_c5_nostride_3 bench_threads1_1 bench_threads1_2 bench_threads1_3 bench_step_by_methods bench_without_the_stride bench_fallbacks bench_under_rocprof; do python3 -c "
import json
try:
    j = json.loads(open('$f.json').read().strip().splitlines()[-1])
    s = j.get('sustained') or {}
    c = j.get('cpu_baseline') or {}
    p = j.get('parity_check') or {}
    w = j['window']
    print('$f', j['value'], 'sust', s.get('steps_s'), 'K', j['config'].get('K_end'), 'traffic', (j.get('roofline') or {}).get('traffic'), 'cpu', c.get('value'), 'x', j.get('speedup_vs_cpu_baseline'), 'parity', p.get('assignments_identical'), p.get('ml_max_rel'), 'dev', w.get('device_ms_per_step'), w.get('device_busy_frac'), 'other', w['other']['ms_per_step'], 'first', j.get('first_step_s'))
except Exception as e: print('$f', 'ERR', e)
"; done
