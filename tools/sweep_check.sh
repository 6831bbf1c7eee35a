#!/bin/bash
# headline + divergence sweep of bench.py (configs[2]) in one line each: sweep_check.sh
timeout 900 python bench.py --no-cpu-baseline 2>gpurun_out/sweep_check.err | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(b['value']), round(b['ms_per_step'], 2), {k: round(v, 2) for k, v in b['stage_ms_per_step'].items()})
print('incl_h2d', round(b['value_incl_h2d']['value']), 'e2e', round(b['value_end_to_end']['value']), 'unpruned', round(b['value_unpruned']['value']))
for x in b['divergence_sweep']['rows']: print(x['mu_q'], round(x['value']), 'unpruned', round(x['value_unpruned']), 'live/query', round(x['live_tiles_per_query'], 2))"
