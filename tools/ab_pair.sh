#!/bin/bash
# full GPU suite + A/B of RTX_OPT_HIT_PAIR on the bench workloads (gpurun)
set -u
mkdir -p gpurun_out
RTX_SKIP_5M=1 timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for c in 2 1; do
  for f in "" "--no-pair"; do
    echo "== config $c $f"
    timeout 900 python bench.py --config $c --no-cpu-baseline $f 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(b['value']), round(b['ms_per_step'], 2), {k: round(v, 2) for k, v in b['stage_ms_per_step'].items()}, b['roofline']['requested_bytes_per_query'])"
  done
done
