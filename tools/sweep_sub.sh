#!/bin/bash
# bench configs at several sub-batch sizes (gpurun): sweep_sub.sh <config> <sizes...>
c=$1; shift
for sb in "$@"; do
  echo "== config $c sub-batch $sb"
  timeout 900 python bench.py --config $c --no-cpu-baseline --no-extras --sub-batch $sb 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(b['value']), round(b['ms_per_step'], 2), {k: round(v, 2) for k, v in b['stage_ms_per_step'].items()}, b['config']['sub_batch'], round(b['roofline']['requested_bytes_per_query']))"
done
