#!/bin/bash
# A/B of RTX_OPT_TILE_SKIP on the bench workloads (gpurun): bash tools/ab_tile_skip.sh
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_skip or packed or fuzz or quad or shard" 2>&1 | tail -5
for c in 2 1; do
  for f in "" "--no-tile-skip"; do
    echo "== config $c $f"
    timeout 900 python bench.py --config $c --no-cpu-baseline $f 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(b['value']), round(b['ms_per_step'], 2), {k: round(v, 2) for k, v in b['stage_ms_per_step'].items()})"
  done
done
