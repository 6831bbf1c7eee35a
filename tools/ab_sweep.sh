#!/bin/bash
# headline + the 5 % and 10 % rows of the divergence sweep (131 072 queries) for an A/B of builds / options: tools/ab_sweep.sh <name> [bench args]
name=$1; shift
for mu in 0.02 0.05 0.10; do
  if [ $mu = 0.02 ]; then Q=""; else Q="--queries 131072 --mu-q $mu --exact-frac 0"; fi
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline $Q "$@" > gpurun_out/abs_$name.json 2>gpurun_out/ab.err
  python - <<PY
import json
d=json.load(open("gpurun_out/abs_$name.json"))
tp=d["roofline"].get("tile_pruning",{})
print("$name mu=$mu", round(d["value"]), round(d["ms_per_step"],2), {k:round(x,2) for k,x in d["stage_ms_per_step"].items() if k in ("hit_count","tile_bounds","tile_prune")}, {k:round(tp[k],2) for k in tp if 'live' in k})
PY
done
