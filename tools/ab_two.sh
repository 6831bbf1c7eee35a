#!/bin/bash
# headline and the 5 % row of the divergence sweep for an A/B of builds / options: tools/ab_two.sh <name> [bench args]
name=$1; shift
for mu in 0.02 0.05; do
  if [ $mu = 0.02 ]; then Q=""; else Q="--queries 131072 --mu-q $mu --exact-frac 0"; fi
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline $Q "$@" > gpurun_out/abs_$name.json 2>gpurun_out/ab.err
  python - <<PY
import json
d=json.load(open("gpurun_out/abs_$name.json"))
tp=d["roofline"].get("tile_pruning",{})
print("$name mu=$mu", round(d["value"]), round(d["ms_per_step"],2), {k:round(x,2) for k,x in d["stage_ms_per_step"].items() if k in ("hit_count","tile_bounds","tile_prune")}, round(tp.get("live_tiles_per_pair_first_stage",0),2))
PY
done
