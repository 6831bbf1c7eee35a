run() { name=$1; shift; timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline "$@" > gpurun_out/ab_$name.json 2>gpurun_out/ab.err; python - <<PY
import json
d=json.load(open("gpurun_out/ab_$name.json"))
tp=d["roofline"].get("tile_pruning",{})
print("$name", round(d["ms_per_step"],2), {k:round(x,2) for k,x in d["stage_ms_per_step"].items() if k in ("hit_count","tile_bounds","tile_prune","prob_table","taxon_prefix")}, {k:round(tp[k],3) for k in tp if 'live' in k})
PY
}
