#!/bin/bash
# bench.py at BASELINE configs[2] with experimental builds of the library (tools/build_variant.py): variant_bench.sh <name...>  ("default" = the in-tree build)
for v in "$@"; do
  echo "== $v"
  if [ "$v" = default ]; then unset RTX_LIB_PATH; else export RTX_LIB_PATH=$PWD/gpurun_scratch/lib_$v.so; fi
  timeout 600 python bench.py --config 2 --no-cpu-baseline --no-extras 2>gpurun_out/variant_$v.err | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(b['value']), round(b['ms_per_step'], 2), {k: round(v, 2) for k, v in b['stage_ms_per_step'].items()})"
done
