#!/bin/bash
# times hit_count of the default build and of experimental builds in gpurun_scratch/ (tools/quad_variants.sh)
set -u
python tools/quad_time.py single 40960 2>&1 | tail -1
for v in "$@"; do
  echo "== $v"
  RTX_LIB_PATH=gpurun_scratch/lib_$v.so python tools/quad_time.py single 40960 2>&1 | tail -1
done
