#!/bin/bash
RTX_HIT_PAIR=1 python tools/quad_time.py single 40960 2>&1 | tail -3
for v in "$@"; do echo "== $v"; RTX_HIT_PAIR=1 RTX_LIB_PATH=gpurun_scratch/lib_$v.so python tools/quad_time.py single 40960 2>&1 | tail -3; done
