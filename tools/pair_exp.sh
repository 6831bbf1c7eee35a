#!/bin/bash
# correctness + timing of the pair kernels (gpurun)
set -u
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair" 2>&1 | tail -8
python tools/quad_time.py single 40960 2>&1 | tail -3
RTX_HIT_PAIR=1 python tools/quad_time.py single 40960 2>&1 | tail -3
RTX_HIT_PAIR=2 python tools/quad_time.py single 40960 2>&1 | tail -3
for v in "$@"; do echo "== $v"; RTX_HIT_PAIR=2 RTX_LIB_PATH=gpurun_scratch/lib_$v.so python tools/quad_time.py single 40960 2>&1 | tail -3; done
