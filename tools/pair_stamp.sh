#!/bin/bash
# Phase stamps of hit_count_pair_kernel (builds with -DRTX_PAIR_STAMP=k, tools/build_variant.py stamp<k>): s_memtime ticks per live
# (pair, tile) block, one phase per build: 1 prologue, 2 row loop, 3 epilogue A, 4 epilogue B; inside the epilogues 5 start .. histogram
# zeroed, 7 sparse segments, 6 unpack + stores + histogram, 8 high bits + flush.
for k in "$@"; do
  RTX_LIB_PATH=$PWD/gpurun_scratch/lib_stamp$k.so timeout 600 python bench.py --config 2 --no-cpu-baseline --no-extras --steps 1 --warmup 1 2>gpurun_out/stamp_$k.err | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = b['roofline']; nq = b['config']['queries_per_gpu']
blocks = r['tile_pruning']['live_tiles_per_pair'] * nq / 2
print('stamp $k: ticks per live block', round(r['requested_bytes_per_query'] / 1024 * 64 * nq / blocks, 1), ' bounds pass per pair', round(r['bounds_pass']['requested_bytes_per_query'] / 1024 * 64 * 2, 1), ' hit_count ms', round(b['stage_ms_per_step']['hit_count'], 2))"
done
