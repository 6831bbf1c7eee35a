#!/bin/bash
# The tile-pruning tests and a kernel trace of the pruned bench (gpurun_out/prune_t.log, prune_b.log, prune_top.txt).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
R=$PWD
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "tile_pruning" > gpurun_out/prune_t.log 2>&1 || { tail -30 gpurun_out/prune_t.log; exit 1; }
tail -4 gpurun_out/prune_t.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_prune
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_prune -o p -- python3 $R/bench.py --tile-prune --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prune_b.log 2>&1 || { tail -30 $R/gpurun_out/prune_b.log; exit 1; }
cd $R
python3 - <<'PY' | tee gpurun_out/prune_top.txt
import sqlite3, json
c = sqlite3.connect('gpurun_out/prof_prune/p_results.db')
for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 9"):
    print(f"{r[0][:70]:70s} calls {r[1]:5d} avg {r[3]/1000:9.1f} us  {r[4]:5.1f} %")
for line in open('gpurun_out/prune_b.log'):
    if line.startswith('{'):
        j = json.loads(line)
        print("value", j["value"], "ms_per_step", j["ms_per_step"], j["stage_ms_per_step"])
PY
