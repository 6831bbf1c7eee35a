#!/bin/bash
# Profiles bench.py under rocprofv3 on the GPU box: kernel trace + stats, then (separate passes,
# as the MI355X guide prescribes) the TCC fabric counters behind FETCH_SIZE / WRITE_SIZE.
# Usage: tools/profile_bench.sh <tag> [bench args...]   -> gpurun_out/<tag>_{trace,fetch,write}/
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/${TAG}_trace.log" 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_fetch" -- python3 "$ROOT/bench.py" $ARGS --queries 16384 > "$OUT/${TAG}_fetch.log" 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_write" -- python3 "$ROOT/bench.py" $ARGS --queries 16384 > "$OUT/${TAG}_write.log" 2>&1
echo "write rc=$?"
find "$OUT/${TAG}_trace" "$OUT/${TAG}_fetch" "$OUT/${TAG}_write" -type f | head -40
for f in $(find "$OUT/${TAG}_trace" -name '*kernel_stats.csv'); do echo "== $f"; head -12 "$f"; done
tail -2 "$OUT/${TAG}_trace.log"
