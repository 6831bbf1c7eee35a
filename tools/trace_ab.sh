#!/bin/bash
# Kernel trace + stats of the headline run for an A/B of bench options: tools/trace_ab.sh <tag> [bench args...] -> gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"; rm -rf "$OUT/${TAG}_trace"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$OUT/${TAG}_trace.log" 2>&1
echo "trace $TAG rc=$?"
for f in $(find "$OUT/${TAG}_trace" -name '*kernel_stats.csv'); do cp "$f" "$OUT/${TAG}_kernel_stats.csv"; done
rm -rf "$OUT/${TAG}_trace"
cut -d, -f1-4 "$OUT/${TAG}_kernel_stats.csv" | cut -c1-150 | head -14
