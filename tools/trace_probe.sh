#!/bin/bash
# Kernel trace + stats of any probe script: tools/trace_probe.sh <tag> <script.py> [args...] -> gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"; rm -rf "$OUT/${TAG}_trace"
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 "$SCRIPT" "$@" > "$OUT/${TAG}_trace.log" 2>&1
echo "trace $TAG rc=$?"
for f in $(find "$OUT/${TAG}_trace" -name '*kernel_stats.csv'); do cp "$f" "$OUT/${TAG}_kernel_stats.csv"; done
rm -rf "$OUT/${TAG}_trace"
cut -d, -f1-4 "$OUT/${TAG}_kernel_stats.csv" | cut -c1-150 | head -${HEAD:-16}
