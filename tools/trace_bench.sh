#!/bin/bash
# rocprofv3 kernel trace + stats of a short bench run. Usage: tools/trace_bench.sh <tag> [bench args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$OUT/${TAG}.log" 2>&1
echo "rc=$?"
for f in $(find "$OUT/${TAG}" -name '*kernel_stats.csv'); do cut -d, -f1-4,6,7 "$f" | head -14; done
tail -1 "$OUT/${TAG}.log" | cut -c1-300
