#!/bin/bash
# kernel trace + stats of two steps of bench.py at BASELINE configs[2]: trace_bench.sh <tag> -> gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=${1:-tr}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/${TAG}_trace.log" 2>&1
echo "trace rc=$?"
for f in $(find "$OUT/${TAG}_trace" -name '*kernel_stats.csv'); do cp "$f" "$OUT/${TAG}_kernel_stats.csv"; done
python3 - "$OUT/${TAG}_kernel_stats.csv" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'].split('(')[0][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
P
