#!/bin/bash
# Every kernel of one chunk period of rtx_raxtax in steady state (from the exact_match_kernel of chunk 4 to that of chunk 6 of the last call), with its queue:
# start offset, duration, queue, name.   tools/e2e_kernels.sh [chunk]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"; rm -rf "$OUT/e2e_trace"
cd /tmp && export TMPDIR=/tmp PYTHONPATH="$ROOT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/e2e_trace" -- python3 "$ROOT/tools/e2e_case.py" 1048576 ${1:-131072} 1 > "$OUT/e2e_trace.log" 2>&1
echo "trace rc=$?"; grep "rep 3" "$OUT/e2e_trace.log" | cut -c1-160
python3 - "$OUT/e2e_trace" <<'PY'
import csv, sys
from pathlib import Path
rows = []
for f in Path(sys.argv[1]).rglob("*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("rtx::", "").replace("void ", "")[:44], r.get("Queue_Id", "?")))
rows.sort()
starts = [s for s, e, n, q in rows if "exact_match_kernel" in n][-9:]
t0, t1 = starts[4], starts[6]
print(f"two chunk periods: {(t1 - t0) / 1e6:.2f} ms")
for s, e, n, q in rows:
    if t0 <= s < t1 and e - s > 20000:
        print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f} q{q} {n}")
PY
rm -rf "$OUT/e2e_trace"
