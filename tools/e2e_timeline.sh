#!/bin/bash
# Where the device is idle between the chunks of rtx_raxtax: kernel trace with timestamps of tools/e2e_case.py, union of the busy intervals of all streams per chunk
# (a chunk = from one exact_match_kernel to the next).   tools/e2e_timeline.sh [chunk]
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
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
starts = [s for s, e, n in rows if "exact_match_kernel" in n]
starts = starts[-9:]          # the last call: eight chunks (+ the end)
for k in range(len(starts) - 1):
    t0, t1 = starts[k], starts[k + 1]
    sel = [(s, e, n) for s, e, n in rows if s >= t0 and s < t1]
    busy = 0; cs, ce = sel[0][0], sel[0][1]; gaps = []; last = sel[0][2]
    for s, e, n in sel[1:]:
        if s > ce:
            gaps.append((s - ce, last, n)); busy += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
        if e >= ce: last = n
    busy += ce - cs
    big = sorted(gaps, reverse=True)[:3]
    print(f"chunk {k}: span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps; largest: " +
          "; ".join(f"{g / 1e3:.0f} us after {a[5:40]} before {b[5:40]}" for g, a, b in big))
PY
rm -rf "$OUT/e2e_trace"
