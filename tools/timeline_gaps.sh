#!/bin/bash
# GPU idle time inside the steps of bench.py: kernel trace with timestamps, union of the busy intervals of all streams, the gaps between them.
#   tools/timeline_gaps.sh [bench args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"; rm -rf "$OUT/tl_trace"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/tl_trace" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$OUT/tl_trace.log" 2>&1
echo "trace rc=$?"
python3 - "$OUT/tl_trace" <<'PY'
import csv, sys
from pathlib import Path
rows = []
hdr = None
for f in Path(sys.argv[1]).rglob("*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        hdr = hdr or list(r.keys())
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50] + " q" + str(r.get("Queue_Id", "?")) + " s" + str(r.get("Stream_Id", "?"))))
print("columns:", hdr)
rows.sort()
# the timed steps: from the first exact_match_kernel of the second-last step on (one per step)
starts = [s for s, e, n in rows if "exact_match_kernel" in n]
t0, t1 = starts[-2], max(e for s, e, n in rows if "records_tail_kernel" in n)   # (behind the last step: the taps of the parity sample)
sel = [(s, e, n) for s, e, n in rows if s >= t0 and e <= t1]
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]; gaps = []
last_name = sel[0][2]
for s, e, n in sel[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, last_name, n)); busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e: last_name = n
busy += cur_e - cur_s
span = t1 - t0
print(f"last two steps: span {span/1e6:.2f} ms, GPU busy (union over streams) {busy/1e6:.2f} ms, idle {100*(span-busy)/span:.1f} %; {len(gaps)} gaps, {sum(g for g,_,_ in gaps)/1e6:.2f} ms")
big = sorted(gaps, reverse=True)[:12]
for g, a, b in big: print(f"  {g/1e3:8.1f} us  after {a}  before {b}")
# the events around the largest gap of the last step
g0 = max((g for g in gaps if True), key=lambda x: x[0])
# find its position
t_gap = None
cur_e2 = sel[0][1]
for s_, e_, n_ in sel[1:]:
    if s_ > cur_e2 and s_ - cur_e2 == g0[0]: t_gap = cur_e2; break
    cur_e2 = max(cur_e2, e_)
if t_gap:
    print("around the largest gap (us relative to its start):")
    for s_, e_, n_ in sel:
        if t_gap - 3_000_000 < s_ < t_gap + 3_000_000: print(f"   {(s_-t_gap)/1e3:9.1f} .. {(e_-t_gap)/1e3:9.1f}  {n_}")
from collections import Counter
c = Counter()
for g, a, b in gaps: c[(a, b)] += g
print("by pair of kernels (ms):")
for (a, b), g in c.most_common(12): print(f"  {g/1e6:6.2f}  {a} -> {b}")
PY
rm -rf "$OUT/tl_trace"
