#!/bin/bash
# Collects a set of rocprofv3 PMC counters over a short bench.py run (own pass: no trace flags).
# Usage: tools/pmc_bench.sh <tag> "<counters>" [bench args...]
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/${TAG}" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extras --queries 16384 "$@" > "$OUT/${TAG}.log" 2>&1
echo "rc=$?"
python3 - "$OUT/${TAG}" <<'PY'
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0][-32:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "rocclr" in k or "bitmap" in k: continue
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
