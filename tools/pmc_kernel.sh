#!/bin/bash
# Counters of one pass of bench.py (headline, one step), per kernel: tools/pmc_kernel.sh <tag> "<counters>" [bench args...]
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"; rm -rf "$OUT/${TAG}_pmc"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/${TAG}_pmc" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > "$OUT/${TAG}_pmc.log" 2>&1
echo "pmc $TAG rc=$?"
python3 - "$OUT/${TAG}_pmc" <<'PY'
import csv, sys
from collections import defaultdict
from pathlib import Path
agg = defaultdict(lambda: defaultdict(float))
for f in Path(sys.argv[1]).rglob("*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
    print(k[:60], {c: f"{v:.4g}" for c, v in d.items()})
PY
rm -rf "$OUT/${TAG}_pmc"
