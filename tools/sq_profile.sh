#!/bin/bash
# SQ counters per kernel over one step of bench.py at BASELINE configs[2] (a --pmc pass of its own, no trace flags):
#   tools/sq_profile.sh <tag>  -> gpurun_out/<tag>_sq/ and a per-kernel table on stdout (tools/sq_summary.py)
set -u
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${TAG}_sq" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extras > "$OUT/${TAG}_sq.log" 2>&1
echo "sq rc=$?"
python3 "$ROOT/tools/sq_summary.py" "$OUT/${TAG}_sq" --json | tee "$OUT/${TAG}_sq_summary.txt"
cp "$ROOT/profiles/sq_counters.json" "$OUT/${TAG}_sq_counters.json"
