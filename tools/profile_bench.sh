#!/bin/bash
# Profiles bench.py under rocprofv3 on the GPU box: kernel trace + stats of the bench command itself, then (separate
# passes, as the MI355X guide prescribes: no trace flags beside --pmc) the fabric counters behind FETCH_SIZE /
# WRITE_SIZE and the L2 hit/miss counters, over ONE WHOLE STEP of the same configuration (every sub-batch of the 1 M queries: the
# traffic of a launch depends on which slice of the lineage-ordered database its queries span).
# Usage: tools/profile_bench.sh <tag> [bench args...]   -> gpurun_out/<tag>_{trace,fetch,write,tcc}/
#        then tools/make_traffic.py turns the passes into profiles/<tag>_pmc_summary.csv and profiles/traffic.json
set -u
TAG=${1:-r2}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
rm -rf "$OUT/${TAG}_trace" "$OUT/${TAG}_fetch" "$OUT/${TAG}_write" "$OUT/${TAG}_tcc"   # (a fresh box has none; a local rerun would mix runs)
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/${TAG}_trace.log" 2>&1
echo "trace rc=$?"
PARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-extras $*"   # the configuration the line names: every sub-batch of the 1 M queries
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_fetch" -- python3 "$ROOT/bench.py" $PARGS > "$OUT/${TAG}_fetch.log" 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_write" -- python3 "$ROOT/bench.py" $PARGS > "$OUT/${TAG}_write.log" 2>&1
echo "write rc=$?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/${TAG}_tcc" -- python3 "$ROOT/bench.py" $PARGS > "$OUT/${TAG}_tcc.log" 2>&1
echo "tcc rc=$?"
for f in $(find "$OUT/${TAG}_trace" -name '*kernel_stats.csv'); do echo "== $f"; head -12 "$f"; cp "$f" "$OUT/${TAG}_kernel_stats.csv"; done
tail -2 "$OUT/${TAG}_trace.log"
