#!/bin/bash
# One gpurun call of a round: the GPU test-suite (recording the excused ties), the CPU scaling probe, profiles of
# the bench at BASELINE configs[2] with the counts packed and as u16.
set -u
TAG=${1:-r2b}
mkdir -p gpurun_out
rm -f gpurun_out/excuses_$TAG.jsonl
RTX_RECORD_EXCUSES=$PWD/gpurun_out/excuses_$TAG.jsonl timeout 2400 python -m pytest tests -m gpu -x -q -s --durations=15 > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"
grep -E "passed|failed" gpurun_out/${TAG}_tests.log | tail -3
timeout 900 python tools/cpu_scaling.py 500000 5 > gpurun_out/${TAG}_cpu_scaling.log 2>&1
echo "cpu scaling rc=$?"; cat gpurun_out/${TAG}_cpu_scaling.log
QPL=10240 bash tools/profile_bench.sh ${TAG}_u16 --u16-counts
QPL=10240 bash tools/profile_bench.sh ${TAG}
