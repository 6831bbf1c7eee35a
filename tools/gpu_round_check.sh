#!/bin/bash
# One gpurun call of a round: the GPU test-suite, the bench line as the driver takes it, profiles of the bench at
# BASELINE configs[2] (kernel trace + PMC passes).  Usage: tools/gpu_round_check.sh <tag>
set -u
TAG=${1:-r3}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -s --durations=12 > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"
grep -E "passed|failed" gpurun_out/${TAG}_tests.log | tail -3
bash tools/profile_bench.sh ${TAG}
python tools/make_traffic.py --tag ${TAG} --refs 500000 --queries 1000000 --fetch gpurun_out/${TAG}_fetch --write gpurun_out/${TAG}_write --tcc gpurun_out/${TAG}_tcc --note "default options: locator order, two queries per wave, packed counts, tile pruning"
cp profiles/traffic.json profiles/${TAG}_pmc_summary.csv gpurun_out/
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"; cat gpurun_out/${TAG}_bench.json
timeout 600 python bench.py --config 1 > gpurun_out/${TAG}_bench_config1.json 2>> gpurun_out/${TAG}_bench.err
echo "bench config 1 rc=$?"; cat gpurun_out/${TAG}_bench_config1.json
