#!/bin/bash
# Re-takes the profiles traffic.json is keyed to (the device-source hash changes with every edit of raxtax_amd/csrc/rtx_*).
# Usage (gpurun): bash tools/refresh_profiles.sh <tag>
set -u
TAG=${1:-r3}
bash tools/profile_bench.sh ${TAG}
python tools/make_traffic.py --tag ${TAG} --refs 500000 --queries 1000000 --fetch gpurun_out/${TAG}_fetch --write gpurun_out/${TAG}_write --tcc gpurun_out/${TAG}_tcc --note "default options: locator order, two queries per wave, packed counts, tile pruning"
cp profiles/traffic.json profiles/${TAG}_pmc_summary.csv gpurun_out/
cp gpurun_out/${TAG}_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_copy.csv 2>/dev/null
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/${TAG}_bench.json
