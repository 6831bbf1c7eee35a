#!/bin/bash
# bench at configs[2] with other stream counts / sub-batch sizes (gpurun_out/exp_streams.txt)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; : > gpurun_out/exp_streams.txt
IFS="|" read -r -a VS <<< "${VARIANTS:-|--streams 2|--sub-batch 8192|--sub-batch 32768}"
for v in "${VS[@]}"; do
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $v 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$v'.ljust(32), round(j['value']), round(j['ms_per_step'], 1), {k: round(x, 1) for k, x in j['stage_ms_per_step'].items()})" >> gpurun_out/exp_streams.txt || exit 1
done
cat gpurun_out/exp_streams.txt
