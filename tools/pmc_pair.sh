#!/bin/bash
# memory-side counters of hit_count: pair kernel vs one query per wave (gpurun)
for v in "" "--no-pair"; do
bash tools/pmc_bench.sh pf1 "FETCH_SIZE" $v 2>&1 | grep "hit_count\|rc="
bash tools/pmc_bench.sh pf2 "WRITE_SIZE" $v 2>&1 | grep "hit_count\|rc="
bash tools/pmc_bench.sh pf3 "TCC_HIT_sum TCC_REQ_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" $v 2>&1 | grep "hit_count\|rc="
done
