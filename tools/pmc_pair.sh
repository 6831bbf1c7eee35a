#!/bin/bash
# SQ counters of hit_count: pair kernel vs one query per wave (gpurun)
bash tools/pmc_bench.sh pp1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" 2>&1 | grep "hit_count\|rc="
bash tools/pmc_bench.sh pp2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" 2>&1 | grep "hit_count\|rc="
bash tools/pmc_bench.sh ps1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" --no-pair 2>&1 | grep "hit_count\|rc="
bash tools/pmc_bench.sh ps2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --no-pair 2>&1 | grep "hit_count\|rc="
