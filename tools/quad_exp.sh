#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out
for v in "$@"; do
  RTX_LIB_PATH=$ROOT/gpurun_scratch/lib_$v.so timeout 300 python3 $ROOT/tools/quad_time.py quad 20480 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /"
done
RTX_LIB_PATH=$ROOT/gpurun_scratch/lib_base.so timeout 300 python3 $ROOT/tools/quad_time.py single 20480 2>&1 | grep -v amdgpu.ids | sed "s/^/single: /"
cd /tmp && export TMPDIR=/tmp
export RTX_LIB_PATH=$ROOT/gpurun_scratch/lib_base.so
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $ROOT/gpurun_out/quad_sq -- python3 $ROOT/tools/quad_time.py quad 20480 > $ROOT/gpurun_out/quad_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $ROOT/gpurun_out/quad_tcc -- python3 $ROOT/tools/quad_time.py quad 20480 > $ROOT/gpurun_out/quad_tcc.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
root=os.environ.get("GRAFT_REPO_ROOT", ".")
for d in ("quad_sq","quad_tcc"):
    for f in glob.glob(f"{root}/gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,dd in agg.items():
            if "hit_count" in k:
                print(k, {c: round(sum(v)/len(v)) for c,v in dd.items()}, "n=", len(next(iter(dd.values()))))
PY
