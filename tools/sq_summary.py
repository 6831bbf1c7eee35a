#!/usr/bin/env python3
"""Per-kernel sums of the counters of a rocprofv3 --pmc pass (counter_collection.csv): sq_summary.py <dir>"""
import csv
import sys
from collections import defaultdict
from pathlib import Path

agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in Path(sys.argv[1]).rglob("*counter_collection.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"]
print(f"{'kernel':58s} {'calls':>6s} {'waves':>10s} {'gui_Mcyc':>9s} {'wavecyc/wave':>12s} {'wait%':>6s} {'stall%':>6s} {'active%':>7s} {'valu/wave':>9s} {'vmem/wave':>9s} {'waves_in_flight/CU':>18s}")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    w = max(d.get("SQ_WAVES", 0), 1)
    wc = d.get("SQ_WAVE_CYCLES", 0) * 4   # quad-cycles -> cycles
    gui = d.get("GRBM_GUI_ACTIVE", 0)
    tot = max(d.get("SQ_WAVE_CYCLES", 0), 1)
    inflight = wc / max(gui, 1) / 256.0
    print(f"{k[:58]:58s} {len(calls[k]):6d} {w:10.0f} {gui / 1e6:9.2f} {wc / w:12.0f} {100 * d.get('SQ_WAIT_ANY', 0) / tot:6.1f} {100 * d.get('SQ_WAIT_INST_ANY', 0) / tot:6.1f} "
          f"{100 * d.get('SQ_ACTIVE_INST_ANY', 0) / tot:7.1f} {d.get('SQ_INSTS_VALU', 0) / w:9.0f} {d.get('SQ_INSTS_VMEM_RD', 0) / w:9.0f} {inflight:18.1f}")
