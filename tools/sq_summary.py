#!/usr/bin/env python3
"""Per-kernel sums of the counters of a rocprofv3 --pmc pass (counter_collection.csv): sq_summary.py <dir> [--json]
--json: also writes profiles/sq_counters.json (keyed by the hash of the device sources: bench.py builds its `bounds` table from it)."""
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

if "--json" in sys.argv:
    import hashlib
    import json
    root = Path(__file__).resolve().parent.parent
    h = hashlib.sha256()
    for f in sorted((root / "raxtax_amd" / "csrc").glob("rtx_*")):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    out = {"device_source_sha": h.hexdigest()[:16], "what": "SQ counters per kernel over one step of bench.py at configs[2] (tools/sq_profile.sh): shares of the wave cycles", "kernels": {}}
    for k, d in agg.items():
        if not k.startswith("rtx::"):
            continue
        w = max(d.get("SQ_WAVES", 0), 1)
        tot = max(d.get("SQ_WAVE_CYCLES", 0), 1)
        out["kernels"][k] = {"launches": len(calls[k]), "waves": w, "wait_pct": round(100 * d.get("SQ_WAIT_ANY", 0) / tot, 1),
                             "stall_pct": round(100 * d.get("SQ_WAIT_INST_ANY", 0) / tot, 1), "issue_pct": round(100 * d.get("SQ_ACTIVE_INST_ANY", 0) / tot, 1),
                             "valu_per_wave": round(d.get("SQ_INSTS_VALU", 0) / w), "vmem_rd_per_wave": round(d.get("SQ_INSTS_VMEM_RD", 0) / w),
                             "gui_mcycles": round(d.get("GRBM_GUI_ACTIVE", 0) / 1e6, 2),
                             "waves_in_flight_per_cu": round(d.get("SQ_WAVE_CYCLES", 0) * 4 / max(d.get("GRBM_GUI_ACTIVE", 0), 1) / 256.0, 2)}
    (root / "profiles" / "sq_counters.json").write_text(json.dumps(out, indent=1) + "\n")
    print("wrote profiles/sq_counters.json")
