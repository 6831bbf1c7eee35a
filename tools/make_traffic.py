#!/usr/bin/env python3
"""Turns rocprofv3 PMC passes of bench.py into the committed evidence under profiles/:

  tools/make_traffic.py --tag r2 --refs 500000 --queries-per-launch 10240 \
        --fetch gpurun_out/r2_fetch --write gpurun_out/r2_write [--tcc gpurun_out/r2_tcc] [--note "..."]

  * profiles/<tag>_pmc_summary.csv   per kernel and counter: dispatches, mean and max per dispatch
  * profiles/traffic.json            entry "refs=<refs>,query_len=658": FETCH_SIZE / WRITE_SIZE (KB) of one hit_count
                                     launch + the fingerprint of the device sources they were measured on; bench.py
                                     turns it into roofline.traffic only while the fingerprint still matches.
FETCH_SIZE and WRITE_SIZE come from passes of their own (they do not fit one pass: MI355X_MICROARCH.md, PMC slots)."""
import argparse
import collections
import csv
import glob
import hashlib
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def device_source_sha() -> str:   # the same fingerprint as bench.py
    h = hashlib.sha256()
    for f in sorted((ROOT / "raxtax_amd" / "csrc").glob("rtx_*")):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def short(name: str) -> str:
    name = name.split("(")[0].strip()
    for pre in ("void ",):
        if name.startswith(pre):
            name = name[len(pre):]
    return name


GRID = collections.defaultdict(lambda: collections.defaultdict(list))   # grid size of every value in read_pass's lists


def read_pass(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(str(Path(d) / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            GRID[(id(agg), short(r["Kernel_Name"]))][r["Counter_Name"]].append(int(r.get("Grid_Size", 0) or 0))
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--refs", type=int, required=True)
    ap.add_argument("--query-len", type=int, default=658)
    ap.add_argument("--queries-per-launch", type=int, required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--tcc", default=None, help="optional pass with TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum ...")
    ap.add_argument("--full-launches-only", action="store_true", default=True)
    ap.add_argument("--note", default="")
    ap.add_argument("--with-bounds-pass", action="store_true", help="tile pruning on: a sub-batch is two launches of the hit_count kernel "
                    "(bounds on the union bitmap + the live tiles); the entry holds their sum, launches_per_sub_batch = 2")
    a = ap.parse_args()
    passes = [read_pass(a.fetch), read_pass(a.write)] + ([read_pass(a.tcc)] if a.tcc else [])
    rows = []
    for agg in passes:
        for k, d in sorted(agg.items()):
            if "rocclr" in k or "rocprim" in k:
                continue
            for c, v in sorted(d.items()):
                rows.append((k, c, len(v), sum(v) / len(v), max(v)))
    out = ROOT / "profiles" / f"{a.tag}_pmc_summary.csv"
    with open(out, "w") as f:
        f.write("kernel,counter,dispatches,mean_per_dispatch,max_per_dispatch\n")
        for k, c, n, mean, mx in rows:
            f.write(f"\"{k}\",{c},{n},{mean:.1f},{mx:.1f}\n")
    print("wrote", out)

    def hit(agg, ctr):
        for k, d in agg.items():
            if k.startswith("rtx::hit_count") and ctr in d:   # hit_count_kernel / hit_count_pair_kernel: whichever the run used
                v = d[ctr]
                g = GRID[(id(agg), k)][ctr]
                # launches of a full sub-batch (the last one of a step may be short): those with the largest grid.  Their
                # traffic differs from launch to launch -- the processing order gives every sub-batch another part of the
                # database -- so the mean over all full launches is what a step sees
                full = [x for x, gs in zip(v, g) if gs == max(g)]
                mean = sum(full) / len(full)
                if a.with_bounds_pass:   # tile pruning: the same kernel runs once more per sub-batch, on the union bitmap (a smaller grid)
                    g2 = max(gs for gs in g if gs < max(g) and g.count(gs) * 2 >= len(full))
                    coarse = [x for x, gs in zip(v, g) if gs == g2]
                    mean += sum(coarse) / len(coarse)
                return mean, len(full)
        raise SystemExit(f"no hit_count dispatches with {ctr}")

    fetch_kb, nf = hit(passes[0], "FETCH_SIZE")
    write_kb, nw = hit(passes[1], "WRITE_SIZE")
    tf = ROOT / "profiles" / "traffic.json"
    t = json.loads(tf.read_text()) if tf.exists() else {}
    if "configs" not in t:
        t = {"configs": {}, "note": "gfx950: FETCH_SIZE reports half of 16-byte-per-lane reads (MI355X_MICROARCH.md, HBM): bench.py doubles it; "
                                    "WRITE_SIZE reads exactly for 16-byte-per-lane stores"}
    t["configs"][f"refs={a.refs},query_len={a.query_len}"] = {
        "queries_per_launch": a.queries_per_launch, "hit_count_fetch_kb": fetch_kb, "hit_count_write_kb": write_kb,
        "launches_profiled": min(nf, nw), "launches_per_sub_batch": 2 if a.with_bounds_pass else 1, "device_source_sha": device_source_sha(),
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_bench.sh {a.tag}); mean KB per full "
                  f"sub-batch of {a.queries_per_launch} queries ({'bounds pass + live tiles' if a.with_bounds_pass else 'one hit_count launch'}). {a.note}".strip(),
    }
    tf.write_text(json.dumps(t, indent=1) + "\n")
    print("updated", tf, json.dumps(t["configs"][f"refs={a.refs},query_len={a.query_len}"]))


if __name__ == "__main__":
    main()
