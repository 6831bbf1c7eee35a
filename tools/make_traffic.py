#!/usr/bin/env python3
"""Turns rocprofv3 PMC passes of bench.py into the committed evidence under profiles/:

  tools/make_traffic.py --tag r3 --refs 500000 --queries 1000000 \
        --fetch gpurun_out/r3_fetch --write gpurun_out/r3_write [--tcc gpurun_out/r3_tcc] [--note "..."]

  * profiles/<tag>_pmc_summary.csv   per kernel and counter: dispatches, mean and max per dispatch
  * profiles/traffic.json            entry "refs=<refs>,query_len=658,queries=<per step>": FETCH_SIZE / WRITE_SIZE (KB) and L2 requests /
                                     hits of the hit_count launches of ONE WHOLE STEP, per kind of launch (the counting of the live
                                     tiles / the bounds pass of the tile pruning), + the fingerprint of the device sources they were
                                     measured on; bench.py turns it into roofline.traffic only while the fingerprint still matches.
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
    # gpurun MERGES a call's output into gpurun_out/: a directory that was profiled into before still holds the older runs (of older
    # builds).  Only the newest result file of the pass counts (until late in round 4 all of them were averaged: three runs of 16 launches).
    files = sorted(glob.glob(str(Path(d) / "**" / "*counter_collection.csv"), recursive=True), key=lambda f: Path(f).stat().st_mtime)
    if len(files) > 1:
        print(f"{d}: {len(files)} result files, taking the newest ({files[-1]})")
    for f in files[-1:]:
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0) or 0))   # dispatch order: the last step is the last dispatches
        for r in rows:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            GRID[(id(agg), short(r["Kernel_Name"]))][r["Counter_Name"]].append(int(r.get("Grid_Size", 0) or 0))
    return agg


def last_step_only(agg, per_step: int):
    """Round 6: the first run of a fresh handle may be repeated (the counts rows / record segments of the scratch start small and double when a
    run overflows them: rtx_api_download.hip), so the profiled command `bench.py --steps 1 --warmup 0` can hold the step two or three times.
    Only its LAST execution counts: runs = launches of the counting kernel / launches per step; every kernel whose launches are a multiple of
    the runs (+ the one launch of the handle's self-sample at creation) keeps its last share."""
    n_live = max((len(v) for k, d in agg.items() if k.startswith("rtx::hit_count") for v in d.values()), default=0)
    runs = n_live // per_step if per_step else 1
    if runs <= 1:
        return agg, 1
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, d in agg.items():
        for c, v in d.items():
            keep = len(v) // runs if len(v) >= runs else len(v)
            out[k][c] = v[-keep:] if k.startswith("rtx::") and len(v) >= runs else v
    return out, runs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--refs", type=int, required=True)
    ap.add_argument("--query-len", type=int, default=658)
    ap.add_argument("--queries", type=int, required=True, help="queries per step of the profiled bench command (the whole step is profiled)")
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--tcc", default=None, help="optional pass with TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum")
    ap.add_argument("--note", default="")
    ap.add_argument("--sub-batch", type=int, default=65536, help="queries per launch of the profiled run (launches per step = queries / sub-batch)")
    ap.add_argument("--unpruned", action="store_true", help="the run counted every tile (--no-tile-prune): one kind of launch")
    a = ap.parse_args()
    passes = [read_pass(a.fetch), read_pass(a.write)] + ([read_pass(a.tcc)] if a.tcc else [])
    per_step = (a.queries + a.sub_batch - 1) // a.sub_batch
    trimmed = [last_step_only(p_, per_step) for p_ in passes]
    if any(r > 1 for _, r in trimmed):
        print(f"the profiled command ran the step {[r for _, r in trimmed]} times (the handle's first run was repeated with larger scratch): the last execution counts")
    passes = [p_ for p_, _ in trimmed]
    rows = []
    for agg in passes:
        for k, d in sorted(agg.items()):
            if "rocclr" in k or "rocprim" in k:
                continue
            for c, v in sorted(d.items()):
                rows.append((k, c, len(v), sum(v) / len(v), max(v), sum(v)))
    out = ROOT / "profiles" / f"{a.tag}_pmc_summary.csv"
    with open(out, "w") as f:
        f.write("kernel,counter,dispatches,mean_per_dispatch,max_per_dispatch,sum\n")
        for k, c, n, mean, mx, tot in rows:
            f.write(f"\"{k}\",{c},{n},{mean:.1f},{mx:.1f},{tot:.1f}\n")
    print("wrote", out)

    def hit(agg, ctr):
        """Sum and number of the dispatches of the hit_count kernel per kind of launch.  The bounds pass of the tile pruning is an
        instantiation of its own (hit_count_pair_kernel<10, true, true, false>: the third argument is kBounds), so the kernel name tells the kinds apart."""
        out = {}
        for k, d in agg.items():
            if k.startswith("rtx::hit_count") and ctr in d:   # hit_count_kernel / hit_count_pair_kernel: whichever the run used
                # hit_count_pair_kernel<NP, kPacked, kBounds, kItems>: kBounds 1 (or `true`, round 3) = the bounds pass over blocks of 64,
                # 2 = the fine bounds pass over blocks of 8 (round 4), 0 = the tiles of the database
                targs = k.replace(" ", "").split("<", 1)[1].rstrip(">").split(",") if "<" in k else []
                kb = targs[2] if len(targs) >= 3 and "pair_kernel" in k else "0"
                kind = "all" if a.unpruned else ("bounds" if kb in ("1", "true") else "fine" if kb == "2" else "live")
                tot, n = out.get(kind, (0.0, 0))
                out[kind] = (tot + sum(d[ctr]), n + len(d[ctr]))
        return out or None

    fetch, write = hit(passes[0], "FETCH_SIZE"), hit(passes[1], "WRITE_SIZE")
    if fetch is None or write is None:
        raise SystemExit("no hit_count dispatches in the FETCH_SIZE / WRITE_SIZE passes")
    req = hit(passes[2], "TCC_REQ_sum") if a.tcc else None
    hitc = hit(passes[2], "TCC_HIT_sum") if a.tcc else None
    kinds = {}
    for kind in fetch:
        assert fetch[kind][1] == write[kind][1], "the passes saw different numbers of launches"
        kinds[kind] = {"launches": fetch[kind][1], "fetch_kb": fetch[kind][0], "write_kb": write[kind][0]}
        if req and hitc:
            kinds[kind]["tcc_req"] = req[kind][0]
            kinds[kind]["tcc_hit"] = hitc[kind][0]
    # the whole step, every kernel of the library (step_fabric_frac = (2 FETCH + WRITE) / ms_per_step / 8 TB/s)
    per_kernel = {}
    for k in sorted(set(passes[0]) | set(passes[1])):
        if not k.startswith("rtx::"):
            continue
        fk, wk = passes[0].get(k, {}).get("FETCH_SIZE", []), passes[1].get(k, {}).get("WRITE_SIZE", [])
        per_kernel[k] = {"launches": max(len(fk), len(wk)), "fetch_kb": sum(fk), "write_kb": sum(wk)}
    step = {"fetch_kb": sum(v["fetch_kb"] for v in per_kernel.values()), "write_kb": sum(v["write_kb"] for v in per_kernel.values()),
            "per_kernel": per_kernel,
            "note": "sums over one whole step of the profiled command, index-build kernels included only if the pass saw them (the passes profile "
                    "bench.py --steps 1 --warmup 0: the index build and the table build run once in front of the step -- bench.py leaves "
                    "out the kernels that do not belong to a step)"}
    tf = ROOT / "profiles" / "traffic.json"
    t = json.loads(tf.read_text()) if tf.exists() else {}
    if "configs" not in t:
        t = {"configs": {}}
    t["note"] = ("gfx950: FETCH_SIZE reports half of 16-byte-per-lane reads (MI355X_MICROARCH.md, HBM): bench.py doubles it; WRITE_SIZE reads "
                 "exactly for 16-byte-per-lane stores.  Per kind of launch of the hit_count kernel: sums over ALL launches of one step of the "
                 "named size (KB), and their number")
    key = f"refs={a.refs},query_len={a.query_len},queries={a.queries}"
    t["configs"] = {k: v for k, v in t["configs"].items() if "kinds" in v}      # entries of the older layout are of older builds anyway
    t["configs"][key] = {
        "pruned": not a.unpruned, "kinds": kinds, "step": step, "device_source_sha": device_source_sha(),
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_* (separate passes, tools/profile_bench.sh {a.tag}: one whole step of "
                  f"`bench.py` at this size, every sub-batch). {a.note}".strip(),
    }
    tf.write_text(json.dumps(t, indent=1) + "\n")
    print("updated", tf, json.dumps(t["configs"][key]))


if __name__ == "__main__":
    main()
