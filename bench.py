#!/usr/bin/env python3
"""bench.py -- classified queries/s of the raxtax hot path on N MI355X of one node.

A "step" is one pass of the whole device path (kmer_extract -> hit_count -> prob_table ->
taxon_prefix -> lineage_walk -> result rows on the host) over one batch of synthetic queries
per GPU, with the queries and the index already resident in HBM when the timed region starts
(BASELINE.json configs[1]: 100k COI-length queries vs a 50k-sequence reference database,
replicated per GPU; queries are sharded, so scaling is weak: every rank classifies its own
--queries).  With N > 1 the per-rank result records are gathered to rank 0 over RCCL inside the
timed region (the only collective on the path).

One JSON line on rank 0: metric/value (whole-job queries/s), ms_per_step, `roofline` of the
dominant kernel (hit_count: algorithmic bytes 4*H_q + L_q per query, SURVEY.md 8d, over the
kernel time measured with HIP events on the library's stream) and `cpu_baseline` (the CPU oracle,
a C port of the reference algorithm, timed on this host's cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 measured for a copy


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--refs", type=int, default=50_000, help="reference sequences (replicated per GPU)")
    ap.add_argument("--queries", type=int, default=100_000, help="queries per GPU per step")
    ap.add_argument("--sub-batch", type=int, default=0, help="queries per kernel wave (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams per handle (0 = library default)")
    ap.add_argument("--no-cluster", action="store_true", help="process the queries in input order (RTX_OPT_CLUSTER = 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-exact-matches", action="store_true")
    ap.add_argument("--stage-times", action="store_true",
                    help="HIP events around every kernel (stage_ms_per_step for all stages; ~1 ms slower per step)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL over xGMI; gloo only to smoke-test the "
                         "multi-rank flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--shard-db", action="store_true",
                    help="BASELINE configs[4]: shard the REFERENCES over the GPUs (every rank classifies the same "
                         "queries; RCCL all-reduce of histograms + all-gather of prefix sums per sub-batch)")
    return ap.parse_args()


def cpu_baseline(db, qs, target_s: float):
    """Times the oracle (oracle/, kind "port") on this host's cores on a prefix of the queries."""
    from oracle.oracle_py import Oracle

    cores = os.cpu_count() or 1
    orc = Oracle(native=True)
    otree = orc.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    L = db.length

    def run(n):
        t0 = time.perf_counter()
        bad, _, _ = otree.classify_batch(qs.bases[: n * L], qs.base_off[: n + 1], threads=cores)
        return time.perf_counter() - t0, bad

    probe = min(qs.n, 4 * cores)
    dt, _ = run(probe)
    rate = probe / dt
    n = int(min(qs.n, max(probe, rate * target_s)))
    dt, bad = run(n)
    return {"value": n / dt, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"first {n} queries of the same workload, {cores} threads, {dt:.1f} s, "
                      f"chunking as src/main.rs:119-124, no string formatting"}


def measured_traffic(args, launches_per_step):
    """Fabric-side bytes per hit_count launch from the committed rocprofv3 PMC passes (profiles/traffic.json:
    FETCH_SIZE x 2 as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, + WRITE_SIZE), if the
    profile was taken on this configuration; else null."""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None
    try:
        t = json.loads(f.read_text())
        if t.get("refs") == args.refs and t.get("query_len") == 658:
            per_query = (2.0 * t["hit_count_fetch_kb"] + t["hit_count_write_kb"]) * 1024.0 / t["queries_per_launch"]
            return per_query * args.queries / launches_per_step      # bytes per launch of this run
    except Exception:
        return None
    return None


def bench_sharded_db(args, rx, synth, db, tree, dist, rank, local_rank, world, flags):
    """Reference-sharded database: rank r holds references [cuts[r], cuts[r+1]); all ranks classify the same
    --queries; strong scaling (total work fixed)."""
    import torch

    from raxtax_amd import sharded

    qs = synth.make_queries(db, args.queries, seed=3)              # identical on every rank
    cuts = sharded.shard_cuts(tree.num_tips, world)
    shard = sharded.ShardIndex(tree, rank, cuts, device=local_rank, sub_batch=args.sub_batch or 2048)
    if dist is not None:
        w = torch.tensor([shard.n_bnd_local], device="cuda", dtype=torch.int64)
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        comm = sharded.TorchComm(dist, world, [int(x.item()) for x in ws])
    else:
        comm = sharded.LocalComm()
    clf = sharded.ShardedClassifier([shard], comm)
    ex_ids, ex_off = shard.exact_matches(qs.bases, qs.base_off)

    def barrier():
        if pending[0] is not None:      # the gather of the last step belongs to the timed region
            dist_util.gather_finish(pending[0])
            pending[0] = None
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        clf.classify(qs.bases, qs.base_off, ex_ids, ex_off, skip_exact_matches=bool(flags))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = clf.classify(qs.bases, qs.base_off, ex_ids, ex_off, skip_exact_matches=bool(flags))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        print(json.dumps({
            "metric": "classified queries/sec (whole node)", "value": args.queries * args.steps / elapsed,
            "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32 bit-planes + f64", "data": "synthetic",
            "config": {"workload": f"{args.queries} synthetic 658-bp queries vs {args.refs}-seq DB sharded by reference id "
                                   f"over {world} GPU(s) (BASELINE.json configs[4] shape; includes H2D of the queries)",
                       "refs": args.refs, "queries": args.queries, "parallelism": f"references sharded x{world}",
                       "classified_ok": int((res.status == 0).sum())},
            "roofline": None, "cpu_baseline": None}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the device path)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()     # smoke test: several ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dist = None
    coll_device = "cuda" if args.backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import raxtax_amd as rx
    from raxtax_amd import dist_util, synth

    # ---- inputs (untimed): identical database on every rank, rank-specific queries
    db = synth.make_db(args.refs)
    qs = synth.make_queries(db, args.queries, seed=3 + rank, first_label=rank * args.queries)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    flags = rx.RTX_SKIP_EXACT_MATCHES if args.skip_exact_matches else 0
    if args.shard_db:
        return bench_sharded_db(args, rx, synth, db, tree, dist, rank, local_rank, world, flags)
    index = rx.Index(tree, device=local_rank, sub_batch=args.sub_batch, streams=args.streams, stage_timing=args.stage_times,
                     cluster=False if args.no_cluster else None)
    ex_ids, ex_off = index.exact_matches(qs.bases, qs.base_off)   # Tree.sequences.get, raxtax.rs:42 (host)
    index.upload(qs.bases, qs.base_off, ex_ids, ex_off)            # inputs resident in HBM from here on

    lib = rx._lib.load()
    # N > 1: packing and gathering the records of step i happen while the device classifies step i+1 (two sets of
    # buffers); those of the last step are completed inside the timed region
    rec_buf = [None, None]
    gather_cache = [{}, {}]
    pending = [None]
    step_no = [0]

    prev_view = [None]

    def ship(view):
        """Packs the result records of a finished step and starts their gather on rank 0 (the only collective: RCCL
        over xGMI); the gather started before is completed first (two sets of buffers alternate)."""
        k = step_no[0] & 1
        step_no[0] += 1
        need = lib.rtx_result_pack(ctypes.byref(view), None, 0)       # native pack: 24 B/query + 21 B/row
        if rec_buf[k] is None or rec_buf[k].shape[0] < need:
            rec_buf[k] = dist_util.pinned_bytes(int(need * 1.25) + 64)
        n = lib.rtx_result_pack(ctypes.byref(view), rec_buf[k].ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), rec_buf[k].shape[0])
        assert n == need, "rtx_result_pack failed"
        if pending[0] is not None:
            dist_util.gather_finish(pending[0])
        pending[0] = dist_util.gather_start(dist, rec_buf[k][:n], rank, world, device=coll_device, cache=gather_cache[k])

    def step():
        index.run(flags)                        # enqueues every kernel of this step
        if dist is not None and prev_view[0] is not None:
            ship(prev_view[0])                  # host work of the step before (its view stays valid until the second-next
            prev_view[0] = None                 # download) while the device classifies this one
        view = index.download(copy=False)       # streams the result records back + host finalisation
        prev_view[0] = view
        return view

    def barrier():
        if dist is not None and prev_view[0] is not None:   # records and gather of the last step belong to the timed region
            ship(prev_view[0])
            prev_view[0] = None
        if pending[0] is not None:
            dist_util.gather_finish(pending[0])
            pending[0] = None
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage_ms = {s: 0.0 for s in rx._lib.STAGES}
    stage_n = {s: 0 for s in rx._lib.STAGES}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for s, (ms, n) in index.stage_times().items():   # reads already-recorded HIP events
            stage_ms[s] += ms
            stage_n[s] += n
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=coll_device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    work = index.work()
    view = index.download(copy=False)
    ok = int((np.ctypeslib.as_array(view.status, shape=(args.queries,)) == 0).sum())
    if rank == 0:
        total_q = args.queries * world * args.steps
        # roofline of the dominant kernel: algorithmic bytes (4 B per posting the reference would
        # touch + the query bytes) per launch / mean launch duration (HIP events, library stream)
        bytes_alg = 4 * work["sum_hits"] + work["sum_query_bytes"]     # one step, this rank
        n_launch = max(stage_n["hit_count"], 1)
        hit_ms = stage_ms["hit_count"] / n_launch
        launches_per_step = n_launch / args.steps
        achieved = (bytes_alg / launches_per_step) / (hit_ms * 1e-3) / 1e9
        line = {
            "metric": "classified queries/sec (whole node)",
            "value": total_q / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 bit-planes + f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.queries} synthetic COI-length (658 bp) queries per GPU vs {args.refs}-seq "
                            f"reference DB replicated in HBM (BASELINE.json configs[1])",
                "refs": args.refs, "queries_per_gpu": args.queries, "query_len": db.length,
                "synthetic_data": "phylo (SURVEY.md 8d)", "parallelism": f"queries sharded x{world}, index replicated",
                "classified_ok": ok, "skip_exact_matches": bool(args.skip_exact_matches),
            },
            "roofline": {
                "bound": "hbm", "kernel": "hit_count_kernel",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(args, launches_per_step),
                "algorithmic_bytes_per_query": bytes_alg / args.queries,
                "bitmap_bytes_per_query": work["bitmap_bytes_read"] / args.queries,
                # what actually limits the row loop: bitmap-row bytes delivered by the vector L1 (64 B per clock and CU,
                # 256 CUs, 2.4 GHz), padding rows of the row lists not counted (DESIGN.md section 3, K2+K3)
                "l1_achieved": (work["bitmap_bytes_read"] / launches_per_step) / (hit_ms * 1e-3) / 1e9,
                "l1_peak": 256 * 64 * 2.4, "l1_frac": (work["bitmap_bytes_read"] / launches_per_step) / (hit_ms * 1e-3) / 1e9 / (256 * 64 * 2.4),
                "launch_ms": hit_ms, "launches_per_step": launches_per_step,
            },
            "stage_ms_per_step": {s: stage_ms[s] / args.steps for s in stage_ms},
        }
        if not args.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(db, qs, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
