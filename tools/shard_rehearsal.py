#!/usr/bin/env python3
"""BASELINE configs[4] rehearsed on ONE GPU: which design serves a database that still fits one GPU better on a node of S GPUs --
(R) the index replicated on every GPU and the QUERIES sharded (configs[3] at this database size), or (S) the REFERENCES sharded, every
GPU classifying every query against its range (mode B of SURVEY.md 8e, with tile pruning)?
    python tools/shard_rehearsal.py [refs] [queries] [shards]
Measures on the one GPU of the box: the unsharded pruned handle (what every GPU does in (R): node rate = S x this rate), and the S
emulated shards one after the other (exchange in-process: LocalComm): the S shards of a real node work side by side, so the node rate of
(S) is about queries / (total device time / S) -- an upper bound, it leaves the RCCL exchanges out.  Stage times of one shard are printed."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import sharded, synth  # noqa: E402


def main():
    n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    t0 = time.time()
    db = synth.make_db(n_refs)
    qs = synth.make_queries(db, n_q)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)          # with Tree.k_mer_map: the shards are cut out of it
    print(f"data + host tree with k-mer map: {time.time() - t0:.0f} s", flush=True)
    whole = rx.Index(rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False), stage_timing=True)
    whole.upload(qs.bases, qs.base_off)
    for _ in range(3):
        t0 = time.time()
        whole.run(0)
        whole.download(copy=False)
        dt_w = time.time() - t0
    st = whole.debug_prune_stats()
    print(f"(R) unsharded, pruned, {whole.device_bytes / 1e9:.1f} GB index: {n_q} queries in {dt_w * 1e3:.1f} ms = {n_q / dt_w:.0f} q/s per GPU -> "
          f"{S} GPUs with sharded queries: {S * n_q / dt_w:.0f} q/s; live tiles per pair {st['live_tiles_per_pair']:.2f}; stages {whole.stage_times()}", flush=True)
    ref = whole.download()
    del whole
    cuts = sharded.shard_cuts(tree.num_tips, S)
    shards = [sharded.ShardIndex(tree, r, cuts, sub_batch=4096) for r in range(S)]      # 2 scratch sets x S shards share the one GPU here
    lib = rx._lib.load()
    for s in shards:
        rx._lib.check(lib.rtx_index_set_option(s._h, 6, 1))
    clf = sharded.ShardedClassifier(shards, sharded.LocalComm())
    ex = shards[0].exact_matches(qs.bases, qs.base_off)
    clf.upload(qs.bases, qs.base_off, *ex)
    for _ in range(2):
        t0 = time.time()
        view = clf.run(copy=False)
        dt_s = time.time() - t0
    got = clf.run()
    same = float(np.mean(got.row_lineage == ref.row_lineage)) if len(got.row_lineage) == len(ref.row_lineage) else -1.0
    print(f"(S) {S} reference shards of {shards[0].n_refs} references, pruned = {shards[0].prunes}: all shards one after the other {dt_s * 1e3:.1f} ms for {n_q} queries "
          f"-> per shard {dt_s / S * 1e3:.1f} ms -> node rate at most {n_q / (dt_s / S):.0f} q/s (exchanges not counted); "
          f"live tiles per pair and shard {[round(s.debug_prune_stats()['live_tiles_per_pair'], 2) for s in shards]}; stages of shard 0 {shards[0].stage_times()}; "
          f"rows equal to the unsharded run: {same:.4f}", flush=True)


if __name__ == "__main__":
    main()
