#!/usr/bin/env python3
"""Upper bound for a better processing order at BASELINE configs[2] size: the queries sorted on the host by their true
source reference (in lineage order) and processed in input order, against the library's min-hash order and against
plain input order.  Usage: tools/exp_order_potential2.py [queries]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
db = synth.make_db(500_000)
qs = synth.make_queries(db, n_q)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
orig = tree.original_index()
inv = np.empty(len(orig), np.int64)
inv[orig.astype(np.int64)] = np.arange(len(orig))
L = db.length
B = qs.bases.reshape(-1, L)
by_source = np.ascontiguousarray(B[np.argsort(inv[qs.source], kind="stable")]).reshape(-1)
# RTX_EXP_PAIR_ANY_ORDER=1 in the environment keeps the pair kernel on for the two host-made orders as well
for name, bases, cluster, locator in (("input order", qs.bases, False, False), ("min-hash order", qs.bases, True, False),
                                      ("locator + min-hash (library)", qs.bases, True, True), ("true source order", by_source, False, False)):
    ix = rx.Index(tree, cluster=cluster, stage_timing=True, locator=locator)
    ex = ix.exact_matches(bases, qs.base_off)
    ix.upload(bases, qs.base_off, *ex)
    for _ in range(3):
        t0 = time.time()
        ix.run(0)
        ix.download(copy=False)
        dt = time.time() - t0
    st = ix.stage_times()
    print(f"{name:28s}: {dt * 1e3:8.1f} ms, hit_count {st['hit_count'][0]:8.1f} ms, prefix {st['taxon_prefix'][0]:6.1f} ms", flush=True)
    del ix
