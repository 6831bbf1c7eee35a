#!/usr/bin/env python3
"""Host-side simulation of row sharing inside a workgroup of G neighbouring queries (no GPU): union rows per tile,
padding of the 8-row folds when every wave may carry unconsumed rows over at most `carry` rounds of RR union rows.
Usage: tools/exp_quad_sim.py [refs] [groups]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.oracle_py import Oracle  # noqa: E402
from raxtax_amd import synth  # noqa: E402

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
n_groups = int(sys.argv[2]) if len(sys.argv) > 2 else 24
db = synth.make_db(n_refs)
o = Oracle()
ot = o.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
off, post = ot.csr()
nt = (n_refs + 8191) // 8192
pop = np.zeros((65536, nt), np.uint32)
for k0 in range(0, 65536, 2048):
    a, b = int(off[k0]), int(off[min(k0 + 2048, 65536)])
    if a == b:
        continue
    lens = np.diff(off[k0:k0 + 2049].astype(np.int64))
    rows = np.repeat(np.arange(len(lens)), lens)
    key = rows * nt + (post[a:b] >> 13).astype(np.int64)
    pop[k0:k0 + len(lens)] += np.bincount(key, minlength=len(lens) * nt).reshape(len(lens), nt).astype(np.uint32)
dense = pop > 16
qs = synth.make_queries(db, 1_000_000 if n_refs >= 500_000 else 100_000)
# min-hash order is not available on the host: order by source reference, then degrade by shuffling inside windows of
# 64 queries (the library's order puts 56 % of the k-mers of neighbours in common, the source order 80 %)
orig = ot.original_index()
inv = np.empty(n_refs, np.int64)
inv[orig.astype(np.int64)] = np.arange(n_refs)
order = np.argsort(inv[qs.source], kind="stable")
rng = np.random.default_rng(1)
for G in (4, 8):
    for RR, carry in ((8, 1), (8, 0), (16, 0), (16, 1), (12, 1)):
        tot_rows = tot_union = tot_groups8 = tot_rounds = 0
        for gi in range(n_groups):
            s = (gi * 7919 * G) % (len(order) - G)
            s -= s % G
            ks = [o.sequence_to_kmers(qs.seq(int(q))).astype(np.int64) for q in order[s:s + G]]
            allk = np.unique(np.concatenate(ks))
            memb = np.stack([np.isin(allk, k) for k in ks])           # [G][union k-mers]
            for tile in rng.integers(0, nt, 6):
                d = dense[allk, tile]
                m = memb[:, d]                                          # union rows of this tile (dense segments only)
                U = m.shape[1]
                tot_union += U
                tot_rows += int(m.sum())
                nround = (U + RR - 1) // RR
                tot_rounds += nround
                for w in range(G):
                    queue = []        # rounds of the queued rows
                    g8 = 0
                    for r in range(nround):
                        new = int(m[w, r * RR:(r + 1) * RR].sum())
                        queue += [r] * new
                        while len(queue) >= 8:
                            queue = queue[8:]
                            g8 += 1
                        if queue and queue[0] <= r - carry:            # would be overwritten: padded fold
                            queue = []
                            g8 += 1
                    if queue:
                        g8 += 1
                    tot_groups8 += g8
        print(f"G={G} RR={RR:2d} carry={carry}: union/sum = {tot_union / tot_rows:.3f}, folded slots / rows = {8 * tot_groups8 / tot_rows:.3f}, "
              f"rows per wave and tile {tot_rows / (n_groups * 6 * G):.0f}, union per tile {tot_union / (n_groups * 6):.0f}, rounds {tot_rounds / (n_groups * 6):.0f}")
