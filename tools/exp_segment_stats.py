#!/usr/bin/env python3
"""Host-side study of what hit_count reads (no GPU): for a sample of the bench queries, the population of every
(k-mer row, 8192-reference tile) segment they ask for, by class, and what sharing rows between neighbouring queries
could save.  Usage: tools/exp_segment_stats.py [refs] [sample queries]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.oracle_py import Oracle  # noqa: E402
from raxtax_amd import synth  # noqa: E402


def main():
    n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
    n_s = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    db = synth.make_db(n_refs)
    o = Oracle()
    t0 = time.time()
    ot = o.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    off, post = ot.csr()
    print(f"tree + csr {time.time() - t0:.1f}s, postings {len(post)}")
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
    print("rows present", int((pop.sum(1) > 0).sum()))
    # queries in the order of their source reference (a stand-in for the min-hash order of the library)
    qs = synth.make_queries(db, 20000)
    orig = ot.original_index()
    inv = np.empty(n_refs, np.int64)
    inv[orig.astype(np.int64)] = np.arange(n_refs)
    order = np.argsort(inv[qs.source], kind="stable")
    start = len(order) // 3
    sel = order[start:start + n_s]
    edges = [0, 1, 17, 33, 65, 129, 257, 513, 1025, 2049, 4097, 8193]
    hist = np.zeros(len(edges) - 1, np.int64)
    bits = np.zeros(len(edges) - 1, np.int64)
    ksets = []
    for q in sel:
        km = o.sequence_to_kmers(qs.seq(int(q))).astype(np.int64)
        ksets.append(km)
        p = pop[km].reshape(-1)
        h, _ = np.histogram(p, bins=edges)
        hist += h
        bits += np.histogram(p, bins=edges, weights=p)[0].astype(np.int64)
    tot = hist.sum()
    print(f"\nsegments asked for per query: {tot / n_s:.0f} ({nt} tiles); by population of the segment:")
    for i in range(len(hist)):
        print(f"  [{edges[i]:5d}, {edges[i + 1] - 1:5d}]  {100 * hist[i] / tot:6.2f} % of the requests   {bits[i] / n_s / 1e6:8.3f} M references/query")
    dense_now = hist[2:].sum() * 1024 / n_s
    print(f"bytes per query now (dense > 16): {dense_now / 1e6:.2f} MB; all dense: {tot * 1024 / n_s / 1e6:.2f} MB")
    for cut_i, cut in ((3, 32), (4, 64), (5, 128), (6, 256)):
        dense = hist[cut_i:].sum() * 1024 / n_s
        lists = bits[2:cut_i].sum() * 2 / n_s
        print(f"  lists up to {cut:4d} references: dense {dense / 1e6:6.2f} MB + lists {lists / 1e6:5.2f} MB = {(dense + lists) / 1e6:6.2f} MB "
              f"({100 * (dense + lists) / dense_now:5.1f} % of now), {bits[2:cut_i].sum() / n_s / nt:7.0f} list entries per (query, tile)")
    # sharing between g consecutive queries: dense segments in the union / sum over the members
    dmask = pop > 16
    for g in (2, 4, 8):
        u = s = 0
        for i in range(0, n_s - g + 1, g):
            rows = np.unique(np.concatenate(ksets[i:i + g]))
            u += int(dmask[rows].sum())
            s += sum(int(dmask[k].sum()) for k in ksets[i:i + g])
        print(f"groups of {g} neighbours: union / sum of dense segments = {u / s:.3f}")


if __name__ == "__main__":
    main()
