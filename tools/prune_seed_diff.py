#!/usr/bin/env python3
"""Rows of the queries on which the pruned and the full count of tests/test_gpu_parity.py::test_tile_pruning_randomised differ for a
seed (exact ties are the expected cause).  Usage: tools/prune_seed_diff.py <seed>"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
n_refs = int(rng.integers(8 * 8192 + 1, 14 * 8192))
L = int(rng.choice([320, 658, 900]))
db = synth.make_db(n_refs, length=L, seed_root=100 + seed, seed_db=200 + seed)
parts = [synth.make_queries(db, 150, seed=seed, mu_q=0.02, exact_frac=0.1),
         synth.make_queries(db, 80, seed=seed + 1, mu_q=float(rng.choice([0.06, 0.1, 0.15])), exact_frac=0.0),
         synth.make_queries(db, 40, seed=seed + 2, mu_q=0.25, exact_frac=0.0)]
seqs = [p.seq(q) for p in parts for q in range(p.n)]
order = rng.permutation(len(seqs))
seqs = [seqs[i] for i in order]
off = np.zeros(len(seqs) + 1, np.uint64)
off[1:] = np.cumsum([len(s) for s in seqs])
bases = np.concatenate(seqs)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
a, b = rx.Index(tree, tile_prune=False), rx.Index(tree)
ex = a.exact_matches(bases, off)
for skip in (False, True):
    ra = a.classify(bases, off, *ex, skip_exact_matches=skip)
    rb = b.classify(bases, off, *ex, skip_exact_matches=skip)
    for q in range(len(seqs)):
        xa, xb = ra.rows(q), rb.rows(q)
        la, lb = [r.lineage for r in xa], [r.lineage for r in xb]
        if la != lb:
            print(f"skip={skip} query {q} (t = {ra.t[q]}): {len(xa)} / {len(xb)} rows")
            for r1, r2 in zip(xa, xb):
                mark = "  " if r1.lineage == r2.lineage else "<>"
                print(f"  {mark} full {r1.lineage:7d} {[round(x, 17) for x in r1.confidence_values]}   pruned {r2.lineage:7d} {[round(x, 17) for x in r2.confidence_values]}")
            c = b.debug_hit_counts(q)
            top = np.argsort(-c.astype(np.int64))[:8]
            print("   best hits", [(int(i), int(c[i])) for i in top])
