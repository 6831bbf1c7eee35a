#!/usr/bin/env python3
"""How many 512-reference blocks of a query's count row hold a reference with a non-zero probability (table[count] > 0)?
taxon_prefix streams all N counts per query; a per-block maximum written by hit_count would let it skip dead blocks.
Usage: tools/exp_live_blocks.py [refs] [queries]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.oracle_py import Oracle  # noqa: E402
from raxtax_amd import synth  # noqa: E402

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 200
EPS = (1e-14, 1e-18, 1e-22, 1e-30, 1e-60, 1e-100)
db = synth.make_db(n_refs)
o = Oracle()
ot = o.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
qs = synth.make_queries(db, 1_000_000 if n_refs >= 500_000 else 100_000)
sel = np.arange(0, qs.n, qs.n // nq)[:nq]
bases = np.concatenate([qs.seq(int(q)) for q in sel])
off = np.concatenate([[0], np.cumsum([len(qs.seq(int(q))) for q in sel])]).astype(np.uint64)
t, counts = ot.hit_counts_batch(bases, off, threads=8)
tables, z, rc = o.prob_tables_batch(t, counts, threads=8)
for blk in (128, 512, 2048, 8192):
    nb = (n_refs + blk - 1) // blk
    pad = nb * blk - n_refs
    live_frac, live1e = [], []
    for i in range(len(sel)):
        tab = tables[i]
        nz = np.nonzero(tab > 0)[0]
        m_lo = int(nz.min()) if len(nz) else 0
        c = np.pad(counts[i], (0, pad)).reshape(nb, blk).max(axis=1)
        live_frac.append(float((c >= m_lo).mean()) if m_lo > 0 else 1.0)
        # refs whose probability can change a printed confidence: > 1e-12 of the total
        row = []
        for eps in EPS:
            nz2 = np.nonzero(tab >= eps)[0]
            m2 = int(nz2.min()) if len(nz2) else 0
            row.append(float((c >= m2).mean()) if m2 > 0 else 1.0)
        live1e.append(row)
    lf = np.array(live_frac)
    print(f"block {blk:5d}: live fraction mean {lf.mean():.4f} median {np.median(lf):.4f} p90 {np.quantile(lf, .9):.4f} max {lf.max():.4f};"
          f" with per-reference threshold {EPS}: mean {np.mean(live1e, axis=0).round(5)} max {np.max(live1e, axis=0).round(4)}")
mlos = [int(np.nonzero(tables[i] > 0)[0].min()) for i in range(len(sel))]
print("m_lo median", np.median(mlos), "t median", np.median(t), "queries with m_lo == 0:", int(np.sum(np.array(mlos) == 0)))
