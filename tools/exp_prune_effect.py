#!/usr/bin/env python3
"""Host-side study (no GPU): what changes in the reference algorithm's probabilities (the CPU oracle, prob.rs in f64) when every
reference of the tiles of 8192 whose largest hit count stays below a threshold is dropped -- the numerical headroom of the tile pruning
of DESIGN.md section 8.  Usage: tools/exp_prune_effect.py"""
import sys, time
import numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from raxtax_amd import synth
from oracle.oracle_py import Oracle
N = 500_000
db = synth.make_db(N); qs = synth.make_queries(db, 4000)
o = Oracle(native=True)
t0 = time.time()
ot = o.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
print("oracle tree", time.time() - t0, flush=True)
orig = ot.original_index().astype(np.int64)       # orig[pos] = input index?  counts are in tree order
rng = np.random.default_rng(3)
sel = rng.choice(qs.n, 24, replace=False)
ntile = (N + 8191) // 8192
for thr in (170, 200, 250, 300, 350, 400):
    worst = 0.0; worst_leaf = 0.0; live = []
    for qi in sel:
        t, counts = ot.hit_counts(qs.seq(int(qi)), skip_exact=False)
        counts = np.asarray(counts)
        M = int(counts.max())
        if M >= t: continue            # exact copies: the other branch
        p = o.highest_hit_prob_per_reference(t, t // 2, counts)
        tm = np.array([counts[a:a + 8192].max() for a in range(0, N, 8192)])
        dead = tm < thr
        c2 = counts.copy()
        for T in np.nonzero(dead)[0]:
            c2[T * 8192:(T + 1) * 8192] = 0
        p2 = o.highest_hit_prob_per_reference(t, t // 2, c2)
        # the dropped references get probability 0 in the pruned run (they are never summed)
        for T in np.nonzero(dead)[0]:
            p2[T * 8192:(T + 1) * 8192] = 0.0
        pre = np.concatenate([[0.0], np.cumsum(p)]); pre2 = np.concatenate([[0.0], np.cumsum(p2)])
        worst = max(worst, float(np.abs(pre - pre2).max()))
        worst_leaf = max(worst_leaf, float(np.abs(p - p2).max()))
        live.append(int((~dead).sum()))
    print(f"threshold {thr}: live tiles {np.mean(live):.1f} of {ntile}; largest change of any prefix sum {worst:.3e}, of any reference's probability {worst_leaf:.3e}", flush=True)
