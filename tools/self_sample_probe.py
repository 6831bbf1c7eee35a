#!/usr/bin/env python3
"""What would a self-sample of the references say about tile pruning?  Every k-th reference classified as a query with its exact copies
skipped (RTX_SKIP_EXACT_MATCHES): live tiles per query on the real-composition database and on a synthetic one.
   python tools/self_sample_probe.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402


def probe(name, lineages, seq_bytes, seq_off, n_sample=2048):
    tree = rx.Tree.new_flat(lineages, seq_bytes, seq_off, kmer_map=False)
    index = rx.Index(tree, prune_self_sample=False)
    print(f"{name}: verdict of the handle (pruning on, live share of its self-sample):", rx.Index(tree).prune_verdict)
    n = len(seq_off) - 1
    pick = np.linspace(0, n - 1, n_sample).astype(np.int64)
    seqs = [seq_bytes[int(seq_off[i]):int(seq_off[i + 1])] for i in pick]
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    bases = np.concatenate(seqs)
    for skip in (True, False):
        res = index.classify(bases, off, *index.exact_matches(bases, off), skip_exact_matches=skip)
        st = index.debug_prune_stats()
        print(f"{name}: skip_exact={skip}: {n} references, {index.ntiles if hasattr(index, 'ntiles') else '?'} tiles; live tiles per query {st.get('live_tiles_per_query'):.2f}, "
              f"above threshold {st.get('tiles_above_threshold_per_query', float('nan')):.2f}, with threshold {st.get('queries_with_threshold')} of {len(seqs)}, "
              f"mean threshold {st.get('mean_threshold'):.1f}, mean best hit {st.get('mean_best_hit_lower_bound'):.1f}", flush=True)


h = synth.real_composition_holdout(ROOT / "tests" / "golden" / "diptera_queries.fasta")
probe("real composition", h.lineages, h.seq_bytes, h.seq_off)
db = synth.make_db(120_000)
probe("synthetic 120k", db.lineages, db.seq_bytes, db.seq_off)
