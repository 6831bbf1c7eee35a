#!/usr/bin/env python3
"""Tile pruning (RTX_OPT_TILE_PRUNE) against the full count at BASELINE configs[2] size: results and time.
Usage: tools/exp_prune.py [queries] [refs]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n_refs = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
db = synth.make_db(n_refs)
qs = synth.make_queries(db, n_q)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
out = {}
for name, prune in (("full", False), ("pruned", True)):
    ix = rx.Index(tree, tile_prune=prune, stage_timing=True)
    ex = ix.exact_matches(qs.bases, qs.base_off)
    ix.upload(qs.bases, qs.base_off, *ex)
    for flags in (0, rx.RTX_SKIP_EXACT_MATCHES):
        for _ in range(2):
            t0 = time.time()
            ix.run(flags)
            ix.download(copy=False)
            dt = time.time() - t0
        res = ix.download()
        st = ix.stage_times()
        w = ix.work()
        print(f"{name:7s} skip={int(bool(flags))}: {dt * 1e3:8.1f} ms ({n_q / dt / 1e3:7.0f} k queries/s), hit_count {st['hit_count'][0]:7.1f} ms, kmer {st['kmer_extract'][0]:5.1f}, "
              f"prob {st['prob_table'][0]:5.1f}, prefix {st['taxon_prefix'][0]:5.1f}; rows loaded {w['bitmap_bytes_read'] / n_q / 1e6:.2f} MB/query", flush=True)
        out[(name, flags)] = res
        if prune:
            print("   ", ix.debug_prune_stats())
    del ix
for flags in (0, rx.RTX_SKIP_EXACT_MATCHES):
    a, b = out[("full", flags)], out[("pruned", flags)]
    same_rows = np.array_equal(a.row_off, b.row_off) and np.array_equal(a.row_lineage, b.row_lineage)
    dconf = float(np.abs(a.row_conf - b.row_conf).max()) if same_rows else float("nan")
    print(f"skip={int(bool(flags))}: status equal {np.array_equal(a.status, b.status)}, t equal {np.array_equal(a.t, b.t)}, rows identical {same_rows}, "
          f"largest confidence difference {dconf:.3e}, global signal {float(np.abs(a.global_signal - b.global_signal).max()):.3e}, "
          f"local signal {float(np.abs(a.row_local_signal - b.row_local_signal).max()) if same_rows else float('nan'):.3e}")
