#!/usr/bin/env python3
"""Full-size sanity run (BASELINE.json configs[2] shape: 500k-sequence database): size-independent properties
on every query of the batch + oracle parity on a small sample.  Usage: tools/scale_check.py [refs] [queries] [sample]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402


def main():
    n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
    n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    n_sample = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    t0 = time.time()
    db = synth.make_db(n_refs)
    qs = synth.make_queries(db, n_q)
    print(f"synthetic data: {time.time() - t0:.1f}s")
    t0 = time.time()
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    print(f"host tree (sort + taxonomy + exact map): {time.time() - t0:.1f}s")
    t0 = time.time()
    ix = rx.Index(tree)
    print(f"GPU index build from sequences: {time.time() - t0:.1f}s, {ix.device_bytes / 1e9:.2f} GB")
    ex_ids, ex_off = ix.exact_matches(qs.bases, qs.base_off)
    ix.upload(qs.bases, qs.base_off, ex_ids, ex_off)
    for _ in range(3):
        t0 = time.time()
        ix.run(0)
        ix.download(copy=False)          # device + host finalisation (no Python copies)
        dt = time.time() - t0
    res = ix.download()
    print(f"classify {n_q} queries: {dt * 1e3:.1f} ms -> {n_q / dt:.0f} q/s; stages {ix.stage_times()}")
    print("tile pruning:", ix.debug_prune_stats())
    print("HBM: index", round(ix.device_bytes / 1e9, 2), "GB; workspace", round(ix.workspace_bytes / 1e9, 2), "GB", {k: round(v / 1e9, 2) for k, v in ix.workspace_parts().items()})
    work = ix.work()
    print("work", work, "H_q/N =", work["sum_hits"] / n_q / n_refs)
    assert (res.status == 0).all()
    assert (np.diff(res.row_off) >= 1).all()
    # size-independent properties + oracle parity on the last few queries, re-run as a batch of their own
    # (the debug taps see the last sub-batch, and the processing order of a large batch is not the input order)
    m = max(n_sample, 6)
    lo = int(qs.base_off[n_q - m])
    sub_off = (qs.base_off[n_q - m:] - qs.base_off[n_q - m]).astype(np.uint64)
    sub_bases = qs.bases[lo:]
    sx_ids, sx_off = ix.exact_matches(sub_bases, sub_off)
    sub = ix.classify(sub_bases, sub_off, sx_ids, sx_off)
    for j in range(m):
        c = ix.debug_hit_counts(j)
        p = ix.debug_probs(j)
        assert abs(p.sum() - 1.0) < 1e-9, p.sum()
        assert int(c.astype(np.uint64).sum()) > 0
        assert int(c.max()) <= int(sub.t[j])
        q = n_q - m + j
        assert [(r.lineage, r.confidence_values) for r in sub.rows(j)] == [(r.lineage, r.confidence_values) for r in res.rows(q)]
    if n_sample:
        from oracle.oracle_py import Oracle

        t0 = time.time()
        otree = Oracle().tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
        print(f"oracle tree: {time.time() - t0:.1f}s")
        bad = 0
        for q in range(n_q - n_sample, n_q):
            t, counts = otree.hit_counts(qs.seq(q))
            assert np.array_equal(ix.debug_hit_counts(q - (n_q - m)), counts), q
            rows, _ = otree.classify(qs.seq(q), raw_confidence=True)
            got = res.rows(q)
            if [r.lineage for r in got] != [r["idx"] for r in rows] or [r.confidence_values for r in got] != [r["conf"] for r in rows]:
                bad += 1
        print(f"oracle parity on {n_sample} queries: counts bit-exact, {bad} row mismatches")
        assert bad == 0
    print("scale check ok")


if __name__ == "__main__":
    main()
