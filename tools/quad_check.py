#!/usr/bin/env python3
"""hit_count_quad_kernel (RTX_OPT_HIT_QUAD) against hit_count_kernel: identical results, then the time of both at
BASELINE.json configs[2] size.  Usage: tools/quad_check.py [refs] [queries]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

FIELDS = ("row_off", "row_lineage", "row_conf", "row_local_signal", "global_signal", "t", "status")


def same(a, b, what):
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), (what, f)


def small(n_refs, n_q, sub_batch=0, **kw):
    db = synth.make_db(n_refs, **kw)
    qs = synth.make_queries(db, n_q, exact_frac=0.15, n_frac=0.05)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    a = rx.Index(tree, hit_quad=False, sub_batch=sub_batch)
    b = rx.Index(tree, hit_quad=True, sub_batch=sub_batch)
    ex = a.exact_matches(qs.bases, qs.base_off)
    for skip in (False, True):
        ra = a.classify(qs.bases, qs.base_off, *ex, skip_exact_matches=skip)
        rb = b.classify(qs.bases, qs.base_off, *ex, skip_exact_matches=skip)
        same(ra, rb, (n_refs, n_q, skip))
        last0 = (n_q - 1) // (sub_batch or 10240) * (sub_batch or 10240)
    print(f"ok: {n_refs} refs, {n_q} queries, sub-batch {sub_batch}", a.work(), b.work(), flush=True)


def main():
    small(5184, 163)
    small(5184, 162, sub_batch=37)
    small(70000, 3001)
    small(20000, 64, length=150)
    n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
    n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 40_960
    db = synth.make_db(n_refs)
    qs = synth.make_queries(db, n_q)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    res = {}
    for quad in (False, True):
        ix = rx.Index(tree, hit_quad=quad, stage_timing=True)
        ex = ix.exact_matches(qs.bases, qs.base_off)
        ix.upload(qs.bases, qs.base_off, *ex)
        for _ in range(3):
            t0 = time.time()
            ix.run(0)
            ix.download(copy=False)
            dt = time.time() - t0
        res[quad] = ix.download()
        w = ix.work()
        print(f"quad={quad}: {n_q} queries in {dt * 1e3:.1f} ms = {n_q / dt:.0f} q/s; stages {ix.stage_times()}; "
              f"requested bytes/query {w['bitmap_bytes_read'] / n_q / 1e6:.2f} MB", flush=True)
        del ix
    same(res[False], res[True], "full size")
    print("quad check ok")


if __name__ == "__main__":
    main()
